#!/usr/bin/env python
"""Step time and contact load by position in the 32-step epoch (the bench workload): where the launch time goes.

For every step k of the epoch: wall time of the fused step (events around it, mean over the epochs), and the distribution over
envs of the number of bodies that carry a contact force after the step (max / p99 / mean) together with the number of sweep
GROUPS those touched bodies form (a touched body whose parent is the touched body right before it continues a group).
usage (GPU box): python tools/epoch_profile.py [--epochs 6] [--num-envs 8192]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--num-envs", type=int, default=8192)
    ap.add_argument("--action-noise", type=float, default=0.17)
    ap.add_argument("--job-mono", type=int, default=None, help="v2p_sim_cfg.job_mono_permille")
    ap.add_argument("--kernel-build", type=int, default=None, help="v2p_sim_cfg.kernel_build (1 = LDS-parked, 2 = registers)")
    ap.add_argument("--brief", action="store_true", help="one line: ms of every 4th step")
    args = ap.parse_args()
    n = args.num_envs
    task = bench.build_task(n, 0, seed=7, substep_jobs=True, env_extra={k: v for k, v in (("job_mono_permille", args.job_mono), ("kernel_build", args.kernel_build)) if v is not None})
    from vid2player3d_amd.model import load_baked_model
    parents = np.asarray(load_baked_model().parents)
    dev = task.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    H = bench.HORIZON
    noise = [args.action_noise * torch.randn((n, 75), device=dev, generator=gen) for _ in range(H)]
    ms = np.zeros((args.epochs, H))
    kmax = np.zeros((args.epochs, H)); k99 = np.zeros_like(kmax); kmean = np.zeros_like(kmax)
    gmax = np.zeros_like(kmax); g99 = np.zeros_like(kmax)
    par_t = torch.tensor(parents, device=dev)
    for ep in range(args.epochs + 1):
        task.reset()
        for k in range(H):
            a = bench.make_actions(task, noise[k])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            task.step_fused(a)
            e1.record()
            torch.cuda.synchronize()
            if ep == 0:
                continue  # warm-up epoch
            ms[ep - 1, k] = e0.elapsed_time(e1)
            touched = (task._contact_forces.abs().sum(-1) > 0)  # [N,24]
            K = touched.sum(1).float()
            # groups: a touched body continues a group when the previous touched body (ascending) is its parent
            idx = torch.arange(24, device=dev).expand(n, 24)
            prev = torch.where(touched, idx, torch.full_like(idx, -1))
            prev = torch.cummax(prev, 1).values  # last touched index <= j
            prevb = torch.cat([torch.full((n, 1), -1, device=dev, dtype=prev.dtype), prev[:, :-1]], 1)  # last touched index < j
            cont = touched & (prevb == par_t.expand(n, 24)) & (prevb >= 0)
            G = (touched & ~cont).sum(1).float()
            kmax[ep - 1, k] = K.max().item(); k99[ep - 1, k] = torch.quantile(K, 0.99).item(); kmean[ep - 1, k] = K.mean().item()
            gmax[ep - 1, k] = G.max().item(); g99[ep - 1, k] = torch.quantile(G, 0.99).item()
    if args.brief:
        print("job_mono %s: " % args.job_mono + " ".join("%d:%.3f" % (k, ms[:, k].mean()) for k in range(0, H, 4)) + " 31:%.3f | mean %.4f" % (ms[:, 31].mean(), ms.mean()))
        return
    print("step  ms/step   touched bodies max / p99 / mean    groups max / p99")
    for k in range(H):
        print("%3d   %.4f    %5.1f %5.1f %5.2f                 %5.1f %5.1f" % (k, ms[:, k].mean(), kmax[:, k].mean(), k99[:, k].mean(), kmean[:, k].mean(), gmax[:, k].mean(), g99[:, k].mean()))
    print("mean ms/step %.4f" % ms.mean())


if __name__ == "__main__":
    main()
