#!/usr/bin/env python
"""bench.py -- env-steps/s of the imitation rollout hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--num-envs E] [--no-contact]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): amass_im, num_envs=8192 per
GPU, full contact PGS + imitation reward, 64 seeded synthetic clips, one body shape,
random residual-policy actions a ~ N(target_dof_pos (+) 0, 0.17^2).  A "step" is one VecTask
`step()` of ALL envs of a rank (pre-physics + 4 physics substeps + export + post: new target, obs,
reward, reset); every `horizon`=32 steps the per-epoch `reset()` of all envs (RSI + target + 48-frame
context window) runs INSIDE the timed region, as in the reference's play_steps.  Policy inference is
excluded.  Envs shard across ranks with no data-path collective ("scaling": "weak").

One JSON line on rank 0, with
  roofline      dominant kernel = physics_ll_kernel: algorithmic HBM bytes of one step (SURVEY.md 8d:
                9,896 B per env-step x envs per launch) / its mean duration measured here with HIP
                events on the launch stream; peak = 8 TB/s.
  cpu_baseline  the oracle (C physics restatement + numpy task ops) timed on this host's cores on a
                bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ALGO_BYTES_PER_ENV_STEP = 9896          # SURVEY.md 8(d) per-step total (config 3)
ALGO_BYTES_PER_ENV_STEP_AMORTISED = 16090  # + per-epoch reset/context / 32
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
HORIZON = 32


def build_task(num_envs, device_id, seed, contact=True, per_clip_shapes=False, djokovic=False, freeze=False):
    from vid2player3d_amd.tasks import HumanoidSMPLIM, default_cfg

    cfg = default_cfg(num_envs, synthetic_motions={"seed": 7, "num_clips": 64, "min_frames": 90, "max_frames": 300},
                      enable_contact=contact)
    if freeze:  # NOT the reference's behaviour (it keeps simulating terminated envs as ragdolls): reported separately, never as `value` of the default run
        cfg["env"]["freeze_terminated_envs"] = True
    if djokovic:  # BASELINE config 4 = cfg/djokovic_im.yaml: same task class, head termination height -0.5, faster (tennis-like) clips
        cfg["env"]["terminationHeadHeight"] = -0.5
        cfg["env"]["synthetic_motions"]["speed"] = 2.0
    if per_clip_shapes:  # one body shape per clip like the reference's per-clip SMPL assets: 64 uniformly scaled bodies, 0.85 .. 1.15
        from vid2player3d_amd.model import load_baked_model

        base = load_baked_model()
        cfg["env"]["body_model"] = [base.scaled(0.85 + 0.3 * k / 63.0) for k in range(64)]
    torch.manual_seed(seed)
    return HumanoidSMPLIM(cfg, device_type="cuda", device_id=device_id)


_ACT_MASK = {}


def make_actions(task, noise):
    """Stand-in policy: a = [target_dof_pos + noise, noise] (sigma 0.17), one elementwise kernel: the 75 columns starting at the
    target's dof_pos (dof_pos 69 | root_vel 3 | root_ang_vel 3 of the packed target row) times a [1]*69 + [0]*6 mask."""
    dev = noise.device
    if dev not in _ACT_MASK:
        _ACT_MASK[dev] = torch.cat([torch.ones(69, device=dev), torch.zeros(6, device=dev)])
    tgt = task._target_dof_pos  # [N,69] view into the current packed target [N,331]
    tgt75 = torch.as_strided(tgt, (tgt.shape[0], 75), (tgt.stride(0), 1), tgt.storage_offset())
    return torch.addcmul(noise, tgt75, _ACT_MASK[dev])


def cpu_baseline(num_envs_sample=None, max_steps=100000, budget_s=12.0):
    """Oracle on the host cores: C physics (one env per thread-pool task) + numpy task ops."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import task_oracle as O
    from oracle.phys_oracle import PhysOracle, default_params
    from vid2player3d_amd import motion_tables, synth
    from vid2player3d_amd.model import load_baked_model

    cores = os.cpu_count() or 1
    bm = load_baked_model()
    clips = synth.make_clips(7, 8, 90, 300)
    tabs = motion_tables.build_tables(clips, bm.parents, bm.local_pos)
    rng = np.random.default_rng(7)
    n = num_envs_sample or max(64, 4 * cores)
    ids = np.arange(n) % 8
    task = O.TaskOracle(tabs, ids, bm.kp.astype(np.float32))
    task.reset_all(rng.uniform(0.1, 1.0, size=n).astype(np.float32))
    oracles = [PhysOracle(bm, default_params()) for _ in range(n)]
    for e in range(n):
        oracles[e].set_state(task.root_states[e], task.dof_pos[e], task.dof_vel[e])

    def one(e, pd, f, t):
        oracles[e].step(pd_target=pd, ext_force=f, ext_torque=t, nsub=4, hold=2)
        return oracles[e].get_state()

    pool = ThreadPoolExecutor(max_workers=cores)
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and time.perf_counter() - t0 < budget_s:
        steps += 1
        act = np.concatenate([task.target[2] + rng.normal(0, 0.17, size=(n, 69)), rng.normal(0, 0.17, size=(n, 6))], axis=1).astype(np.float32)
        _, pd, _, force, torque = task.pre_physics_step(act)
        res = list(pool.map(lambda e: one(e, pd[e], force[e], torque[e]), range(n)))
        task.set_sim_state(np.stack([r[1] for r in res]).astype(np.float32), np.stack([r[2] for r in res]).astype(np.float32),
                           np.stack([r[3] for r in res]).astype(np.float32))
        task.post_physics_step()
    dt = time.perf_counter() - t0
    pool.shutdown()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d control steps (4 substeps each, contacts on), C float64 dense oracle + numpy task ops, %d threads, %.1f s"
                      % (n, steps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--num-envs", type=int, default=8192, help="envs per GPU")
    ap.add_argument("--no-contact", action="store_true", help="BASELINE config 2 (PD only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--freeze-terminated", action="store_true", help="opt-in engine feature: terminated envs are not simulated until the epoch reset (not reference behaviour)")
    ap.add_argument("--djokovic", action="store_true", help="BASELINE config 4 (djokovic_im.yaml: terminationHeadHeight -0.5, faster clips)")
    ap.add_argument("--per-clip-shapes", action="store_true", help="one body shape per clip (64 scaled bodies) instead of one shape for all envs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rollout engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("V2P_BENCH_FORCE_DIST"):  # (the env var runs the collective path on a single GPU: CI of the N>1 code)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vid2player3d_amd import build
    if local_rank == 0:
        build.build()  # no-op when the in-tree .so is current; one rank per node compiles otherwise
    if dist is not None:
        dist.barrier()
    n = args.num_envs
    task = build_task(n, local_rank, seed=7 + rank, contact=not args.no_contact, per_clip_shapes=args.per_clip_shapes, djokovic=args.djokovic, freeze=args.freeze_terminated)  # per-rank seed like run.py:37
    dev = task.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)
    noise = [0.17 * torch.randn((n, 75), device=dev, generator=gen) for _ in range(HORIZON)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ev_pairs = []

    def run(nsteps, timed_events=False):
        for i in range(nsteps):
            if i % HORIZON == 0:
                task.reset()
            a = make_actions(task, noise[i % HORIZON])
            if timed_events:
                task.pre_physics_step(a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                from vid2player3d_amd import _lib
                _lib.check(task._lib.v2p_env_physics(task._h_env, task._stream()), "v2p_env_physics")
                e1.record()
                _lib.check(task._lib.v2p_env_export(task._h_env, task._stream()), "v2p_env_export")
                task.post_physics_step()
                ev_pairs.append((e0, e1))
            else:
                task.step_fused(a)

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel duration with HIP events on the launch stream (separate, untimed pass)
    run(HORIZON, timed_events=True)
    torch.cuda.synchronize()
    phys_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs]))
    alive = float((task.reset_buf == 0).float().mean().item())

    if rank == 0:
        value = world * n * args.steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (phys_ms * 1e-3) / 1e9
        valu = None  # VALU-side view of the same kernel from the committed SQ counter passes (tools/valu_probe.sh): the binding resource
        vc = os.path.join(REPO, "profiles", "r01e_valu_counters.json")
        if os.path.exists(vc):
            try:
                c = json.load(open(vc))
                valu = {"valu_insts_per_wave": c["SQ_INSTS_VALU"]["avg"] / c["SQ_WAVES"]["avg"],
                        "wave_time_issuing_valu": c["SQ_ACTIVE_INST_VALU"]["avg"] / c["SQ_WAVE_CYCLES"]["avg"],
                        "lanes_active_per_valu_op": c["SQ_THREAD_CYCLES_VALU"]["avg"] / c["SQ_ACTIVE_INST_VALU"]["avg"],
                        "source": "profiles/r01e_valu_counters.json (rocprofv3 --pmc, 8192 envs)"}
            except Exception:
                valu = None
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("physics_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec at num_envs=8192, SMPL humanoid imitation", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "amass_im num_envs=%d per GPU, %s, imitation reward, per-epoch reset+context every %d steps, 64 synthetic clips%s"
                                   % (n, "PD control only (no contact solve)" if args.no_contact else "full contact PGS (4 substeps x 4 iterations)", HORIZON,
                                      (", one body shape per clip" if args.per_clip_shapes else "") + (", djokovic_im variant" if args.djokovic else "") + (", TERMINATED ENVS FROZEN (not reference behaviour)" if args.freeze_terminated else "")),
                       "num_envs_per_gpu": n, "global_envs": world * n, "parallelism": "env-sharded x%d, no data-path collective" % world,
                       "alive_fraction_at_end": alive},
            "roofline": {"bound": "hbm", "kernel": "physics_ll_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": phys_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "note": "latency/VALU bound, not HBM bound: see DESIGN.md", "valu": valu},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
