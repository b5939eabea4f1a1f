#!/usr/bin/env python
"""Time the physics kernel alone under a few configurations (GPU box)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vid2player3d_amd import _lib
from vid2player3d_amd.tasks import HumanoidSMPLIM, default_cfg


def run(n, iters, contact, steps=64, fall=True):
    cfg = default_cfg(n, debug_contacts=1, synthetic_motions={"seed": 7, "num_clips": 64}, enable_contact=contact)
    cfg["sim"]["physx"]["num_position_iterations"] = iters
    task = HumanoidSMPLIM(cfg, device_type="cuda", device_id=0)
    g = torch.Generator(device=task.device); g.manual_seed(1)
    ms = []
    for i in range(steps):
        if i % 32 == 0:
            task.reset()
        a = 0.17 * torch.randn((n, 75), device=task.device, generator=g)
        a[:, :69] += task._target_dof_pos
        task.pre_physics_step(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(task._lib.v2p_env_physics(task._h_env, task._stream()), "phys")
        e1.record()
        _lib.check(task._lib.v2p_env_export(task._h_env, task._stream()), "exp")
        task.post_physics_step()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ncont = (task.debug_contacts() >= 0).any(dim=2).float().sum(dim=1).mean().item()
    task.close()
    ms = np.array(ms)
    return {"n": n, "iters": iters, "contact": contact, "mean_ms": float(ms.mean()), "first8_ms": float(ms[:8].mean()), "last8_ms": float(ms[-8:].mean()),
            "bodies_in_contact_end": ncont}


if __name__ == "__main__":
    for (n, it, c) in [(8192, 4, True), (8192, 1, True), (8192, 0, True), (8192, 4, False), (1024, 4, True), (65536, 4, True)]:
        print(json.dumps(run(n, it, c)), flush=True)
