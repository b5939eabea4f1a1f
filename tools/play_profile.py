"""T_play on its own: the rollout half of the PPO loop - play_steps of ppo.PPOAgent, 8192 envs, horizon 32 - with no update
in the process, so that a rocprofv3 kernel trace of it lists what runs between two physics launches.

    python tools/play_profile.py [epochs, default 4]                                      -> wall-clock per step (no profiler)
    rocprofv3 --kernel-trace --stats -d DIR -o play -- python tools/play_profile.py 4      -> DIR/play_results.db; tools/play_summary.py"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vid2player3d_amd.ppo import PPOAgent  # noqa: E402

if __name__ == "__main__":
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    task = bench.build_task(8192, 0, 7)
    agent = PPOAgent(task, seed=7)
    agent.play_steps()  # warm-up (allocator, rocBLAS heuristics)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enq = 0.0
    for _ in range(epochs):
        ta = time.perf_counter()
        agent.play_steps()
        enq += time.perf_counter() - ta  # (host time to ENQUEUE the epoch: equal to the wall clock = the rollout is launch bound)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * agent.horizon_length
    print("[play] %d epochs x %d steps x %d envs: %.3f ms per step, %.2f M frames/s (fps step); host enqueue time %.3f ms per step" % (epochs, agent.horizon_length, task.num_envs, 1e3 * dt / steps, task.num_envs * steps / dt / 1e6, 1e3 * enq / steps))
