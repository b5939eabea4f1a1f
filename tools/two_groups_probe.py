#!/usr/bin/env python
"""Throughput of G independent env batches (8192/G envs each) stepped on G streams: the tail of one batch's physics launch (few heavy
waves left) overlaps the head of another's.  Same total work as bench.py's single 8192-env batch; a usage pattern, not the headline."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = 8192 // G
STEPS, WARM, H = 320, 64, bench.HORIZON
tasks = [bench.build_task(N, 0, 7 + g) for g in range(G)]
streams = [torch.cuda.Stream() for _ in range(G)]
gens = [torch.Generator(device="cuda") for _ in range(G)]
noise = []
for g in range(G):
    gens[g].manual_seed(7 + g)
    noise.append([0.17 * torch.randn((N, 75), device="cuda", generator=gens[g]) for _ in range(H)])
torch.cuda.synchronize()


def run(n):
    for i in range(n):
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                if i % H == 0:
                    tasks[g].reset()
                tasks[g].step_fused(bench.make_actions(tasks[g], noise[g][i % H]))


run(WARM)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(STEPS)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("groups %d x %d envs: %.2f M env-steps/s  (%.3f ms per step of all groups)" % (G, N, G * N * STEPS / dt / 1e6, 1e3 * dt / STEPS))
