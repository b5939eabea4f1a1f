"""Two PROCESSES on one GPU: the substep jobs of a launch wait for each other across workgroups, and their
progress rests on dispatch order with a time-out + replay as the fallback.  Under a second process the GPU time-slices two queues of
such launches: this soak runs 2 x 4096 envs for N steps in two processes side by side and reports, per process, finiteness, the number
of jobs that had to be recomputed (`job_recoveries`) and the throughput.      python tools/soak2.py [steps, default 2000] [envs per process]
[--procs P, default 2] [--cpus-per-proc C: every process pinned to its own C CPUs with sched_setaffinity - P = 4 or 8 with C = 2 is the
host-side budget 8 ranks see under the GPU boxes' 16-CPU cgroup quota (VERDICT r5 #6c); the line then also carries the HOST time a step
takes to enqueue (wall time of task.step without synchronisation, median over the run) next to the device-bound step time]"""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def worker(rank, steps, nenv, cpus=0):
    if cpus > 0:  # this process on its own `cpus` CPUs (of the ones the cgroup lets it run on)
        allowed = sorted(os.sched_getaffinity(0))
        mine = allowed[(rank * cpus) % len(allowed):][:cpus] or allowed[:cpus]
        os.sched_setaffinity(0, mine)
    import torch
    if cpus > 0:
        torch.set_num_threads(cpus)

    sys.path.insert(0, ROOT)
    import bench

    task = bench.build_task(nenv, 0, 7 + rank, substep_jobs=True, env_extra={"substep_jobs": 2})  # 2: always cut into jobs
    g = torch.Generator(device="cuda")
    g.manual_seed(1 + rank)
    n = task.num_envs
    bad = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enq = []
    for i in range(steps):
        if i % 32 == 0:
            task.reset()
        a = bench.make_actions(task, 0.25 * torch.randn((n, 75), device="cuda", generator=g))
        h0 = time.perf_counter()
        task.step(a)
        enq.append(time.perf_counter() - h0)
        if i % 200 == 199:
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(task.obs_buf).all() and torch.isfinite(task._rigid_body_state).all() and torch.isfinite(task.rew_buf).all())
            bad += not ok
    task.check()
    dt = time.perf_counter() - t0
    enq.sort()
    print("[soak2 rank %d] %d steps of %d envs: %s | %.2f M env-steps/s beside the other process(es) | substep jobs recomputed after waiting in vain: %d | build: %s | "
          "host: %s CPUs, task.step enqueue median %.1f us, p99 %.1f us; wall per step %.1f us"
          % (rank, steps, n, "finite" if bad == 0 else "NON-FINITE VALUES", n * steps / dt / 1e6, task.job_recoveries(), task.kernel_build(),
             len(os.sched_getaffinity(0)), 1e6 * enq[len(enq) // 2], 1e6 * enq[int(0.99 * len(enq))], 1e6 * dt / steps), flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        sys.exit(worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])))
    argv = list(sys.argv[1:])
    nproc, cpus = 2, 0
    for flag in ("--procs", "--cpus-per-proc"):
        if flag in argv:
            k = argv.index(flag)
            val = int(argv[k + 1])
            del argv[k:k + 2]
            nproc, cpus = (val, cpus) if flag == "--procs" else (nproc, val)
    steps = int(argv[0]) if len(argv) > 0 else 2000
    nenv = int(argv[1]) if len(argv) > 1 else 4096
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(steps), str(nenv), str(cpus)], env=env) for r in range(nproc)]
    rcs = [p.wait(timeout=1800) for p in procs]
    print("SOAK2", "OK" if all(rc == 0 for rc in rcs) else "FAILED %s" % rcs)
