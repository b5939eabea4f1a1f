"""Two PROCESSES on one GPU: the substep jobs of a launch wait for each other across workgroups, and their
progress rests on dispatch order with a time-out + replay as the fallback.  Under a second process the GPU time-slices two queues of
such launches: this soak runs 2 x 4096 envs for N steps in two processes side by side and reports, per process, finiteness, the number
of jobs that had to be recomputed (`job_recoveries`) and the throughput.      python tools/soak2.py [steps, default 2000] [envs per process]"""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def worker(rank, steps, nenv):
    import torch

    sys.path.insert(0, ROOT)
    import bench

    task = bench.build_task(nenv, 0, 7 + rank, substep_jobs=True, env_extra={"substep_jobs": 2})  # 2: always cut into jobs
    g = torch.Generator(device="cuda")
    g.manual_seed(1 + rank)
    n = task.num_envs
    bad = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i % 32 == 0:
            task.reset()
        task.step(bench.make_actions(task, 0.25 * torch.randn((n, 75), device="cuda", generator=g)))
        if i % 200 == 199:
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(task.obs_buf).all() and torch.isfinite(task._rigid_body_state).all() and torch.isfinite(task.rew_buf).all())
            bad += not ok
    task.check()
    dt = time.perf_counter() - t0
    print("[soak2 rank %d] %d steps of %d envs: %s | %.2f M env-steps/s beside the other process | substep jobs recomputed after waiting in vain: %d | build: %s"
          % (rank, steps, n, "finite" if bad == 0 else "NON-FINITE VALUES", n * steps / dt / 1e6, task.job_recoveries(), task.kernel_build()), flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        sys.exit(worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])))
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(steps), str(nenv)], env=env) for r in range(2)]
    rcs = [p.wait(timeout=1800) for p in procs]
    print("SOAK2", "OK" if all(rc == 0 for rc in rcs) else "FAILED %s" % rcs)
