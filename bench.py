#!/usr/bin/env python
"""bench.py -- env-steps/s of the imitation rollout hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--num-envs E] [--no-contact] ...

With --gpus N > 1 and no torch.distributed environment the script launches itself as N ranks (python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...), one rank per GPU over RCCL; started under an external torchrun it
just takes its rank from the environment.

Workload (BASELINE.json configs[2], the one the metric is quoted on): amass_im, num_envs=8192 per GPU, full contact PGS +
imitation reward, 64 seeded synthetic clips, one body shape, random residual-policy actions
a ~ N(target_dof_pos (+) 0, 0.17^2).  A "step" is one VecTask `step()` of ALL envs of a rank (pre-physics + 4 physics substeps +
post: new target, obs, reward, reset); every `horizon`=32 steps the per-epoch `reset()` of all envs (RSI + target + 48-frame
context window) runs INSIDE the timed region, as in the reference's play_steps.  Policy inference is excluded.  Envs shard across
ranks with no data-path collective ("scaling": "weak").

One JSON line on rank 0, with
  roofline      dominant kernel = physics_ll_kernel.  kernel_ms = its mean duration over the TIMED steps themselves, from HIP
                events the engine records on the launch stream around one launch in eight, the bracketed position rotating through
                the epoch (v2p_env_profile_begin_sampled/_end; --kernel-events 1: every launch, which costs 2 %).  achieved =
                algorithmic HBM bytes of one step (9,896 B per env-step x envs per launch) / kernel_ms against
                8 TB/s.  The kernel is bound by the dependent VALU / LDS chains of its waves, not by HBM, so the line also carries valu_frac = fp32 FLOP/s of the
                kernel / 157.3 TFLOP/s (vector peak), with the FLOPs per launch taken from the committed SQ counter profile
                (labelled "from_profiles"); `traffic` (HBM bytes per launch, PMC) is "from_profiles" as well.
  cpu_baseline  the oracle timed on this host's cores on a bounded sample of the same workload (rank 0, N=1 only): the C float64
                physics restatement stepped as ONE batched OpenMP call per control step + the numpy task ops, on as many threads as the
                process may run at once (affinity capped by the cgroup CPU quota); kind "port".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ALGO_BYTES_PER_ENV_STEP = 9896          # SURVEY.md 8(d) per-step total (config 3)
ALGO_BYTES_PER_ENV_STEP_AMORTISED = 16090  # + per-epoch reset/context / 32
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md
# the measured VALU issue ceiling of a saturated SIMD (tools/ubench/valu_issue.hip cu, tools/valu_issue_probe.sh), in cycles of the clock the
# microbenchmark was MEASURED to run at
VALU_CEILING = {"cycles": 2.34, "note": "profiles/r06_valu_issue.txt (tools/valu_issue_probe.sh): one workgroup per CU, 4 waves of dependent v_fma_f32 chains on every SIMD, "
                                         "4 x 262144 instructions per SIMD in 1.100 ms = 1.049 ns each at a MEASURED 2228 MHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time) = 2.34 cycles "
                                         "(2.38 at 2 waves); round 4's 2.67 - 2.88 were 'nominal cycles at an assumed 2.4 GHz' of sub-0.3 ms kernels, launch overhead included"}
HORIZON = 32
WHOLE_EPOCHS = 10                       # the separately timed whole-epoch block behind a short / ragged --steps


# ---------------------------------------------------------------------------------------------- launching the ranks
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_command(gpus, argv, port=None):
    """The command line that runs this script as `gpus` ranks of one node (the driver's own launch line)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv, stub):
    """--gpus N without a torch.distributed environment: become the launcher of N ranks; their rank 0 prints the JSON line."""
    if not stub:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < gpus:
            raise SystemExit("--gpus %d: only %d GPU(s) visible on this node" % (gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between the ranks
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(rank_command(gpus, argv), env=env)


# ---------------------------------------------------------------------------------------------- the workload
def build_task(num_envs, device_id, seed, contact=True, per_clip_shapes=False, num_shapes=64, djokovic=False, freeze=False, solver="pgs", racket_ball=False, substep_jobs=False, joint_limits=None, env_extra=None):
    from vid2player3d_amd.tasks import HumanoidSMPLIM, HumanoidSMPLIMRacketBall, default_cfg

    num_clips = num_shapes if per_clip_shapes else 64
    cfg = default_cfg(num_envs, synthetic_motions={"seed": 7, "num_clips": num_clips, "min_frames": 90, "max_frames": 300},
                      enable_contact=contact, contact_solver=solver, substep_jobs=substep_jobs)
    if joint_limits is not None:
        cfg["env"]["joint_limits"] = bool(joint_limits)
    cfg["env"].update(env_extra or {})
    if freeze:  # NOT the reference's behaviour (it keeps simulating terminated envs as ragdolls): reported separately, never as `value` of the default run
        cfg["env"]["freeze_terminated_envs"] = True
    if djokovic:  # BASELINE config 4 = cfg/djokovic_im.yaml: same task class, head termination height -0.5, faster (tennis-like) clips
        cfg["env"]["terminationHeadHeight"] = -0.5
        cfg["env"]["synthetic_motions"]["speed"] = 2.0
    if per_clip_shapes:  # one body shape per clip like the reference's per-clip SMPL assets: non-uniform shapes built from vertex clouds
        from vid2player3d_amd import body_shapes
        from vid2player3d_amd.model import load_baked_model

        t0 = time.perf_counter()
        cfg["env"]["body_model"] = body_shapes.synthetic_shape_family(load_baked_model(), num_shapes, seed=7, device="cuda:%d" % device_id)
        sys.stderr.write("bench.py: %d body shapes compiled on the device in %.2f s\n" % (num_shapes, time.perf_counter() - t0))
        if num_shapes != 64:  # env i -> clip i %% num_shapes -> its shape: every shape is simulated (64: clips sampled at random, as in rounds 3-4)
            cfg["env"]["sample_first_motions"] = True
    torch.manual_seed(seed)
    if racket_ball:  # BASELINE config 4 as it is worded: racket welded to the wrist + free ball in every env
        task = HumanoidSMPLIMRacketBall(cfg, device_type="cuda", device_id=device_id)
        inner_reset = task.reset

        def reset_with_serve(env_ids=None):  # every epoch a ball is served at the player from 8 m (so that ground bounces and some racket hits occur)
            inner_reset(env_ids)
            n, dev = task.num_envs, task.device
            root = task._humanoid_root_states[:, 0:3]
            g = torch.Generator(device=dev)
            g.manual_seed(seed + int(task.progress_buf.sum().item()) % 7)
            jitter = torch.rand((n, 3), device=dev, generator=g)
            pos = root + torch.tensor([8.0, 0.0, 0.3], device=dev) + jitter * torch.tensor([1.0, 1.0, 1.0], device=dev)
            vel = torch.tensor([-22.0, 0.0, 4.0], device=dev) + (jitter - 0.5) * torch.tensor([6.0, 3.0, 3.0], device=dev)
            task.reset_balls(torch.arange(n, device=dev), pos, vel, torch.tensor([0.0, -150.0, 0.0], device=dev).expand(n, 3))

        task.reset = reset_with_serve
        return task
    return HumanoidSMPLIM(cfg, device_type="cuda", device_id=device_id)


class StubTask:
    """Stand-in for the task when the LAUNCH logic of this script is tested without GPUs (tests/test_bench_launch.py): a few torch
    ops on the CPU per step, the same surface bench.py drives.  It simulates nothing and is never part of a measured number."""

    def __init__(self, num_envs):
        self.num_envs, self.device = num_envs, "cpu"
        self._target_dof_pos = torch.zeros((num_envs, 331))[:, 7:76]
        self.reset_buf = torch.zeros(num_envs, dtype=torch.long)
        self._x = torch.zeros(num_envs, 75)
        self._launches = 0

    def reset(self):
        self.reset_buf.zero_()

    def step_fused(self, a):
        self._x = 0.5 * self._x + a
        self._launches += 1

    def profile_begin(self, n):
        self._launches = 0

    def profile_end(self):
        return 0.001 * self._launches, self._launches


class StubAgent:
    """Stand-in for PPOAgent when the launch / aggregation logic of `--ppo` is tested without GPUs: an epoch of it moves a few CPU
    tensors and all-reduces one of them like the update does.  Never part of a measured number."""

    def __init__(self, task):
        self.task, self.epoch = task, 0

    def train_epoch(self):
        t0 = time.perf_counter()
        for _ in range(HORIZON):
            self.task.step_fused(torch.zeros(self.task.num_envs, 75))
        t1 = time.perf_counter()
        g = torch.ones(16)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(g)
        self.epoch += 1
        return {"play_time": t1 - t0, "total_time": time.perf_counter() - t0, "frames": self.task.num_envs * HORIZON, "step_rewards": 0.0,
                "step_sub_rewards": [0.0] * 4, "alive_ratio": 1.0, "world": float(g[0])}

    def format_epoch_line(self, r):
        return "stub epoch %d: world %d" % (self.epoch, int(r["world"]))


def print_last(line, dist):
    """The JSON line as the LAST thing on stdout: RCCL writes its version banner through C stdio, which a redirected stdout holds back
    until the process exits - after anything Python printed.  Every rank drains its C buffers, the ranks meet, rank 0 prints."""
    import ctypes

    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
    if line is not None:
        print(line)
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


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


def make_epoch_actions(task, noise_all, out):
    """The same stand-in policy for a whole epoch at once, right after the reset: the target DOF positions of step k are frame
    context_padding + k of the context window the reset has just built (what the reference's residual policy adds to its mean,
    im_network_builder.py:226-228) - two elementwise kernels per epoch instead of one per step between the physics launches."""
    pad = task.context_padding
    torch.add(noise_all[..., :69], task.context_feat[:, pad:pad + noise_all.shape[0], 168:237].transpose(0, 1), out=out[..., :69])
    out[..., 69:].copy_(noise_all[..., 69:])
    return out


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """CPUs this process can actually run on at once: the affinity mask, capped by the cgroup's CPU quota (the GPU boxes of this pool
    show 256 logical CPUs and a quota of 16: more threads than the quota only throttle, profiles/r05_cpu_scaling_*.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(sizes=((4, 32, 3.0), (1024, 8, 5.0), (8192, 3, 12.0)), sigma=0.17):
    """The oracle on the host cores, same workload (contacts on, sigma-noise actions around the target pose) at the env counts BASELINE.md
    section 3 names (4, 1024, 8192): per control step ONE batched C call for the physics of all envs (OpenMP over envs; the float64 dense
    restatement built with -O3 -mavx2 -mfma, oracle/phys/Makefile `fast`) and the numpy task ops, sharded over the same number of
    threads (numpy releases the GIL inside its loops).  Threads = the CPUs the process may use at once (affinity capped by the cgroup
    quota).  sizes: (envs, most steps, seconds budget) - a bounded sample each; `value` is the 8192-env figure (the size the metric is
    quoted on); the per-thread physics rate is reported at every size, and `scaling_ok` says whether the largest size keeps at least half
    the per-thread rate of the smallest (a harness that loses more than that is measuring itself)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import task_oracle as O
    from oracle.phys_oracle import FAST_FLAGS, BatchOracle, default_params, lib_fast
    from vid2player3d_amd import motion_tables, synth
    from vid2player3d_amd.model import load_baked_model

    logical = os.cpu_count() or 1
    threads, quota = usable_cores()
    threads = min(threads, lib_fast().v2p_oracle_max_threads())
    bm = load_baked_model()
    clips = synth.make_clips(7, 8, 90, 300)
    tabs = motion_tables.build_tables(clips, bm.parents, bm.local_pos)
    kp32 = bm.kp.astype(np.float32)
    by_n = {}
    pool = ThreadPoolExecutor(max_workers=threads)
    for n, max_steps, budget_s in sizes:
        rng = np.random.default_rng(7)
        th = min(threads, n)
        shards = max(1, min(th, n // 128))  # task ops: one numpy oracle per shard of >= 128 envs, stepped in parallel
        bounds = np.linspace(0, n, shards + 1).astype(int)
        ids = np.arange(n) % 8
        tasks = [O.TaskOracle(tabs, ids[a:b], kp32) for a, b in zip(bounds[:-1], bounds[1:])]
        times0 = rng.uniform(0.1, 1.0, size=n).astype(np.float32)
        for t, a, b in zip(tasks, bounds[:-1], bounds[1:]):
            t.reset_all(times0[a:b])
        oracle = BatchOracle(bm, n, default_params(), threads=th, fast=True)
        oracle.set_state(np.concatenate([t.root_states for t in tasks]), np.concatenate([t.dof_pos for t in tasks]), np.concatenate([t.dof_vel for t in tasks]))
        pd, force, torque = np.zeros((n, 69), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)

        def pre(k):
            t, a, b = tasks[k], bounds[k], bounds[k + 1]
            r = np.random.default_rng((7, steps, k))
            act = np.concatenate([t.target[2] + r.normal(0, sigma, size=(b - a, 69)), r.normal(0, sigma, size=(b - a, 6))], axis=1).astype(np.float32)
            _, pd[a:b], _, force[a:b], torque[a:b] = t.pre_physics_step(act)

        def post(k):
            t, a, b = tasks[k], bounds[k], bounds[k + 1]
            t.set_sim_state(res["dpos"][a:b].astype(np.float32), res["dvel"][a:b].astype(np.float32), res["rb"][a:b].astype(np.float32))
            t.post_physics_step()

        t_phys = t_task = 0.0
        steps = -1  # (one untimed step first: OpenMP team start-up, first touch of the work arrays)
        t0 = time.perf_counter()
        while steps < max_steps and (steps <= 0 or time.perf_counter() - t0 < budget_s):
            steps += 1
            if steps == 0:
                list(pool.map(pre, range(shards)))
                res = oracle.step(pd, force, torque, nsub=4, hold=2)
                list(pool.map(post, range(shards)))
                t0 = time.perf_counter()
                continue
            ta = time.perf_counter()
            list(pool.map(pre, range(shards)))
            tb = time.perf_counter()
            res = oracle.step(pd, force, torque, nsub=4, hold=2)
            tc = time.perf_counter()
            list(pool.map(post, range(shards)))
            td = time.perf_counter()
            t_phys += tc - tb
            t_task += (tb - ta) + (td - tc)
        by_n[str(n)] = {"value": n * steps / (t_phys + t_task), "physics_env_steps_per_s": n * steps / t_phys, "task_ops_env_steps_per_s": n * steps / t_task,
                        "physics_threads": th, "physics_env_steps_per_s_per_thread": n * steps / t_phys / th, "task_ops_threads": shards, "steps": steps,
                        "seconds": {"physics": t_phys, "task_ops": t_task}}
    pool.shutdown()
    ref_ops = None
    try:  # the reference's OWN task code timed in the build container (tools/ref_cpu_baseline.py; the reference tree does not travel)
        r = json.load(open(os.path.join(REPO, "profiles", "r03_ref_cpu_task_ops.json")))
        ref_ops = {"env_steps_per_s": r["env_steps_per_s_task_ops_only"], "kind": "reference (its HumanoidSMPLIM reset / pre_physics_step / post_physics_step on "
                   "CPU tensors, physics replaced by a state copy)", "cores": r["threads"], "host": "build container (%s), NOT this box" % r["host"],
                   "envs": r["envs"], "ms_per_step": r["ms_per_step"], "source": "from_profiles: profiles/r03_ref_cpu_task_ops.json"}
    except Exception:
        pass
    small, big = by_n[str(sizes[0][0])], by_n[str(sizes[-1][0])]
    ratio = big["physics_env_steps_per_s_per_thread"] / small["physics_env_steps_per_s_per_thread"]
    return {"value": big["value"], "unit": "env-steps/s", "cores": threads, "kind": "port",
            "host": "%s, %d logical CPUs, %s: %d threads used" % (host_cpu_model(), logical, "no cgroup CPU quota" if quota is None else "cgroup CPU quota %.1f" % quota, threads),
            "by_num_envs": by_n, "reference_task_ops": ref_ops, "oracle_build_flags": FAST_FLAGS,
            "physics_env_steps_per_s": big["physics_env_steps_per_s"], "physics_env_steps_per_s_per_core": big["physics_env_steps_per_s_per_thread"],
            "task_ops_env_steps_per_s": big["task_ops_env_steps_per_s"], "task_ops_threads": big["task_ops_threads"],
            # a harness check, not a property of the CPU: the per-thread physics rate must survive the batch size
            "per_thread_rate_largest_over_smallest": ratio, "scaling_ok": bool(ratio >= 0.5),
            "sample": "num_envs %s: %s control steps each (4 substeps, contacts on): C float64 dense oracle (%s), one batched OpenMP call per step on %d threads "
                      "(= the CPUs the process may use at once: affinity capped by the cgroup quota) + numpy task ops sharded over the same threads; value = the %d-env "
                      "figure; NOT the reference's PhysX-CPU path (closed, absent)"
                      % ("/".join(str(x[0]) for x in sizes), "/".join(str(by_n[str(x[0])]["steps"]) for x in sizes), FAST_FLAGS, threads, sizes[-1][0])}


def profiles_view():
    """What the committed profiles say about the physics kernel: FLOPs and HBM bytes per launch (labelled from_profiles in the line).
    The counters describe ONE kernel: each file carries the hash of the kernel sources it was collected with
    (vid2player3d_amd.build.kernel_source_hash) and is DROPPED from the line - null + a warning on stderr - when the sources have moved on."""
    from vid2player3d_amd import build

    now = build.kernel_source_hash()

    def current(path, j):
        have = j.get("kernel_source_sha16")
        if have == now:
            return True
        sys.stderr.write("bench.py: %s was collected with kernel sources %s, the library is built from %s: its counters are NOT quoted (roofline.valu / traffic = null); "
                         "re-run tools/valu_probe.sh / tools/pmc_probe.sh on the GPU box\n" % (os.path.relpath(path, REPO), have or "(unstamped)", now))
        return False

    valu = traffic = None
    vc = os.path.join(REPO, "profiles", "valu_counters.json")
    if os.path.exists(vc):
        try:
            c = json.load(open(vc))
            if not current(vc, c):
                raise KeyError("stale")
            g = lambda k: c[k]["avg"]
            lanes = g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")
            wave_ops = 2.0 * g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_ADD_F32")
            valu = {"flops_per_launch": wave_ops * lanes, "lanes_active_per_valu_op": lanes,
                    "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
                    "wave_time_issuing_valu": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
                    # what bounds the kernel while its wave slots are full: cycles between two VALU instructions of a SIMD that holds three
                    # waves (SQ_WAVE_CYCLES counts quad-cycles of wave residency) against the issue ceiling of a saturated SIMD measured by
                    # tools/ubench/valu_issue.hip (profiles/r04_valu_issue.txt: 2.67 .. 2.88 nominal cycles per v_fma_f32, 2 .. 8 waves per SIMD)
                    # Two ceilings, both reported: the guide's issue rate the 157.3 TFLOP/s peak assumes (MI355X_MICROARCH.md: a wave64
                    # v_fma_f32 occupies a SIMD-32 for 2 cycles) and this box's measurement (see VALU_CEILING below)
                    "valu_issue": {"cycles_per_inst_per_simd_at_3_waves": 4.0 * g("SQ_WAVE_CYCLES") / 3.0 / g("SQ_INSTS_VALU"),
                                   "ceiling_cycles_per_inst_guide": 2.0, "frac_of_guide_ceiling": 2.0 / (4.0 * g("SQ_WAVE_CYCLES") / 3.0 / g("SQ_INSTS_VALU")),
                                   "ceiling_cycles_per_inst_measured": VALU_CEILING["cycles"], "ceiling_measured_note": VALU_CEILING["note"],
                                   "frac_of_measured_ceiling": VALU_CEILING["cycles"] / (4.0 * g("SQ_WAVE_CYCLES") / 3.0 / g("SQ_INSTS_VALU")),
                                   "frac": 2.0 / (4.0 * g("SQ_WAVE_CYCLES") / 3.0 / g("SQ_INSTS_VALU")), "frac_of": "the guide's 2-cycle wave64 issue (the conservative one)",
                                   "float_math_share_of_valu": (g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_TRANS_F32")) / g("SQ_INSTS_VALU")},
                    "kernel_clock_mhz": (c.get("clock_mhz") or {}).get("avg"),
                    "kernel_source_sha16": c.get("kernel_source_sha16"), "profile_git_head": c.get("git_head"),
                    "source": "from_profiles: " + c.get("source", "profiles/valu_counters.json")}
        except Exception:
            valu = None
    pmc = os.path.join(REPO, "profiles", "pmc_summary.json")
    if os.path.exists(pmc):
        try:
            p = json.load(open(pmc))
            if not current(pmc, p):
                raise KeyError("stale")
            traffic = {"kernel_source_sha16": p.get("kernel_source_sha16"), "profile_git_head": p.get("git_head"),
                       "bytes_per_launch": p.get("physics_kernel_hbm_bytes_calibrated", p.get("physics_kernel_hbm_bytes_per_launch")),
                       "bytes_per_launch_fetch_x2": p.get("physics_kernel_hbm_bytes_per_launch_fetch_x2"),
                       "note": p.get("calibration_note", "raw FETCH_SIZE + WRITE_SIZE; the guide's gfx950 correction doubles FETCH_SIZE for wide coalesced reads: traffic_fetch_x2 is that upper bound"),
                       "source": "from_profiles: " + p.get("source", "profiles/pmc_summary.json")}
        except Exception:
            traffic = None
    return valu, traffic


def run_ppo(args, task, dist, world, rank):
    """BASELINE config 5 (per GPU: --num-envs envs, horizon 32, the amass_im.yaml MLP and PPO hyper-parameters): epochs of
    play_steps + update; the reference's meters `fps step` = frames / T_play and `fps total` = frames / (T_play + T_update)
    (im_agent.py:204-214), whole job = sum over ranks of frames over the slowest rank's time."""
    tasks = task if isinstance(task, list) else [task]
    task = tasks[0]
    if args.stub_task:
        agent = StubAgent(task)
    else:
        from vid2player3d_amd.ppo import PPOAgent

        agent = PPOAgent(tasks if len(tasks) > 1 else task, seed=7, reuse_next_values=not args.ppo_reference_critic_passes, overlap_critic=not args.ppo_no_overlap, mixed_precision=args.ppo_mixed_precision)
    agent.train_epoch()  # warm-up epoch (allocator, rocBLAS heuristics, running statistics)
    rows = []
    for _ in range(args.ppo_epochs):
        r = agent.train_epoch()
        rows.append(r)
        if rank == 0:
            sys.stderr.write(agent.format_epoch_line(r) + "\n")
    play = sum(r["play_time"] for r in rows)
    total = sum(r["total_time"] for r in rows)
    frames = sum(r["frames"] for r in rows)
    if dist is not None:
        t = torch.tensor([play, total], device=task.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        play, total = float(t[0]), float(t[1])
    world_seen = 1 if dist is None else dist.get_world_size()
    rccl = None
    if dist is not None and not args.stub_task:
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = "unknown"
    if rank == 0:
        out = {"metric": "env-steps/sec at num_envs=%d, SMPL humanoid imitation" % args.num_envs, "value": world * frames / total, "unit": "env-steps/s", "n_gpus": world,
               "steps": args.ppo_epochs * HORIZON, "warmup": HORIZON, "ms_per_step": 1e3 * total / (args.ppo_epochs * HORIZON), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f16 autocast (update) / f32" if args.ppo_mixed_precision else "f32", "data": "synthetic",
               "config": {"workload": "FULL PPO LOOP (BASELINE config 5 shape, reported separately from the rollout metric): amass_im num_envs=%d per GPU, "
                                      "horizon 32, actor/critic MLP [1024,1024,512] on the 734-d observation (residual action), 6 mini-epochs x minibatches of 512 envs; "
                                      "%d rollout group(s) per GPU, critic %s%s; value = fps total"
                                      % (args.num_envs, len(tasks), "twice per step like the reference" if args.ppo_reference_critic_passes else "once per step (next_values reused as values)" + ("" if args.ppo_no_overlap else ", on a side stream beside the physics launch"),
                                         ", UPDATE IN MIXED PRECISION (fp16 autocast: the reference's non-default cfg option)" if args.ppo_mixed_precision else ""),
                          "num_envs_per_gpu": args.num_envs, "global_envs": world * args.num_envs,
                          "parallelism": "env-sharded x%d; advantage statistics, running norms and gradients all-reduced over RCCL at the update" % world,
                          "world_size_seen": world_seen, "world_size_matches_gpus": world_seen == args.gpus, "rccl_version": rccl,
                          "backend": None if dist is None else ("gloo" if args.stub_task else "nccl(rccl)"),
                          "scaling_curve": "no multi-GPU curve has been measured for this engine (the driver's 8-GPU runs were skipped in rounds 1-3)",
                          "fps_step": world * frames / play, "fps_total": world * frames / total,
                          "T_play_s_per_epoch": play / args.ppo_epochs, "T_update_s_per_epoch": (total - play) / args.ppo_epochs,
                          "rollout_groups": len(tasks),
                          "step_rewards_by_epoch": [r["step_rewards"] for r in rows], "step_sub_rewards_last_epoch": rows[-1]["step_sub_rewards"],
                          "step_rewards_last_epoch": rows[-1]["step_rewards"], "alive_ratio_last_epoch": rows[-1]["alive_ratio"]},
               }
        if args.stub_task:
            out["config"]["workload"] += ", STUB TASK AND AGENT (launch-logic test, not a measurement)"
        else:
            from vid2player3d_amd import build
            out["build"] = build.build_info()
            out["build"]["physics_kernel_build"] = task.kernel_build()
        line = json.dumps(out)
    else:
        line = None
    print_last(line, dist)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--num-envs", type=int, default=8192, help="envs per GPU")
    ap.add_argument("--no-contact", action="store_true", help="BASELINE config 2 (PD only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-events", type=int, default=8, help="HIP events around one physics launch in K of the timed steps (1 = every launch)")
    ap.add_argument("--actions-per-step", action="store_true", help="stand-in policy evaluated before every step (one elementwise kernel between the physics launches) instead of once per epoch from the context window")
    ap.add_argument("--action-noise", type=float, default=0.17, help="sigma of the stand-in policy (0.17 = SURVEY 8d; small values = tracking-quality actions, fewer falls)")
    ap.add_argument("--solver", choices=["pgs", "tgs"], default="pgs")
    ap.add_argument("--friction-frame", choices=["world", "velocity"], default="world", help="v2p_sim_cfg.friction_frame (ABI 14): tangent frame of the hull x ground friction rows; 'velocity' is a labelled variant")
    ap.add_argument("--freeze-terminated", action="store_true", help="opt-in engine feature: terminated envs are not simulated until the epoch reset (not reference behaviour)")
    ap.add_argument("--djokovic", action="store_true", help="BASELINE config 4 (djokovic_im.yaml: terminationHeadHeight -0.5, faster clips)")
    ap.add_argument("--job-mono", type=int, default=None, help="v2p_sim_cfg.job_mono_permille (tuning sweeps)")
    ap.add_argument("--job-lead", type=int, default=None, help="v2p_sim_cfg.job_lead: substeps of the first job of a cut pair (tuning sweeps)")
    ap.add_argument("--kernel-build", type=int, default=None, choices=(0, 1, 2), help="v2p_sim_cfg.kernel_build: 0 engine's choice by env count, 1 LDS-parked / 3 waves per SIMD, 2 registers / 2 waves (A/B)")
    ap.add_argument("--pair-mix", type=int, default=None, help="v2p_sim_cfg.pair_mix_permille (tuning sweeps)")
    ap.add_argument("--substep-jobs", type=int, default=1, help="1: physics launch cut into (substep, env pair) jobs (v2p_sim_cfg.substep_jobs); same results, finer load balancing")
    ap.add_argument("--joint-limits", type=int, default=None, choices=(0, 1),
                    help="enforce the MJCF joint ranges as limit rows (only the racket arm of --racket-ball has any; default: on with --racket-ball, else off)")
    ap.add_argument("--racket-ball", action="store_true", help="BASELINE config 4 as worded: racket welded to the wrist + free ball with drag / Magnus lift, ball-ground and ball-racket contacts (implies --djokovic)")
    ap.add_argument("--ball-body-contacts", type=int, default=1, choices=(0, 1), help="--racket-ball: ball x link-hull contacts (0: only ball x racket and ball x ground, for A/B)")
    ap.add_argument("--per-clip-shapes", action="store_true", help="one NON-UNIFORM body shape per clip (the reference's per-clip SMPL assets) instead of one shape for all envs")
    ap.add_argument("--num-shapes", type=int, default=64, help="--per-clip-shapes: clips = shapes (64; 2048 / 8192 = AMASS scale: env i simulates shape i %% num_shapes)")
    ap.add_argument("--ppo", action="store_true", help="BASELINE config 5 loop: device-resident rollout (play_steps) + GAE + PPO update per epoch; prints the reference's fps step / fps total")
    ap.add_argument("--ppo-epochs", type=int, default=4, help="timed PPO epochs (after one untimed warm-up epoch)")
    ap.add_argument("--groups", type=int, default=1, help="rollout groups per GPU: the rank's envs as G env batches on G HIP streams (reported separately from the headline; "
                    "with --ppo the physics of one group overlaps the policy inference of the other)")
    ap.add_argument("--ppo-mixed-precision", action="store_true", help="--ppo: the reference's cfg option mixed_precision (amass_im.yaml: False): fp16 autocast + loss scaling in the update; a labelled variant, never the headline")
    ap.add_argument("--ppo-no-overlap", action="store_true", help="--ppo: the critic pass on the rollout's own stream instead of a side stream beside the physics (A/B)")
    ap.add_argument("--ppo-reference-critic-passes", action="store_true", help="--ppo: evaluate the critic twice per step like the reference (A/B of reuse_next_values)")
    ap.add_argument("--stub-task", action="store_true", help=argparse.SUPPRESS)  # launch-logic test without GPUs (gloo, CPU); never a measurement
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.stub_task))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    stub = args.stub_task
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rollout engine has no CPU path")
    if not stub:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("V2P_BENCH_FORCE_DIST"):  # (the env var runs the collective path on a single GPU: CI of the N>1 code)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.num_envs
    G = max(1, args.groups)
    # (the engine picks the build of its physics kernel by the envs RESIDENT on the device - G batches that share the GPU are judged
    # together: 8192 envs as 2 x 4096 run the three-wave build, 20.9 M against 19.4 M with a per-batch choice, profiles/r04e_dual_build.txt)
    if n % G:
        raise SystemExit("--num-envs %d is not a multiple of --groups %d" % (n, G))
    if stub:
        task = StubTask(n)
        tasks = [task]
    else:
        from vid2player3d_amd import build
        if local_rank == 0:
            build.build()  # no-op when the in-tree .so is current; one rank per node compiles otherwise
        if dist is not None:
            dist.barrier()
        tasks = [build_task(n // G, local_rank, seed=7 + rank + 100 * g, contact=not args.no_contact, per_clip_shapes=args.per_clip_shapes, num_shapes=args.num_shapes, djokovic=args.djokovic or args.racket_ball,
                          freeze=args.freeze_terminated, solver=args.solver, racket_ball=args.racket_ball, substep_jobs=bool(args.substep_jobs),
                          joint_limits=args.joint_limits,
                          env_extra={k: v for k, v in (("job_mono_permille", args.job_mono), ("pair_mix_permille", args.pair_mix), ("job_lead", args.job_lead), ("kernel_build", args.kernel_build), ("ball_body_contacts", None if args.ball_body_contacts else False), ("friction_frame", None if args.friction_frame == "world" else args.friction_frame)) if v is not None})  # per-rank seed like run.py:37
                 for g in range(G)]
        task = tasks[0]
    if args.ppo:
        return run_ppo(args, tasks if G > 1 else task, dist, world, rank)
    dev = task.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)
    ng = n // G
    noise = [args.action_noise * torch.randn((n, 75), device=dev, generator=gen) for _ in range(HORIZON)]
    per_epoch = not args.actions_per_step and not stub and G == 1
    if per_epoch:
        noise_all = torch.stack(noise)
        epoch_actions = torch.empty_like(noise_all)
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)] if G > 1 else [None]

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    def run(nsteps):
        if G == 1:
            for i in range(nsteps):
                if i % HORIZON == 0:
                    task.reset()
                    if per_epoch:
                        make_epoch_actions(task, noise_all, epoch_actions)
                task.step_fused(epoch_actions[i % HORIZON] if per_epoch else make_actions(task, noise[i % HORIZON]))
            return
        # rollout groups: group g's envs on stream g; the launches of the groups interleave on the GPU (each fills the other's tail)
        main = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(main)
        for i in range(nsteps):
            for g, (tk, st) in enumerate(zip(tasks, streams)):
                with torch.cuda.stream(st):
                    if i % HORIZON == 0:
                        tk.reset()
                    tk.step_fused(make_actions(tk, noise[i % HORIZON][g * ng:(g + 1) * ng]))
        for st in streams:
            main.wait_stream(st)

    def timed_block(nsteps, ev_stride):
        """Time `nsteps` steps (each run() starts with the per-epoch reset: positions 0 .. nsteps-1 of the epoch, over and over) between two
        barrier + synchronize pairs; HIP events around one physics launch in ev_stride (recorded by the engine on the launch stream, the
        bracketed position rotating through the epoch).  Returns (seconds of this rank, mean kernel ms, launches bracketed)."""
        barrier()
        for tk in tasks:
            if stub:
                tk.profile_begin(nsteps)
            else:
                tk.profile_begin(nsteps, stride=ev_stride, period=HORIZON)
        t0 = time.perf_counter()
        run(nsteps)
        barrier()
        el = time.perf_counter() - t0
        ms_total, cnt = 0.0, 0
        for tk in tasks:
            a_, b_ = tk.profile_end()
            ms_total, cnt = ms_total + a_, cnt + b_
            if hasattr(tk, "check"):
                tk.check()  # device-side errors (a substep job that timed out) fail the run instead of producing a number
        return el, ms_total / max(cnt, 1), cnt

    def over_ranks(x):
        """[value of every rank] (all-gather; one entry without a process group)."""
        if dist is None:
            return [x]
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, t)
        return [float(v.item()) for v in allt]

    # (short runs bracket more launches, at least five: 20 steps -> one in four)
    ev_stride = 1 if stub else max(1, min(args.kernel_events, args.steps // 5))
    run(args.warmup)
    elapsed_local, phys_ms, launches = timed_block(args.steps, ev_stride)
    times = over_ranks(elapsed_local)
    elapsed = max(times)  # MAX over ranks
    per_rank = [n * args.steps / x for x in times]
    per_rank_kernel_ms = over_ranks(phys_ms)
    alive = float(torch.cat([(tk.reset_buf == 0).float() for tk in tasks]).mean().item())
    # Every run() starts an epoch (reset + positions 0, 1, ...), and the early positions of an epoch - standing humanoids - are lighter
    # than the late ones (profiles/*_epoch_profile.txt: 0.32 ms at step 3, 0.61 ms at step 31).  A timed region that is not a whole
    # number of epochs is therefore NOT the rollout average: such runs (the driver's --steps 20) are followed by a separately timed
    # block of WHOLE_EPOCHS whole epochs, reported as `whole_epoch`; the roofline fractions are taken from that block.
    whole = None
    if args.steps % HORIZON or args.steps < 2 * HORIZON:
        wsteps = WHOLE_EPOCHS * HORIZON
        w_el, w_ms, w_cnt = timed_block(wsteps, max(1, args.kernel_events))
        w_el = max(over_ranks(w_el))
        whole = {"steps": wsteps, "epochs": WHOLE_EPOCHS, "value": world * n * wsteps / w_el, "ms_per_step": 1e3 * w_el / wsteps, "kernel_ms": w_ms,
                 "kernel_launches_timed": w_cnt, "per_rank_kernel_ms": over_ranks(w_ms)}
    world_seen = 1 if dist is None else dist.get_world_size()

    if rank == 0:
        req_value = world * n * args.steps / elapsed
        # the headline is the ROLLOUT AVERAGE: the requested region when it is whole epochs, else the whole-epoch block timed right after it
        # (barrier + synchronize on both sides, MAX over ranks, like the requested region, which stays in the line as `requested_region`)
        value, ms_per_step = (whole["value"], whole["ms_per_step"]) if whole else (req_value, 1e3 * elapsed / args.steps)
        if whole:
            sys.stderr.write("bench.py: --steps %d is not a whole number of %d-step epochs (positions 0..%d of an epoch are the light ones: %.2f M env-steps/s); "
                             "value = the %d-epoch block timed after it (%.2f M)\n" % (args.steps, HORIZON, args.steps - 1, req_value / 1e6, WHOLE_EPOCHS, value / 1e6))
        nresets = (args.steps + HORIZON - 1) // HORIZON
        valu, traffic = profiles_view()
        # the fractions describe the rollout average: from the whole-epoch block when the requested region is not whole epochs
        k_ms, k_cnt, k_stride = (whole["kernel_ms"], whole["kernel_launches_timed"], max(1, args.kernel_events)) if whole else (phys_ms, launches, ev_stride)
        achieved = ALGO_BYTES_PER_ENV_STEP * ng / (k_ms * 1e-3) / 1e9
        roof = {"bound": "valu-issue", "bound_note": "the contract's choices are hbm | mfma; this kernel is bound by neither: while its wave slots are full its SIMDs issue one VALU "
                "instruction per ~3.4 cycles against a measured ceiling of ~2.7 (valu.valu_issue), at ~24 of 64 lanes active; late in an epoch a launch is as long as "
                "its heaviest env pair (DESIGN.md 4-5).  achieved / peak / frac are the HBM figures BASELINE.json asks for; valu_frac says how good the kernel is",
                "kernel": "physics_ll_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of": "hbm",
                "traffic": None if traffic is None else traffic["bytes_per_launch"],
                "traffic_fetch_x2": None if traffic is None else traffic.get("bytes_per_launch_fetch_x2"),
                "traffic_note": None if traffic is None else traffic.get("note"),
                "traffic_source": None if traffic is None else traffic["source"],
                "traffic_kernel_source_sha16": None if traffic is None else traffic["kernel_source_sha16"], "kernel_ms": k_ms, "kernel_launches_timed": k_cnt,
                "kernel_ms_region": "whole_epoch block (%d epochs)" % WHOLE_EPOCHS if whole else "the timed steps (whole epochs)",
                "kernel_ms_source": "HIP events recorded by the engine around %s (launch stream)%s" % (
                    "every physics launch" if k_stride == 1 else "one physics launch in %d" % k_stride,
                    "" if k_stride == 1 else "; the bracketed step rotates through the positions of the epoch (two event records cost ~8 us of dispatch per bracketed launch: --kernel-events 1 brackets all, -2 % throughput)"),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * ng}
        if whole:
            roof["requested_region"] = {"kernel_ms": phys_ms, "kernel_launches_timed": launches, "events_around_one_launch_in": ev_stride,
                                        "frac": ALGO_BYTES_PER_ENV_STEP * ng / (phys_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if launches else None}
            whole["frac"] = roof["frac"]
        if G > 1:
            roof["bound_note"] += "; %d ROLLOUT GROUPS: the launches of the groups overlap on the GPU, kernel_ms is the duration of one group's launch WHILE the others run" % G
        if valu is not None and n == 8192 and G == 1 and not args.no_contact:
            tflops = valu["flops_per_launch"] / (k_ms * 1e-3) / 1e12
            roof.update({"valu_tflops": tflops, "valu_peak_tflops": FP32_VECTOR_PEAK_TFLOPS, "valu_frac": tflops / FP32_VECTOR_PEAK_TFLOPS, "valu": valu})
            if whole:
                whole["valu_frac"] = roof["valu_frac"]
        rccl = None
        if dist is not None and not stub:
            try:
                rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                rccl = "unknown"
        out = {
            "metric": "env-steps/sec at num_envs=%d, SMPL humanoid imitation" % n, "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "value_region": "the %d timed steps (whole epochs)" % args.steps if not whole else
                            "whole_epoch block: %d steps = %d whole epochs timed right after the %d requested steps, same barriers (a region that is not whole epochs "
                            "times the light start of an epoch: requested_region)" % (whole["steps"], WHOLE_EPOCHS, args.steps),
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "amass_im num_envs=%d per GPU, %s, imitation reward, per-epoch reset+context every %d steps, %s synthetic clips, action noise %.3g%s"
                                   % (n, "PD control only (no contact solve)" if args.no_contact else "full contact %s (4 substeps x 4 iterations)" % args.solver.upper(), HORIZON,
                                      args.num_shapes if args.per_clip_shapes else 64, args.action_noise, (", one NON-UNIFORM body shape per clip (%d shapes from vertex clouds, %d clips)" % (args.num_shapes, args.num_shapes) if args.per_clip_shapes else "") + (", djokovic_im variant" if args.djokovic or args.racket_ball else "") + (", RACKET + BALL in every env (reported separately)" if args.racket_ball else "") + (", joint limits on" if (args.joint_limits if args.joint_limits is not None else args.racket_ball) else "") + (", VELOCITY-ALIGNED FRICTION FRAME (variant)" if args.friction_frame == "velocity" else "") +
                                      (", %d ROLLOUT GROUPS of %d envs on %d streams (reported separately from the headline)" % (G, ng, G) if G > 1 else "") +
                                      (", stand-in policy evaluated once per epoch (targets = context frames)" if per_epoch else ", stand-in policy evaluated before every step") + (", TERMINATED ENVS FROZEN (not reference behaviour)" if args.freeze_terminated else "") + (", STUB TASK (launch-logic test, not a measurement)" if stub else "")),
                       "num_envs_per_gpu": n, "global_envs": world * n, "parallelism": "env-sharded x%d, no data-path collective" % world,
                       "world_size_seen": world_seen, "world_size_matches_gpus": world_seen == args.gpus, "backend": None if dist is None else ("gloo" if stub else "nccl(rccl)"),
                       "rccl_version": rccl, "per_rank_env_steps_per_s": per_rank, "per_rank_kernel_ms": per_rank_kernel_ms,
                       "alive_fraction_at_end": alive, "substep_jobs": bool(args.substep_jobs),
                       # (early steps of an epoch - standing humanoids - are lighter than late ones: only whole epochs average like a rollout does)
                       "timed_steps_cover_whole_epochs": args.steps % HORIZON == 0,
                       "timed_epoch_positions": "all, %d times" % (args.steps // HORIZON) if args.steps % HORIZON == 0 else
                                                ("0..%d" % (args.steps - 1) if args.steps < HORIZON else "all %d times + 0..%d" % (args.steps // HORIZON, args.steps % HORIZON - 1)),
                       "resets_in_timed_region": nresets,
                       "scaling_curve": "no multi-GPU curve has been measured for this engine (the driver's 8-GPU runs were skipped in every round so far: "
                                        "more than one RCCL rank has never run)"},
            "roofline": roof,
        }
        if whole:
            out["whole_epoch"] = whole
            out["requested_region"] = {"steps": args.steps, "value": req_value, "ms_per_step": 1e3 * elapsed / args.steps, "epoch_positions": out["config"]["timed_epoch_positions"]}
        if not stub:
            from vid2player3d_amd import build
            out["build"] = build.build_info()
            out["build"]["physics_kernel_build"] = task.kernel_build()
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline()
        line = json.dumps(out)
    else:
        line = None
    print_last(line, dist)


if __name__ == "__main__":
    main()
