"""Parity of the HIP sampler / reward / reset / obs kernels and of the task state machine, through the C ABI.

Checked against (a) golden vectors recorded from the reference's own Python (tests/golden) and (b) the numpy
oracle on larger seeded inputs.  Tolerances are float32 op-order noise: 5e-6 relative for gathers / blends,
2e-4 for the rotation-angle reward term (acos near 1 is ill-conditioned in float32 in the reference itself)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import task_oracle as O
from tests.gpu_util import DEV, N, T, close, golden_motion_lib, make_task, synth_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mlib(golden_tables):
    return golden_motion_lib(golden_tables)


@pytest.mark.parametrize("adjust", [True, False])
def test_motion_state_matches_reference_golden(mlib, golden_motion_state, adjust):
    g = golden_motion_state
    res = mlib.get_motion_state(T(g["ids"], torch.long), T(g["times"]), return_rigid_body=True, adjust_height=adjust, ground_tolerance=0.0)
    sfx = "" if adjust else "_noadj"
    for name, r in zip(O.MOTION_STATE_NAMES, res):
        close(N(r), g[name + sfx], 5e-6, name + sfx)


def test_motion_state_large_vs_oracle():
    from vid2player3d_amd.motion_lib import MotionLib

    tabs = synth_tables(seed=11, num_clips=24)
    lib = MotionLib(tabs, DEV)
    rng = np.random.default_rng(0)
    q = 20000
    ids = rng.integers(0, 24, size=q)
    times = (rng.uniform(-0.05, 1.1, size=q) * tabs["motion_lengths"][ids]).astype(np.float32)
    res = lib.get_motion_state(T(ids, torch.long), T(times), return_rigid_body=True, adjust_height=True, ground_tolerance=0.01)
    ref = O.get_motion_state(tabs, ids, times, True, 0.01)
    for name, r, o in zip(O.MOTION_STATE_NAMES, res, ref):
        close(N(r), o, 5e-6, name)
    # without rigid bodies the first seven outputs are unchanged
    res7 = lib.get_motion_state(T(ids, torch.long), T(times), return_rigid_body=False, adjust_height=True, ground_tolerance=0.01)
    assert len(res7) == 7
    for a, b in zip(res7, res):
        assert torch.equal(a, b)


def test_motion_state_empty_query(mlib):
    res = mlib.get_motion_state(torch.zeros(0, dtype=torch.long, device=DEV), torch.zeros(0, device=DEV), return_rigid_body=True)
    assert [tuple(r.shape) for r in res][:3] == [(0, 3), (0, 4), (0, 69)]


def _reward(lib, g, n):
    from vid2player3d_amd import _lib

    rew = torch.empty(n, device=DEV)
    sub = torch.empty((n, 4), device=DEV)
    w = (C.c_float * 24)(*g["body_pos_weights"].tolist())
    s = (C.c_float * 8)(60, 0.2, 100, 40, 0.6, 0.1, 0.2, 0.1)
    args = [T(g[k][:n]) for k in ("body_pos", "body_rot", "tgt_pos", "tgt_rot", "dof_pos", "dof_vel", "tgt_dof_pos", "tgt_dof_vel")]
    _lib.check(lib.v2p_reward(n, *[_lib.ptr(a) for a in args], w, s, _lib.ptr(rew), _lib.ptr(sub), None), "v2p_reward")
    torch.cuda.synchronize()
    return N(rew), N(sub)


def test_reward_matches_reference_golden(golden_task_ops):
    from vid2player3d_amd import _lib

    g = golden_task_ops
    rew, sub = _reward(_lib.load(), g, g["reward"].shape[0])
    close(sub[:, :3], g["sub_rewards"][:, :3], 5e-6, "sub[dof,vel,pos]")
    close(sub[:, 3], g["sub_rewards"][:, 3], 2e-4, "sub[rot]")
    close(rew, g["reward"], 5e-5, "reward")


def test_reset_flags_match_reference_golden(golden_task_ops):
    from vid2player3d_amd import _lib

    g = golden_task_ops
    lib = _lib.load()
    n = g["reset_out"].shape[0]
    h = g["reset_heights"].astype(np.float32).copy()
    h[[7, 3]] = -np.inf
    rst = torch.empty(n, dtype=torch.long, device=DEV)
    term = torch.empty(n, dtype=torch.long, device=DEV)
    prog, rb, ct, cl = T(g["reset_progress"], torch.long), T(g["reset_rb_pos"]), T(g["reset_cur_time"]), T(g["reset_clip_len"])  # keep alive
    _lib.check(lib.v2p_reset_flags(n, _lib.ptr(prog), _lib.ptr(rb), (C.c_float * 24)(*h.tolist()), _lib.ptr(ct), _lib.ptr(cl), 300.0, 1,
                                   _lib.ptr(rst), _lib.ptr(term), None), "v2p_reset_flags")
    assert np.array_equal(N(rst), g["reset_out"])
    assert np.array_equal(N(term), g["terminate_out"])


def test_obs_imitation_734_matches_reference_golden(golden_task_ops):
    from vid2player3d_amd import _lib

    g = golden_task_ops
    lib = _lib.load()
    n = g["obs734"].shape[0]
    obs = torch.empty((n, 734), device=DEV)
    args = [T(g[k]) for k in ("body_pos", "body_rot", "tgt_pos", "tgt_rot", "dof_pos", "dof_vel", "tgt_dof_pos", "body_vel", "body_ang_vel",
                              "obs734_motion_bodies")]
    _lib.check(lib.v2p_obs_imitation(n, *[_lib.ptr(a) for a in args], None, None, 0.0, _lib.ptr(obs), None), "v2p_obs_imitation")
    close(N(obs), g["obs734"], 5e-6, "obs734")
    # with RunningNorm (eval) fused
    mean, std = T(g["rn_mean"]), T(g["rn_std"])
    _lib.check(lib.v2p_obs_imitation(n, *[_lib.ptr(a) for a in args], _lib.ptr(mean), _lib.ptr(std), 5.0, _lib.ptr(obs), None), "v2p_obs_imitation")
    close(N(obs), g["obs734_normed"], 2e-5, "obs734 normalised")


def test_gae_matches_reference_golden(golden_task_ops):
    from vid2player3d_amd import _lib

    g = golden_task_ops
    lib = _lib.load()
    t, n = g["gae_fdones"].shape
    fd, va, rw, nv = T(g["gae_fdones"]), T(g["gae_values"]), T(g["gae_rewards"]), T(g["gae_next_values"])
    adv = torch.empty((t, n, 1), device=DEV)
    _lib.check(lib.v2p_gae(t, n, _lib.ptr(fd), _lib.ptr(va), _lib.ptr(rw), _lib.ptr(nv), float(g["gae_gamma"]), float(g["gae_tau"]), _lib.ptr(adv), None),
               "v2p_gae")
    close(N(adv), g["gae_advs"], 2e-6, "gae")
    # full rollout size: 32 x 8192, against the numpy restatement
    rng = np.random.default_rng(1)
    t, n = 32, 8192
    fd_h = (rng.uniform(size=(t, n)) < 0.05).astype(np.float32)
    va_h, rw_h, nv_h = (rng.normal(size=(t, n, 1)).astype(np.float32) for _ in range(3))
    fd, va, rw, nv = T(fd_h), T(va_h), T(rw_h), T(nv_h)
    adv = torch.empty((t, n, 1), device=DEV)
    _lib.check(lib.v2p_gae(t, n, _lib.ptr(fd), _lib.ptr(va), _lib.ptr(rw), _lib.ptr(nv), 0.99, 0.95, _lib.ptr(adv), None), "v2p_gae")
    close(N(adv), O.discount_values(fd_h, va_h, rw_h, nv_h, 0.99, 0.95), 1e-5, "gae 32x8192")


def test_env_trace_replay_matches_reference(mlib, golden_env_trace):
    """reset / pre_physics_step / post_physics_step against the trace recorded from the reference's own
    HumanoidSMPLIM methods, with the recorded 'simulated' states teacher-forced in place of physics."""
    g = golden_env_trace
    task = make_task(6, mlib, motion_ids=g["motion_ids"], record_pd_torque=True)
    for tag, nsteps in (("e0_", int(g["num_steps"])), ("e1_", int(g["num_steps_e1"]))):
        task.reset_with_times(None, T(g[tag + "reset_motion_times"]))
        close(N(task._humanoid_root_states), g[tag + "reset_root_states"], 5e-6, tag + "root_states")
        close(N(task._dof_pos), g[tag + "reset_dof_pos"], 5e-6, tag + "dof_pos")
        close(N(task._dof_vel), g[tag + "reset_dof_vel"], 5e-6, tag + "dof_vel")
        close(N(task._rigid_body_state).reshape(6, 24, 13), g[tag + "reset_rb_state"], 5e-6, tag + "rb_state")
        close(N(task.context_feat), g[tag + "context_feat"], 5e-6, tag + "context_feat")
        assert np.array_equal(N(task.context_mask), g[tag + "context_mask"])
        close(N(task.obs_buf), g[tag + "reset_obs"], 5e-6, tag + "reset_obs")
        assert not N(task.reset_buf).any() and not N(task._terminate_buf).any() and not N(task.progress_buf).any()
        for i in range(nsteps):
            p = "%ss%02d_" % (tag, i)
            a = T(g[p + "actions"])
            task.pre_physics_step(a)
            close(N(a), g[p + "actions_after"], 0.0, p + "actions masked in place")
            close(N(task.pd_torque), g[p + "pd_torque"], 2e-6, p + "pd_torque")
            # teacher-forced physics: the recorded state goes in through the state-tensor views
            task._dof_pos[:] = T(g[p + "sim_dof_pos"])
            task._dof_vel[:] = T(g[p + "sim_dof_vel"])
            task._rigid_body_state.view(6, 24, 13)[:] = T(g[p + "sim_rb_state"])
            task._humanoid_root_states[:] = T(g[p + "sim_rb_state"][:, 0, :])
            task._reset_env_tensors(None, with_rb_state=True)
            task.post_physics_step()
            close(N(task.obs_buf), g[p + "obs"], 5e-6, p + "obs")
            close(N(task.rew_buf), g[p + "rew"], 1e-4, p + "rew")
            close(N(task.extras["sub_rewards"]), g[p + "sub_rewards"], 5e-4, p + "sub_rewards")
            assert np.array_equal(N(task.reset_buf), g[p + "reset"]), p
            assert np.array_equal(N(task.extras["terminate"]), g[p + "terminate"]), p
            assert np.array_equal(N(task.progress_buf), g[p + "progress"]), p
            close(N(task._cur_ref_motion_times), g[p + "cur_time"], 1e-6, p + "cur_time")
            for name in O.MOTION_STATE_NAMES:
                close(N(getattr(task, "_target_" + name)), g[p + "target_" + name], 5e-6, p + "target_" + name)
    task.close()


def test_partial_reset_only_touches_selected_envs(mlib):
    task = make_task(8, mlib)
    task.reset_with_times(None, torch.full((8,), 0.2, device=DEV))
    before = task.obs_buf.clone()
    ids = torch.tensor([1, 6], device=DEV)
    task.reset_with_times(ids, torch.tensor([0.5, 0.7], device=DEV))
    after = task.obs_buf
    keep = [0, 2, 3, 4, 5, 7]
    assert torch.equal(after[keep], before[keep])
    assert not torch.equal(after[1], before[1]) and not torch.equal(after[6], before[6])
    assert torch.allclose(task._cur_ref_motion_times[ids], torch.tensor([0.5, 0.7], device=DEV))
    task.close()


def test_termination_head_height_is_configurable(mlib):
    """djokovic_im differs from amass_im by terminationHeadHeight (-0.5 instead of 1.0, cfg/djokovic_im.yaml:21): with the head
    forced below 1 m the amass config terminates, the djokovic config does not."""
    res = {}
    for name, hh in (("amass", 1.0), ("djokovic", -0.5)):
        task = make_task(4, mlib, terminationHeadHeight=hh)
        task.reset_with_times(None, torch.full((4,), 0.2, device=DEV))
        a = torch.cat([task._target_dof_pos.clone(), torch.zeros((4, 6), device=DEV)], dim=1).contiguous()
        for _ in range(3):
            task.pre_physics_step(a.clone())
            task._rigid_body_state.view(4, 24, 13)[:, 13, 2] = 0.7   # head at 0.7 m (teacher-forced)
            task.post_physics_step()
        res[name] = (N(task.reset_buf).copy(), N(task._terminate_buf).copy())
        task.close()
    assert res["amass"][1].all() and res["amass"][0].all()
    assert not res["djokovic"][1].any() and not res["djokovic"][0].any()


def test_legacy_pth_motion_file_drives_the_task():
    """cfg['env']['motion_file'] = the reference's directory of pickled MotionLib parts: loaded without the reference's classes,
    sampled on the GPU, equal to the reference's own get_motion_state on the merged library."""
    import os

    from tests.conftest import GOLDEN
    from vid2player3d_amd.tasks import HumanoidSMPLIM, default_cfg

    cfg = default_cfg(6, motion_file=os.path.join(GOLDEN, "legacy_mlib"))
    cfg["env"]["sample_first_motions"] = True
    # the fixture's three clips carry three different betas: one baked body for all of them is refused unless the caller says so
    with pytest.raises(RuntimeError, match="different body shapes"):
        HumanoidSMPLIM(cfg, device_type="cuda", device_id=0)
    cfg["env"]["body_shape_mismatch"] = "ignore"
    task = HumanoidSMPLIM(cfg, device_type="cuda", device_id=0)
    assert task._motion_lib.num_motions() == 3
    with pytest.raises(RuntimeError, match="handed to an env batch"):
        task._motion_lib.merge_multiple_motion_libs([])
    bad = default_cfg(6, motion_lib=task._motion_lib, body_shape_mismatch="ignore", motion_ids=[0, 1, 2, 3, 0, 1])
    with pytest.raises(ValueError, match="motion ids"):
        HumanoidSMPLIM(bad, device_type="cuda", device_id=0)
    with np.load(os.path.join(GOLDEN, "legacy_mlib_expected.npz")) as z:
        exp = {k: z[k] for k in z.files}
    res = task._motion_lib.get_motion_state(T(exp["state_ids"], torch.long), T(exp["state_times"]), return_rigid_body=True, adjust_height=True,
                                            ground_tolerance=0.0)
    for name, r in zip(O.MOTION_STATE_NAMES, res):
        close(N(r), exp["state_" + name], 5e-6, name)
    task.reset()
    task.step(torch.zeros((6, 75), device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(task.obs_buf).all()
    task.close()


def _packed_inputs(g, rows, pad=8, length=32, t=5):
    """The golden 734-d inputs packed the way the network receives them: 461-d obs rows + 48 context frames per env."""
    obs = np.concatenate([g["body_pos"][:rows].reshape(rows, 72), g["body_rot"][:rows].reshape(rows, 96), g["dof_pos"][:rows], g["dof_vel"][:rows],
                          g["body_vel"][:rows].reshape(rows, 72), g["body_ang_vel"][:rows].reshape(rows, 72), g["obs734_motion_bodies"][:rows]], axis=1)
    assert obs.shape[1] == 461
    frame = np.concatenate([g["tgt_pos"][:rows].reshape(rows, 72), g["tgt_rot"][:rows].reshape(rows, 96), g["tgt_dof_pos"][:rows],
                            np.zeros((rows, 141), np.float32)], axis=1)
    return obs.astype(np.float32), frame.astype(np.float32)


def test_obs_imitation_from_packed_obs_and_context(golden_task_ops):
    """learning.ImitationObs = preprocess_input + compute_humanoid_obs + running_obs of the reference network, on the packed inputs."""
    from vid2player3d_amd.learning import ImitationObs

    g = golden_task_ops
    n = g["obs734"].shape[0]
    obs, frame = _packed_inputs(g, n)
    rng = np.random.default_rng(0)
    # rollout flavour: env e, step t=5 -> context frame pad + 5; the other frames hold garbage
    ctx = rng.normal(size=(n, 48, 378)).astype(np.float32)
    ctx[:, 8 + 5] = frame
    enc = ImitationObs(context_padding=8)
    close(N(enc.rollout(T(obs), T(ctx), 5)), g["obs734"], 5e-6, "packed rollout")
    enc = ImitationObs(8, T(g["rn_mean"]), T(g["rn_std"]), 5.0)
    close(N(enc.rollout(T(obs), T(ctx), 5)), g["obs734_normed"], 2e-5, "packed rollout normalised")
    # training flavour: n = envs * steps rows, row e*steps + k pairs with frame pad + k of env e
    steps = 4
    envs = n // steps
    rows = envs * steps
    ctx = rng.normal(size=(envs, 48, 378)).astype(np.float32)
    ctx[:, 8:8 + steps] = frame[:rows].reshape(envs, steps, 378)
    out = ImitationObs(8).training(T(obs[:rows].reshape(envs, steps, 461)), T(ctx))
    close(N(out), g["obs734"][:rows], 5e-6, "packed training")
    with pytest.raises(RuntimeError):
        ImitationObs(8).rollout(T(obs), T(ctx[:, :, :100]), 0)


def test_discount_values_wrapper(golden_task_ops):
    from vid2player3d_amd.learning import discount_values

    g = golden_task_ops
    from vid2player3d_amd.learning import DiscountValuesMixin

    args = (T(g["gae_fdones"]), T(g["gae_values"]), T(g["gae_rewards"]), T(g["gae_next_values"]))
    adv = discount_values(*args, float(g["gae_gamma"]), float(g["gae_tau"]))
    close(N(adv), g["gae_advs"], 2e-6, "discount_values")

    class Agent(DiscountValuesMixin):  # the reference's 4-argument method signature (common_agent.py:423)
        gamma, tau = float(g["gae_gamma"]), float(g["gae_tau"])

    assert torch.equal(Agent().discount_values(*args), adv)


def test_imitation_obs_running_norm_semantics(golden_task_ops):
    """RunningNorm.forward (models/running_norm.py:32-43): a fresh model (n == 0) does not normalise, clip None / 0 does not clamp,
    statistics that still live on the CPU are moved to the observations' device instead of reaching the kernel as host pointers."""
    from vid2player3d_amd.learning import ImitationObs

    g = golden_task_ops
    n = g["obs734"].shape[0]
    obs, frame = _packed_inputs(g, n)
    ctx = np.zeros((n, 48, 378), np.float32)
    ctx[:, 8 + 3] = frame
    raw = ImitationObs(8).rollout(T(obs), T(ctx), 3)

    class RN:  # duck-typed RunningNorm
        def __init__(self, n, clip):
            self.n, self.clip = torch.tensor(n), clip
            self.mean, self.std = torch.full((734,), 0.25), torch.full((734,), 2.0)  # CPU buffers on purpose
            self.demean = self.destd = True

    assert torch.equal(ImitationObs.from_running_norm(RN(0, 5.0)).rollout(T(obs), T(ctx), 3), raw)
    unclamped = ImitationObs.from_running_norm(RN(10, None)).rollout(T(obs), T(ctx), 3)
    close(N(unclamped), (N(raw) - 0.25) / (2.0 + 1e-8), 2e-6, "no clamp")
    clamped = ImitationObs.from_running_norm(RN(10, 0.5)).rollout(T(obs), T(ctx), 3)
    close(N(clamped), np.clip((N(raw) - 0.25) / (2.0 + 1e-8), -0.5, 0.5), 2e-6, "clamp 0.5")
    with pytest.raises(ValueError):
        ImitationObs(8, torch.zeros(10), torch.ones(10))


def test_imitation_obs_training_mode_updates_the_statistics(golden_task_ops):
    """RunningNorm in training mode (models/running_norm.py:32-35): the statistics take the batch's raw in-network features in first,
    then normalise that very batch; a second batch merges by count.  The kernel supplies the raw features."""
    from vid2player3d_amd.learning import ImitationObs, RunningNorm

    g = golden_task_ops
    steps = 4
    envs = g["obs734"].shape[0] // steps
    rows = envs * steps
    obs, frame = _packed_inputs(g, rows)
    ctx = np.zeros((envs, 48, 378), np.float32)
    ctx[:, 8:8 + steps] = frame[:rows].reshape(envs, steps, 378)
    enc = ImitationObs(8)
    o, c = T(obs[:rows].reshape(envs, steps, 461)), T(ctx)
    raw = enc.training(o, c)
    rn = RunningNorm(734, device=DEV)
    y = enc.training(o, c, running_norm=rn)
    assert int(rn.n) == rows
    var, mean = torch.var_mean(raw, dim=0, unbiased=False)
    close(N(rn.mean), N(mean), 1e-6, "mean after the first batch")
    close(N(rn.var), N(var), 1e-6, "var after the first batch")
    close(N(y), N(torch.clamp((raw - mean) / (torch.sqrt(var) + 1e-8), -5.0, 5.0)), 1e-5, "training-mode output")
    enc.training(o, c, running_norm=rn)
    assert int(rn.n) == 2 * rows
    close(N(rn.mean), N(mean), 1e-5, "the same batch again leaves the mean")
    rn.eval()
    y_eval = enc.training(o, c, running_norm=rn)
    assert int(rn.n) == 2 * rows, "eval mode leaves the statistics alone"
    fused = ImitationObs.from_running_norm(rn, 8).training(o, c)
    close(N(fused), N(y_eval), 2e-5, "eval mode = the fused kernel with the same statistics")


def test_replay_tool_on_the_golden_trace(tmp_path, golden_tables, capsys, monkeypatch):
    """tools/replay_trace.py (the procedure that pins physics parity on a box with Isaac Gym) runs end to end on the golden trace:
    teacher-forced task ops within float32 rounding of the recording."""
    import importlib.util
    import os
    import sys

    from tests.conftest import GOLDEN, REPO

    keys = ("gts", "grs", "lrs", "grvs", "gravs", "dvs", "motion_lengths", "motion_num_frames", "motion_dt", "motion_fps", "motion_weights",
            "motion_bodies", "motion_min_verts_h")
    flat = tmp_path / "mlib.npz"
    np.savez(str(flat), **{k: golden_tables[k] for k in keys})
    spec = importlib.util.spec_from_file_location("replay_trace", os.path.join(REPO, "tools", "replay_trace.py"))
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.chdir(REPO)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["replay_trace.py", os.path.join(GOLDEN, "env_trace.npz"), "--motion", str(flat)])
    mod.main()
    out = capsys.readouterr().out
    vals = {line[2:50].strip(): float(line[50:]) for line in out.splitlines() if line.startswith("  ")}
    assert vals["obs"] <= 5e-6 and vals["reward"] <= 1e-4 and vals["reset flags"] == 0 and vals["terminate flags"] == 0
    assert vals["target rb_pos"] <= 5e-6 and vals["actions masked in place"] == 0


def test_kernel_time_from_events_around_all_or_a_sample_of_the_launches():
    """v2p_env_profile_begin / _begin_sampled / _end (bench.py's roofline.kernel_ms): launch L is bracketed when
    L % stride == (L // period) % stride"""
    import torch

    from tests.gpu_util import DEV, make_task, synth_tables
    from vid2player3d_amd.motion_lib import MotionLib

    task = make_task(64, MotionLib(synth_tables(), DEV))
    task.reset()
    act = torch.zeros((64, 75), device=DEV)
    task.profile_begin(100)
    for _ in range(5):
        task.step(act)
    ms_all, n_all = task.profile_end()
    assert n_all == 5 and 0.0 < ms_all < 1e3
    task.profile_begin(100, stride=4, period=8)
    for _ in range(32):  # launches 0, 4 | 9, 13 | 18, 22 | 27, 31: two per 8-step epoch, the position moving on by one every epoch
        task.step(act)
    ms, n = task.profile_end()
    assert n == 8 and 0.0 < ms < 1e3
    task.profile_begin(3, stride=2, period=1000)  # the cap still holds
    for _ in range(20):
        task.step(act)
    assert task.profile_end()[1] == 3
    with pytest.raises(RuntimeError):
        task.profile_begin(10, stride=0)
    task.close()


def test_context_frames_are_the_targets_of_the_steps():
    """What bench.py's per-epoch stand-in policy (and the reference's residual action, im_network_builder.py:226-228) relies on: frame
    context_padding + k of the window built by the reset holds the target DOF positions of step k (same clip, same time up to the
    rounding of the accumulated clock)."""
    import torch

    from tests.gpu_util import DEV, make_task, synth_tables
    from vid2player3d_amd.motion_lib import MotionLib

    task = make_task(128, MotionLib(synth_tables(), DEV))
    torch.manual_seed(0)
    task.reset()
    pad, worst = task.context_padding, 0.0
    for k in range(task.context_length):
        worst = max(worst, float((task._target_dof_pos - task.context_feat[:, pad + k, 168:237]).abs().max()))
        task.step((0.1 * torch.randn(128, 75, device=DEV)).contiguous())
    assert worst < 1e-5, worst
    task.close()


def test_test_mode_starts_every_clip_at_its_first_frame():
    """`run.py --test` (humanoid_smpl_im.py:78-81): cfg['args'].test switches the state init to Start (and to the test clips, when named)"""
    import types

    import torch

    from tests.gpu_util import DEV, synth_tables
    from vid2player3d_amd.motion_lib import MotionLib
    from vid2player3d_amd.tasks import HumanoidSMPLIM, default_cfg

    cfg = default_cfg(32, motion_lib=MotionLib(synth_tables(), DEV), sample_first_motions=True, body_shape_mismatch="ignore")
    cfg["args"] = types.SimpleNamespace(test=True)
    task = HumanoidSMPLIM(cfg, device_type="cuda", device_id=0)
    assert task._state_init == HumanoidSMPLIM.StateInit.Start and cfg["env"]["stateInit"] == "Start"
    task.reset()
    assert float(task._reset_ref_motion_times.abs().max()) == 0.0 and float(task._cur_ref_motion_times.abs().max()) == 0.0
    task.close()


# ---------------------------------------------------------------- the reference's outputs at BASELINE config 2's size (VERDICT r5 #4b):
# 4096 motion-state queries, 1024 envs of task ops, a 1024-env x 32-step epoch - recorded by oracle/gen_golden_large.py from the reference's
# own Python; the inputs are regenerated from their seeds (oracle/golden_inputs.py), the HIP kernels meet the REFERENCE's numbers directly
def test_motion_state_4096_queries_match_the_reference(mlib, golden_tables, golden_motion_state_4096):
    from oracle import golden_inputs as GI

    g = golden_motion_state_4096
    ids, times = GI.motion_state_queries(golden_tables)
    assert np.array_equal(ids[:8], g["ids_check"]) and np.array_equal(times[:8], g["times_check"])
    res = mlib.get_motion_state(T(ids, torch.long), T(times), return_rigid_body=True, adjust_height=True, ground_tolerance=0.0)
    for name, r in zip(O.MOTION_STATE_NAMES, res):
        close(N(r)[:len(g[name])], g[name], 5e-6, name + " (4096 queries)")


def test_task_ops_on_1024_envs_match_the_reference(golden_task_ops_1024):
    from oracle import golden_inputs as GI
    from vid2player3d_amd import _lib

    g, x = golden_task_ops_1024, GI.task_ops_inputs()
    lib = _lib.load()
    n = GI.N_LARGE
    rew, sub = _reward(lib, x, n)
    close(sub[:, :3], g["sub_rewards"][:, :3], 5e-6, "sub[dof,vel,pos] (1024 envs)")
    close(sub[:, 3], g["sub_rewards"][:, 3], 2e-4, "sub[rot] (1024 envs)")
    close(rew, g["reward"], 5e-5, "reward (1024 envs)")
    h = x["reset_heights"].astype(np.float32).copy()
    h[[7, 3]] = -np.inf
    rst, term = torch.empty(n, dtype=torch.long, device=DEV), torch.empty(n, dtype=torch.long, device=DEV)
    prog, rb, ct, cl = T(x["reset_progress"], torch.long), T(x["reset_rb_pos"]), T(x["reset_cur_time"]), T(x["reset_clip_len"])
    _lib.check(lib.v2p_reset_flags(n, _lib.ptr(prog), _lib.ptr(rb), (C.c_float * 24)(*h.tolist()), _lib.ptr(ct), _lib.ptr(cl), 300.0, 1, _lib.ptr(rst), _lib.ptr(term), None),
               "v2p_reset_flags")
    assert np.array_equal(N(rst), g["reset_out"]) and np.array_equal(N(term), g["terminate_out"])
    r = GI.OBS734_ROWS
    obs = torch.empty((r, 734), device=DEV)
    args = [T(x[k][:r]) for k in ("body_pos", "body_rot", "tgt_pos", "tgt_rot", "dof_pos", "dof_vel", "tgt_dof_pos", "body_vel", "body_ang_vel", "obs734_motion_bodies")]
    _lib.check(lib.v2p_obs_imitation(r, *[_lib.ptr(a) for a in args], None, None, 0.0, _lib.ptr(obs), None), "v2p_obs_imitation")
    close(N(obs), g["obs734"], 5e-6, "obs734 (256 rows)")


def test_env_trace_of_1024_envs_matches_the_reference(mlib, golden_tables, golden_env_trace_1024):
    """One epoch of the reference's own HumanoidSMPLIM on 1024 envs (physics teacher-forced): reward, sub-rewards, sticky flags, progress and
    clip time of EVERY env at EVERY step, the in-place action masking, full observation and target rows of the 32 sampled envs - through
    reset / pre_physics_step / post_physics_step of the engine's task."""
    from oracle import golden_inputs as GI

    g = golden_env_trace_1024
    n, steps, S = int(g["n"]), int(g["steps"]), g["sample_envs"]
    ids = GI.env_trace_motion_ids(golden_tables, n)
    task = make_task(n, mlib, motion_ids=ids)
    task.reset_with_times(None, T(g["reset_motion_times"]))
    close(N(task._humanoid_root_states), g["reset_root_states"], 5e-6, "reset root states")
    close(N(task.obs_buf)[S], g["reset_obs_sample"], 5e-6, "reset obs")
    close(N(task.context_feat)[S[:8]], g["context_feat_sample"], 5e-6, "context_feat")
    assert np.array_equal(N(task.context_mask).astype(np.uint8), g["context_mask"])
    for i in range(steps):
        x = GI.env_trace_step_inputs(golden_tables, ids, g["reset_motion_times"], i)
        a = T(x["actions"])
        task.pre_physics_step(a)
        assert np.array_equal((N(a) == 0).all(axis=1).astype(np.uint8), g["actions_masked_rows"][i]), i
        task._dof_pos[:] = T(x["dof_pos"])
        task._dof_vel[:] = T(x["dof_vel"])
        task._rigid_body_state.view(n, 24, 13)[:] = T(x["rb_state"])
        task._humanoid_root_states[:] = T(x["rb_state"][:, 0, :])
        task._reset_env_tensors(None, with_rb_state=True)
        task.post_physics_step()
        close(N(task.rew_buf), g["rew"][i], 1e-4, "rew %d" % i)
        close(N(task.extras["sub_rewards"]), g["sub_rewards"][i], 5e-4, "sub_rewards %d" % i)
        assert np.array_equal(N(task.reset_buf), g["reset"][i]) and np.array_equal(N(task.extras["terminate"]), g["terminate"][i]), i
        assert np.array_equal(N(task.progress_buf), g["progress"][i]), i
        close(N(task._cur_ref_motion_times), g["cur_time"][i], 1e-6, "cur_time %d" % i)
        close(N(task.obs_buf)[S], g["obs_sample"][i], 5e-6, "obs %d" % i)
        tgt = np.concatenate([N(getattr(task, "_target_" + k))[S].reshape(len(S), -1) for k in O.MOTION_STATE_NAMES], axis=1)
        close(tgt, g["target_sample"][i], 5e-6, "target %d" % i)
    task.close()
