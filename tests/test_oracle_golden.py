"""Pin the numpy oracle (oracle/task_oracle.py) to golden vectors recorded from the reference's own Python."""
import numpy as np
import pytest

from oracle import task_oracle as T

TOL = 2e-6  # float32 op-order noise between torch and numpy


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    tol = tol * max(1.0, np.abs(b).max() if b.size else 1.0)  # relative to the largest magnitude
    assert err <= tol, "%s: max abs err %.3e > %.1e" % (what, err, tol)


@pytest.mark.parametrize("adjust", [True, False])
def test_get_motion_state(golden_tables, golden_motion_state, adjust):
    g = golden_motion_state
    res = T.get_motion_state(golden_tables, g["ids"], g["times"], adjust_height=adjust, ground_tolerance=0.0)
    sfx = "" if adjust else "_noadj"
    for name, r in zip(T.MOTION_STATE_NAMES, res):
        close(r, g[name + sfx], 5e-6, name + sfx)


def test_reward_and_dof_obs(golden_task_ops):
    g = golden_task_ops
    close(T.dof_to_obs(g["dof_pos"]), g["dof_obs"], what="dof_obs")
    rew, sub = T.compute_humanoid_reward(g["body_pos"], g["body_rot"], g["tgt_pos"], g["tgt_rot"], g["dof_pos"], g["dof_vel"],
                                         g["tgt_dof_pos"], g["tgt_dof_vel"], g["body_pos_weights"])
    # the rotation term goes through acos near 1: float32 conditioning there is ~3e-4 rad in the angle
    close(sub[:, :3], g["sub_rewards"][:, :3], 5e-6, "sub_rewards[dof,vel,pos]")
    close(sub[:, 3], g["sub_rewards"][:, 3], 2e-4, "sub_rewards[rot]")
    close(rew, g["reward"], 5e-5, "reward")


def test_reset(golden_task_ops):
    g = golden_task_ops
    rst, term = T.compute_humanoid_reset(g["reset_progress"], g["reset_rb_pos"], g["reset_heights"], g["reset_cur_time"], g["reset_clip_len"])
    assert np.array_equal(rst, g["reset_out"])
    assert np.array_equal(term, g["terminate_out"])
    assert term.sum() > 3 and (rst - term).sum() > 3  # the fixture exercises both branches


def test_pre_physics(golden_task_ops):
    g = golden_task_ops
    a, pd_tar, pd_torque, force, torque = T.pre_physics(g["pre_actions"], g["pre_reset"], g["dof_pos"], g["body_rot"][:, 0], g["pre_kp"])
    close(a, g["pre_actions_masked"], 0.0, "masked actions")
    close(pd_tar, g["pre_pd_tar"], 0.0, "pd_tar")
    close(pd_torque, g["pre_pd_torque"], 1e-6, "pd_torque")
    close(force, g["pre_res_force"], 2e-6, "res_force")
    close(torque, g["pre_res_torque"], 2e-6, "res_torque")


def test_obs734(golden_task_ops):
    g = golden_task_ops
    o = T.obs_imitation_734(g["body_pos"], g["body_rot"], g["tgt_pos"], g["tgt_rot"], g["dof_pos"], g["dof_vel"], g["tgt_dof_pos"],
                            g["body_vel"], g["body_ang_vel"], g["obs734_motion_bodies"])
    close(o, g["obs734"], 5e-6, "obs734")


def test_running_norm_and_gae(golden_task_ops):
    g = golden_task_ops
    close(T.running_norm_eval(g["obs734"], g["rn_mean"], g["rn_std"]), g["obs734_normed"], 5e-6, "running norm")
    adv = T.discount_values(g["gae_fdones"], g["gae_values"], g["gae_rewards"], g["gae_next_values"], float(g["gae_gamma"]), float(g["gae_tau"]))
    close(adv, g["gae_advs"], 2e-6, "gae")


def test_env_trace(golden_tables, golden_env_trace):
    """The task state machine replayed against the reference HumanoidSMPLIM trace (teacher-forced physics)."""
    g = golden_env_trace
    from vid2player3d_amd.model import load_baked_model

    kp = load_baked_model().kp.astype(np.float32)
    task = T.TaskOracle(golden_tables, g["motion_ids"], kp)
    for tag, nsteps in (("e0_", int(g["num_steps"])), ("e1_", int(g["num_steps_e1"]))):
        task.reset_all(g[tag + "reset_motion_times"])
        close(task.root_states, g[tag + "reset_root_states"], 5e-6, tag + "root_states")
        close(task.dof_pos, g[tag + "reset_dof_pos"], 5e-6, tag + "dof_pos")
        close(task.dof_vel, g[tag + "reset_dof_vel"], 5e-6, tag + "dof_vel")
        close(task.rb_state, g[tag + "reset_rb_state"], 5e-6, tag + "rb_state")
        close(task.context_feat, g[tag + "context_feat"], 5e-6, tag + "context_feat")
        assert np.array_equal(task.context_mask, g[tag + "context_mask"])
        close(task.obs_buf, g[tag + "reset_obs"], 5e-6, tag + "reset obs")
        assert not task.reset_buf.any() and not task.terminate_buf.any()
        for i in range(nsteps):
            p = "%ss%02d_" % (tag, i)
            out = task.pre_physics_step(g[p + "actions"])
            close(out[0], g[p + "actions_after"], 0.0, p + "actions_after")
            close(task.pd_torque, g[p + "pd_torque"], 2e-6, p + "pd_torque")
            task.set_sim_state(g[p + "sim_dof_pos"], g[p + "sim_dof_vel"], g[p + "sim_rb_state"])
            task.post_physics_step()
            close(task.obs_buf, g[p + "obs"], 5e-6, p + "obs")
            close(task.rew_buf, g[p + "rew"], 1e-4, p + "rew")
            close(task.sub_rewards, g[p + "sub_rewards"], 5e-4, p + "sub_rewards")
            assert np.array_equal(task.reset_buf, g[p + "reset"]), p
            assert np.array_equal(task.terminate_buf, g[p + "terminate"]), p
            assert np.array_equal(task.progress_buf, g[p + "progress"]), p
            close(task.cur_time, g[p + "cur_time"], 1e-6, p + "cur_time")
            for k, name in enumerate(T.MOTION_STATE_NAMES):
                close(task.target[k], g[p + "target_" + name], 5e-6, p + "target_" + name)


# ---------------------------------------------------------------- the same functions at BASELINE config 2's size (VERDICT r5 #4b): the
# reference's outputs on 1024 envs / 4096 queries / a 1024-env x 32-step epoch; inputs regenerated from their seeds
def test_large_motion_state(golden_tables, golden_motion_state_4096):
    from oracle import golden_inputs as GI

    g = golden_motion_state_4096
    ids, times = GI.motion_state_queries(golden_tables)
    assert len(ids) == int(g["q"]) == 4096 and np.array_equal(ids[:8], g["ids_check"]) and np.array_equal(times[:8], g["times_check"])
    res = T.get_motion_state(golden_tables, ids, times, adjust_height=True, ground_tolerance=0.0)
    for name, r in zip(T.MOTION_STATE_NAMES, res):
        close(r[:len(g[name])], g[name], 5e-6, name)
    assert (times < 0).sum() > 100 and (times > golden_tables["motion_lengths"][ids]).sum() > 100  # both ends are exercised


def test_large_task_ops(golden_task_ops_1024):
    from oracle import golden_inputs as GI

    g, x = golden_task_ops_1024, GI.task_ops_inputs()
    assert int(g["n"]) == GI.N_LARGE == len(x["dof_pos"])
    close(T.dof_to_obs(x["dof_pos"]), g["dof_obs"], what="dof_obs")
    rew, sub = T.compute_humanoid_reward(x["body_pos"], x["body_rot"], x["tgt_pos"], x["tgt_rot"], x["dof_pos"], x["dof_vel"], x["tgt_dof_pos"], x["tgt_dof_vel"],
                                         x["body_pos_weights"])
    close(sub[:, :3], g["sub_rewards"][:, :3], 5e-6, "sub_rewards[dof,vel,pos]")
    close(sub[:, 3], g["sub_rewards"][:, 3], 2e-4, "sub_rewards[rot]")
    close(rew, g["reward"], 5e-5, "reward")
    rst, term = T.compute_humanoid_reset(x["reset_progress"], x["reset_rb_pos"], x["reset_heights"], x["reset_cur_time"], x["reset_clip_len"])
    assert np.array_equal(rst, g["reset_out"]) and np.array_equal(term, g["terminate_out"]) and term.sum() > 50 and (rst - term).sum() > 50
    a, pd_tar, pd_torque, force, torque = T.pre_physics(x["pre_actions"], x["pre_reset"], x["dof_pos"], x["body_rot"][:, 0], x["pre_kp"])
    close(pd_tar, g["pre_pd_tar"], 0.0, "pd_tar")
    close(pd_torque, g["pre_pd_torque"], 1e-6, "pd_torque")
    close(force, g["pre_res_force"], 2e-6, "res_force")
    close(torque, g["pre_res_torque"], 2e-6, "res_torque")
    assert x["pre_reset"].sum() > 10 and (a[x["pre_reset"] == 1] == 0).all()
    r = GI.OBS734_ROWS
    o = T.obs_imitation_734(x["body_pos"][:r], x["body_rot"][:r], x["tgt_pos"][:r], x["tgt_rot"][:r], x["dof_pos"][:r], x["dof_vel"][:r], x["tgt_dof_pos"][:r],
                            x["body_vel"][:r], x["body_ang_vel"][:r], x["obs734_motion_bodies"][:r])
    close(o, g["obs734"], 5e-6, "obs734")


def test_large_env_trace(golden_tables, golden_env_trace_1024):
    """TaskOracle over one epoch of 1024 envs against what the reference's own HumanoidSMPLIM recorded: reward, sub-rewards, sticky flags,
    progress and clip time of every env at every step; full observation and target rows of the sampled envs."""
    from oracle import golden_inputs as GI
    from vid2player3d_amd.model import load_baked_model

    g = golden_env_trace_1024
    n, steps, S = int(g["n"]), int(g["steps"]), g["sample_envs"]
    ids = GI.env_trace_motion_ids(golden_tables, n)
    task = T.TaskOracle(golden_tables, ids, load_baked_model().kp.astype(np.float32))
    task.reset_all(g["reset_motion_times"])
    close(task.root_states, g["reset_root_states"], 5e-6, "reset root states")
    close(task.obs_buf[S], g["reset_obs_sample"], 5e-6, "reset obs")
    close(task.context_feat[S[:8]], g["context_feat_sample"], 5e-6, "context_feat")
    assert np.array_equal(task.context_mask.astype(np.uint8), g["context_mask"])
    for i in range(steps):
        x = GI.env_trace_step_inputs(golden_tables, ids, g["reset_motion_times"], i)
        out = task.pre_physics_step(x["actions"])
        assert np.array_equal((out[0] == 0).all(axis=1).astype(np.uint8), g["actions_masked_rows"][i]), i
        task.set_sim_state(x["dof_pos"], x["dof_vel"], x["rb_state"])
        task.post_physics_step()
        close(task.rew_buf, g["rew"][i], 1e-4, "rew %d" % i)
        close(task.sub_rewards, g["sub_rewards"][i], 5e-4, "sub_rewards %d" % i)
        assert np.array_equal(task.reset_buf, g["reset"][i]) and np.array_equal(task.terminate_buf, g["terminate"][i]), i
        assert np.array_equal(task.progress_buf, g["progress"][i]), i
        close(task.cur_time, g["cur_time"][i], 1e-6, "cur_time %d" % i)
        close(task.obs_buf[S], g["obs_sample"][i], 5e-6, "obs %d" % i)
        close(np.concatenate([t[S].reshape(len(S), -1) for t in task.target], axis=1), g["target_sample"][i], 5e-6, "target %d" % i)
    assert g["terminate"][-1].sum() >= 20 and (g["reset"][-1] - g["terminate"][-1]).sum() >= 0 and g["reset"][20].sum() > g["reset"][19].sum()
