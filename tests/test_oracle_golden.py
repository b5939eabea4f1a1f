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
