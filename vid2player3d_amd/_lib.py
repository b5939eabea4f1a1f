"""ctypes binding of libv2p_rollout.so (the C ABI in include/v2p_rollout.h).

There is NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised
(the reference's error convention is Python exceptions).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libv2p_rollout.so")

NUM_BODIES, NUM_DOF, NUM_ACTIONS, NUM_OBS = 24, 69, 75, 461
MOTION_STATE_DIM, CONTEXT_DIM = 331, 378
ABI_VERSION = 14

c_f = C.POINTER(C.c_float)
c_i32 = C.POINTER(C.c_int32)
c_i64 = C.POINTER(C.c_int64)
c_u8 = C.POINTER(C.c_uint8)
vp = C.c_void_p


class ModelDesc(C.Structure):
    _fields_ = [("num_bodies", C.c_int32), ("parents", c_i32), ("local_pos", c_f), ("mass", c_f), ("com", c_f), ("inertia", c_f),
                ("kp", c_f), ("kd", c_f), ("armature", c_f), ("hull_offsets", c_i32), ("hull_verts", c_f),
                ("limit_lower", c_f), ("limit_upper", c_f)]


class MotionTables(C.Structure):
    _fields_ = [("num_motions", C.c_int64), ("num_frames_total", C.c_int64), ("gts", vp), ("grs", vp), ("lrs", vp), ("grvs", vp),
                ("gravs", vp), ("dvs", vp), ("motion_lengths", vp), ("motion_num_frames", vp), ("motion_dt", vp),
                ("motion_min_verts_h", vp), ("length_starts", vp), ("motion_bodies", vp), ("key_body_ids", C.c_int32 * 4)]


class SimCfg(C.Structure):
    _fields_ = [("sim_dt", C.c_float), ("substeps", C.c_int32), ("control_freq_inv", C.c_int32), ("num_solver_iterations", C.c_int32),
                ("enable_contact", C.c_int32), ("residual_hold_sims", C.c_int32), ("gravity_z", C.c_float), ("friction", C.c_float),
                ("contact_offset", C.c_float), ("max_depenetration_velocity", C.c_float), ("erp", C.c_float),
                ("angular_damping", C.c_float), ("max_angular_velocity", C.c_float), ("pd_tar_lim", C.c_float),
                ("residual_force_scale", C.c_float), ("residual_torque_scale", C.c_float), ("ground_tolerance", C.c_float),
                ("max_episode_length", C.c_float), ("enable_early_termination", C.c_int32), ("context_length", C.c_int32),
                ("context_padding", C.c_int32), ("term_heights", C.c_float * 24), ("body_pos_weights", C.c_float * 24),
                ("reward_specs", C.c_float * 8), ("freeze_terminated_envs", C.c_int32), ("schedule", C.c_int32), ("pair_envs_by_load", C.c_int32),
                ("solver_type", C.c_int32), ("substep_jobs", C.c_int32), ("job_mono_permille", C.c_int32), ("pair_mix_permille", C.c_int32), ("debug_contacts", C.c_int32),
                ("joint_limits", C.c_int32), ("limit_margin", C.c_float),
                ("rest_offset", C.c_float), ("bounce_threshold_velocity", C.c_float), ("num_velocity_iterations", C.c_int32), ("kernel_build", C.c_int32),
                ("job_timeout_spins", C.c_int32), ("job_len", C.c_int32), ("job_lead", C.c_int32), ("job_no_interleave", C.c_int32),
                ("friction_frame", C.c_int32)]


class EnvBuffers(C.Structure):
    _fields_ = [("root_states", vp), ("dof_state", vp), ("rb_state", vp), ("contact_force", vp), ("dof_force", vp), ("pd_target", vp),
                ("obs", vp), ("rew", vp), ("sub_rewards", vp), ("reset", vp), ("terminate", vp), ("progress", vp), ("cur_time", vp),
                ("reset_time", vp), ("target", vp * 2), ("context_feat", vp), ("context_mask", vp)]


class BallCfg(C.Structure):
    _fields_ = [("radius", C.c_float), ("mass", C.c_float), ("inertia", C.c_float), ("restitution_ground", C.c_float), ("friction_ground", C.c_float),
                ("restitution_racket", C.c_float), ("friction_racket", C.c_float), ("bounce_threshold_velocity", C.c_float),
                ("angular_damping", C.c_float), ("max_angular_velocity", C.c_float), ("spin_scale", C.c_float), ("racket_link", C.c_int32),
                ("num_cylinders", C.c_int32), ("cylinders", (C.c_float * 8) * 2), ("racket_offset", C.c_float * 3),
                ("restitution_body", C.c_float), ("friction_body", C.c_float), ("body_contacts", C.c_int32), ("bounce_height", C.c_float),
                ("poll_racket_hits", C.c_int32)]


class BallBuffers(C.Structure):
    _fields_ = [("ball_state", vp), ("racket_state", vp), ("ball_per_sim", vp), ("racket_hit_per_sim", vp), ("ball_contact", vp),
                ("ball_body_contact", vp), ("has_bounce", vp), ("has_bounce_now", vp), ("bounce_pos", vp), ("has_racket_contact", vp),
                ("has_racket_contact_now", vp), ("contact_force_sum", vp)]


_lib = None


def load():
    """Load the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libv2p_rollout.so is missing (%s): build it with `python -m vid2player3d_amd.build`; "
                           "there is no CPU fallback for the rollout engine" % LIB_PATH)
    # PyTorch-ROCm owns the device memory and the streams we launch on, and it ships its own HIP
    # runtime: import it first so that this library binds to the SAME libamdhip64 instance (two HIP
    # runtimes in one process do not see each other's devices or streams).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    lib.v2p_last_error.restype = C.c_char_p
    lib.v2p_abi_version.restype = C.c_int
    if lib.v2p_abi_version() != ABI_VERSION:
        raise RuntimeError("libv2p_rollout.so ABI %d != binding ABI %d: rebuild" % (lib.v2p_abi_version(), ABI_VERSION))
    sig = {
        "v2p_model_create": [C.POINTER(ModelDesc), C.c_int, C.POINTER(vp)],
        "v2p_mlib_create": [C.POINTER(MotionTables), C.c_int, C.POINTER(vp)],
        "v2p_motion_state": [vp, vp, vp, C.c_int64, C.c_int, C.c_float, C.POINTER(vp * 9), vp],
        "v2p_reward": [C.c_int64] + [vp] * 8 + [c_f, c_f, vp, vp, vp],
        "v2p_reset_flags": [C.c_int64, vp, vp, c_f, vp, vp, C.c_float, C.c_int, vp, vp, vp],
        "v2p_obs_imitation": [C.c_int64] + [vp] * 10 + [vp, vp, C.c_float, vp, vp],
        "v2p_obs_imitation_packed": [C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_float, vp, vp],
        "v2p_policy_head": [C.c_int64, vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp],
        "v2p_policy_head_record": [C.c_int64, vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp],
        "v2p_gae": [C.c_int64, C.c_int64, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp],
        "v2p_value_record": [C.c_int64, vp, vp, vp, C.c_float, vp, vp, vp, vp],
        "v2p_rollout_record": [C.c_int64, vp, C.c_int64] + [vp] * 15,
        "v2p_motion_tables_build": [C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, c_i32, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp],
        "v2p_shapes_compile": [C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_int32, C.c_double, C.c_int32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp],
        "v2p_env_create": [vp, vp, C.POINTER(SimCfg), vp, C.c_int64, C.POINTER(EnvBuffers), C.c_int, C.POINTER(vp)],
        "v2p_env_create_shapes": [C.POINTER(vp), C.c_int32, c_i32, vp, C.POINTER(SimCfg), vp, C.c_int64, C.POINTER(EnvBuffers), C.c_int, C.POINTER(vp)],
        "v2p_env_reset": [vp, vp, C.c_int64, vp, vp],
        "v2p_env_context": [vp, vp, C.c_int64, vp, vp],
        "v2p_env_step": [vp, vp, vp],
        "v2p_env_pre_physics": [vp, vp, vp],
        "v2p_env_physics": [vp, vp],
        "v2p_env_export": [vp, vp],
        "v2p_env_post_physics": [vp, vp],
        "v2p_env_push_state": [vp, vp, C.c_int64, C.c_int, vp],
        "v2p_env_target_index": [vp],
        "v2p_env_kernel_build": [vp],
        "v2p_env_set_schedule": [vp, C.c_int],
        "v2p_env_debug_contacts": [vp, vp, vp],
        "v2p_env_debug_contacts_substeps": [vp, vp, vp],
        "v2p_env_debug_pairing": [vp, vp, vp, vp],
        "v2p_env_check": [vp, vp],
        "v2p_env_check_async": [vp, vp],
        "v2p_env_job_recoveries": [vp, C.POINTER(C.c_int64)],
        "v2p_env_jobs_skipped": [vp, C.POINTER(C.c_int64)],
        "v2p_env_attach_ball": [vp, C.POINTER(BallCfg), C.POINTER(BallBuffers)],
        "v2p_env_profile_begin": [vp, C.c_int64],
        "v2p_env_profile_begin_sampled": [vp, C.c_int64, C.c_int32, C.c_int32],
        "v2p_env_profile_end": [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("v2p_model_destroy", "v2p_mlib_destroy", "v2p_env_destroy"):
        fn = getattr(lib, name)
        fn.argtypes = [vp]
        fn.restype = None
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "v2p_model_create", "v2p_model_destroy", "v2p_mlib_create", "v2p_mlib_destroy", "v2p_motion_state", "v2p_reward", "v2p_reset_flags",
    "v2p_obs_imitation", "v2p_obs_imitation_packed", "v2p_policy_head", "v2p_policy_head_record", "v2p_gae", "v2p_value_record", "v2p_rollout_record", "v2p_motion_tables_build", "v2p_shapes_compile", "v2p_env_create", "v2p_env_create_shapes", "v2p_env_destroy", "v2p_env_reset", "v2p_env_context", "v2p_env_step", "v2p_env_pre_physics", "v2p_env_physics", "v2p_env_export",
    "v2p_env_post_physics", "v2p_env_push_state", "v2p_env_target_index", "v2p_env_kernel_build", "v2p_env_set_schedule", "v2p_env_debug_contacts", "v2p_env_debug_contacts_substeps", "v2p_env_debug_pairing", "v2p_env_attach_ball", "v2p_env_check", "v2p_env_check_async", "v2p_env_job_recoveries", "v2p_env_jobs_skipped", "v2p_env_profile_begin", "v2p_env_profile_begin_sampled", "v2p_env_profile_end", "v2p_last_error", "v2p_abi_version",
)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().v2p_last_error().decode()))


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
