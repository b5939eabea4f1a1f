"""HumanoidSMPLIM: the imitation task of embodied_pose, running on the MI355X rollout engine.

Drop-in for `embodied_pose/env/tasks/humanoid_smpl_im.py:HumanoidSMPLIM` behind the VecTask
surface: same constructor signature `(cfg, sim_params, physics_engine,
device_type, device_id, headless)`, same buffers (`obs_buf rew_buf reset_buf progress_buf
states_buf extras`), same methods (`step reset register_model pre_epoch get_aux_losses
render_vis`), same back-channel attributes the agent reads (`context_feat context_mask
smpl_rest_joints smpl_parents smpl_children body_names _motion_lib _reset_ref_motion_ids
_cur_ref_motion_times context_length`), and the tensor views the reference task keeps over
the Isaac Gym state tensors (`_rigid_body_pos`, `_dof_pos`, `_humanoid_root_states`, ...).

Every per-step / per-reset computation is a HIP kernel reached through the C ABI
(include/v2p_rollout.h); this class only owns the torch tensors and the Python-side RNG.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass, field
from enum import Enum

import numpy as np
import torch

from .. import _lib
from ..model import BodyModel, load_baked_model
from ..motion_lib import MotionLib


@dataclass
class PhysxParams:
    """sim.physx block of cfg/amass_im.yaml:39-48"""
    num_threads: int = 4
    solver_type: int = 1
    num_position_iterations: int = 4
    num_velocity_iterations: int = 0
    contact_offset: float = 0.02
    rest_offset: float = 0.0
    bounce_threshold_velocity: float = 0.2
    max_depenetration_velocity: float = 10.0
    default_buffer_size_multiplier: float = 10.0


@dataclass
class SimParams:
    """What embodied_pose/utils/config.py:190-222 (`parse_sim_params`) hands to the task."""
    dt: float = 1.0 / 60.0
    substeps: int = 2
    gravity: tuple = (0.0, 0.0, -9.81)
    physx: PhysxParams = field(default_factory=PhysxParams)
    given: set = field(default_factory=set)   # sim.physx keys the yaml block names itself (from_cfg)

    @classmethod
    def from_cfg(cls, sim_cfg):
        """The `sim` block of a task yaml (cfg/amass_im.yaml:37-52).  `given` records which sim.physx keys the block names itself, so
        that a choice the file states (solver_type: 1) is told apart from a default of this class."""
        sp = cls()
        sim_cfg = sim_cfg or {}
        sp.dt = float(sim_cfg.get("dt", sp.dt))
        sp.substeps = int(sim_cfg.get("substeps", sp.substeps))
        for k, v in (sim_cfg.get("physx") or {}).items():
            if hasattr(sp.physx, k):
                setattr(sp.physx, k, type(getattr(sp.physx, k))(v))
                sp.given.add(k)
        return sp


SOLVER_NAMES = {0: "pgs", 1: "tgs"}   # gymapi: sim.physx.solver_type 0 = PGS, 1 = TGS (amass_im.yaml:41, config.py:203)


def resolve_contact_solver(env, sim_params, log=None):
    """Which contact solver the engine runs: (name, source).

    The reference's files decide: `sim.physx.solver_type` (cfg/amass_im.yaml:41 says 1 = TGS; `parse_sim_params`, utils/config.py:203,
    sets 1 before the yaml is read) -> 1 = "tgs", 0 = "pgs".  `env.contact_solver` ("pgs" | "tgs"), a key of THIS engine, overrides it
    - and says so through `log` when the two disagree.  With neither stated (this package's `default_cfg()`, which names no solver
    type) the engine's default is PGS, the solver BASELINE.json's config 3 is worded on."""
    physx = getattr(sim_params, "physx", None)
    given = getattr(sim_params, "given", None)   # None: a foreign SimParams object (e.g. gymapi's): what it holds was stated by its maker
    stated = physx is not None and hasattr(physx, "solver_type") and (given is None or "solver_type" in given)
    from_sim = None
    if stated:
        st = int(physx.solver_type)
        if st not in SOLVER_NAMES:
            raise ValueError("sim.physx.solver_type = %r: 0 (PGS) or 1 (TGS)" % (physx.solver_type,))
        from_sim = SOLVER_NAMES[st]
    if "contact_solver" in env:
        name = env["contact_solver"]
        if name not in ("pgs", "tgs"):
            raise ValueError("env.contact_solver = %r: 'pgs' or 'tgs'" % (name,))
        if from_sim is not None and from_sim != name and log is not None:
            log("vid2player3d_amd: env.contact_solver = %r overrides sim.physx.solver_type = %d (%s)" % (name, int(physx.solver_type), from_sim))
        return name, "env.contact_solver"
    if from_sim is not None:
        return from_sim, "sim.physx.solver_type"
    return "pgs", "engine default"


def fill_physx(c, sim_params, env, log=None):
    """The sim.physx block (cfg/amass_im.yaml:39-48) -> v2p_sim_cfg; every key either reaches the engine or is refused there
    (num_threads and default_buffer_size_multiplier size PhysX's own host threads / GPU buffers and have no counterpart).  Returns the
    (solver name, source) pair."""
    px = sim_params.physx
    c.num_solver_iterations = int(px.num_position_iterations)
    c.num_velocity_iterations = int(getattr(px, "num_velocity_iterations", 0))
    c.contact_offset = float(px.contact_offset)
    c.rest_offset = float(getattr(px, "rest_offset", 0.0))
    c.bounce_threshold_velocity = float(getattr(px, "bounce_threshold_velocity", 0.2))
    c.max_depenetration_velocity = float(px.max_depenetration_velocity)
    name, source = resolve_contact_solver(env, sim_params, log)
    c.solver_type = {"pgs": 0, "tgs": 1}[name]
    return name, source


def default_cfg(num_envs=8192, **env_overrides):
    """The env/sim blocks of embodied_pose/cfg/amass_im.yaml as a dict."""
    env = {
        "numEnvs": num_envs, "envSpacing": 5, "episodeLength": 300, "enableDebugVis": False, "pdControl": True, "powerScale": 1.0,
        "controlFrequencyInv": 2, "stateInit": "Hybrid", "hybridInitProb": 1.0, "numAMPObsSteps": 10, "enableHistObs": False,
        "localRootObs": True, "keyBodies": ["R_Ankle", "L_Ankle", "L_Hand", "R_Hand"], "contactBodies": ["R_Ankle", "L_Ankle"],
        "terminationBodyHeight": -0.5, "terminationHeadHeight": 1.0, "enableEarlyTermination": True, "residual_force_scale": 31.85,
        "context_length": 32, "context_padding": 8,
        "plane": {"staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0},
    }
    env.update(env_overrides)
    sim = {"substeps": 2, "physx": {"num_position_iterations": 4, "num_velocity_iterations": 0, "contact_offset": 0.02,
                                    "rest_offset": 0.0, "bounce_threshold_velocity": 0.2, "max_depenetration_velocity": 10.0}}
    return {"env": env, "sim": sim, "args": None}


# uhc/smpllib/smpl_parser.py:10-35: the SMPL joints in SMPL order, by the names the MJCF bodies carry
from ..body_shapes import SMPL_JOINT_NAMES as _SMPL_JOINT_NAMES  # noqa: E402

SMPL_BONE_ORDER_NAMES = list(_SMPL_JOINT_NAMES)


class HumanoidSMPLIM:
    class StateInit(Enum):
        Default = 0
        Start = 1
        Random = 2
        Hybrid = 3

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True):
        self.cfg = cfg
        env = cfg["env"]
        self.args = cfg.get("args")
        if device_type not in ("cuda", "GPU"):
            raise RuntimeError("HumanoidSMPLIM runs on the HIP rollout engine only (device_type=%r); there is no CPU simulation path" % device_type)
        self.device = "cuda:%d" % device_id
        self.device_id = device_id
        self.headless = headless
        self.model = None
        self.viewer = None
        sim_params = sim_params or SimParams.from_cfg(cfg.get("sim"))
        self.sim_params = sim_params

        # ---- config (humanoid_smpl_im.py:54-117, humanoid_smpl.py:24-64)
        self.has_shape_obs = env.get("has_shape_obs", False)
        self.has_self_collision = env.get("has_self_collision", False)
        if self.has_self_collision:
            raise NotImplementedError("has_self_collision=True (hull-hull contacts) is not built; the reference default is False")
        self.residual_force_scale = env.get("residual_force_scale", 0.0)
        self.residual_torque_scale = env.get("residual_torque_scale", self.residual_force_scale)
        self.kp_scale = env.get("kp_scale", 1.0)
        self.kd_scale = env.get("kd_scale", self.kp_scale)
        self.context_length = env.get("context_length", 32)
        self.context_padding = env.get("context_padding", 8)
        self.truncate_time = env.get("truncate_time", True)
        self.pd_tar_lim = env.get("pd_tar_lim", 0.5) * np.pi
        self.control_freq_inv = env.get("controlFrequencyInv", 1)
        self._motion_sync_dt = self.control_freq_inv * sim_params.dt
        self.dt = self.control_freq_inv * sim_params.dt
        self._pd_control = env.get("pdControl", True)
        if not self._pd_control:
            raise NotImplementedError("pdControl=False (direct torque actuation) is not built; amass_im/djokovic_im use PD targets")
        if getattr(self.args, "test", False):
            # `run.py --test` (humanoid_smpl_im.py:78-81): every clip from its first frame, the test clips when the yaml names some
            env["stateInit"] = "Start"
            if "test_motion_file" in env:
                env["motion_file"] = env["test_motion_file"]
        self._state_init = HumanoidSMPLIM.StateInit[env.get("stateInit", "Hybrid")]
        self._hybrid_init_prob = env.get("hybridInitProb", 1.0)
        self.ground_tolerance = env.get("ground_tolerance", 0.0)
        self.max_episode_length = env.get("episodeLength", 300)
        self._local_root_obs = env.get("localRootObs", True)
        self._root_height_obs = env.get("rootHeightObs", True)
        self._enable_early_termination = env.get("enableEarlyTermination", True)
        self.num_envs = int(env["numEnvs"])
        self.record_pd_torque = bool(env.get("record_pd_torque", False))

        # ---- body model (replaces Robot.load_from_skeleton + gym.load_asset, :231-298)
        # `body_model`: one BodyModel for every env, or a list of them = one body shape per clip of the motion library (the
        # reference builds one asset per sampled clip from its betas/scale, :255-296); `motion_shape_ids` [num_motions] maps
        # clips to list entries when several clips share a shape (default: clip i -> shape i)
        bm = env.get("body_model")
        self.body_shapes = list(bm) if isinstance(bm, (list, tuple)) else None
        if self.body_shapes is not None:
            bm = self.body_shapes[0]
        self.body_model = bm if isinstance(bm, BodyModel) else load_baked_model(
            default_humanoid_mass=env.get("default_humanoid_mass", 90.0), kp_scale=self.kp_scale, kd_scale=self.kd_scale)
        self.body_names = list(self.body_model.body_names)
        self.num_bodies = self.body_model.num_bodies
        self.num_dof = self._num_dof = self.body_model.num_dof
        self._dof_body_ids = self.body_model.dof_body_ids
        self._dof_offsets = self.body_model.dof_offsets
        self._dof_obs_size = len(self._dof_body_ids) * 6
        self._num_actions = self._num_dof + (6 if self.residual_force_scale > 0 else 0)
        if self._num_actions != _lib.NUM_ACTIONS:
            raise NotImplementedError("residual_force_scale must be > 0 (75-d actions); got %d actions" % self._num_actions)
        self.humanoid_masses = np.full(self.num_envs, self.body_model.total_mass)

        # ---- motion library (:420-440)
        self._motion_lib = self._load_motion(env)
        mb_dim = self._motion_lib._motion_bodies.shape[-1]
        self.obs_names = ["body_pos", "body_rot", "dof_pos", "dof_vel", "body_vel", "body_ang_vel", "motion_bodies"]
        nb = self.num_bodies
        shape_dict = {"body_pos": (nb, 3), "body_pos_gt": (nb, 3), "body_rot": (nb, 4), "dof_pos": (self._num_dof,), "dof_pos_gt": (self._num_dof,),
                      "dof_vel": (self._num_dof,), "body_vel": (nb, 3), "body_ang_vel": (nb, 3), "motion_bodies": (mb_dim,), "joint_conf": (nb,)}
        self.obs_shapes = [shape_dict[x] for x in self.obs_names]
        self.obs_dims = [int(np.prod(x)) for x in self.obs_shapes]
        self.context_names = ["body_pos", "body_rot", "dof_pos", "body_pos_gt", "dof_pos_gt"]
        if "transform_specs" in env:
            raise NotImplementedError("transform_specs (joint masking / noise on the context) is not built")
        self.context_shapes = [shape_dict[x] for x in self.context_names]
        self.context_dims = [int(np.prod(x)) for x in self.context_shapes]
        self.is_env_dim_setup = False
        self._num_obs = sum(self.obs_dims)
        if self._num_obs != _lib.NUM_OBS:
            raise RuntimeError("observation size %d != %d" % (self._num_obs, _lib.NUM_OBS))
        self.num_obs = self._num_obs
        self.num_states = 0
        self.num_actions = self._num_actions
        env["numObservations"] = self.num_obs
        env["numActions"] = self.num_actions

        # each env is bound to one clip for its lifetime (:247-254)
        mdev = self._motion_lib._device
        if env.get("sample_first_motions", False):
            ids = torch.arange(self.num_envs, device=mdev) % self._motion_lib.num_motions()
        else:
            ids = self._motion_lib.sample_motions(self.num_envs, weights_from_lenth=env.get("motion_weights_from_length", False))
        if "motion_id" in env:
            ids[:] = env["motion_id"]
        if "motion_ids" in env:  # explicit clip of every env (tests)
            ids = torch.as_tensor(np.asarray(env["motion_ids"]), dtype=torch.long)
        if ids.numel() != self.num_envs or int(ids.min()) < 0 or int(ids.max()) >= self._motion_lib.num_motions():
            raise ValueError("motion ids must be %d values in [0, %d)" % (self.num_envs, self._motion_lib.num_motions()))
        self._reset_ref_motion_ids = ids.to(self.device).contiguous()
        self._reset_ref_motion_bodies = self._motion_lib._motion_bodies[self._reset_ref_motion_ids].to(self.device)

        self._check_body_shapes(env)
        self._allocate_buffers()
        self._build_termination_heights()
        key_bodies, contact_bodies = env.get("keyBodies", []), env.get("contactBodies", [])
        self._key_body_ids = torch.tensor([self.body_names.index(b) for b in key_bodies], device=self.device, dtype=torch.long)
        self._contact_body_ids = torch.tensor([self.body_names.index(b) for b in contact_bodies], device=self.device, dtype=torch.long)
        self.body_pos_weights = torch.ones(self.num_bodies, device=self.device)
        for val, bodies in env.get("body_pos_weights", dict()).items():
            for body in bodies:
                self.body_pos_weights[self.body_names.index(body)] = val
        # per-env shape (index into body_shapes) = shape of the env's clip
        self._env_shape_ids = None
        if self.body_shapes is not None and len(self.body_shapes) > 1:
            m2s = np.asarray(env.get("motion_shape_ids", np.arange(self._motion_lib.num_motions())), dtype=np.int64)
            if len(m2s) != self._motion_lib.num_motions() or m2s.min() < 0 or m2s.max() >= len(self.body_shapes):
                raise ValueError("motion_shape_ids must map each of the %d clips to one of the %d body shapes" % (self._motion_lib.num_motions(), len(self.body_shapes)))
            self._env_shape_ids = m2s[self._reset_ref_motion_ids.cpu().numpy()].astype(np.int32)
            self.humanoid_masses = np.array([self.body_shapes[k].total_mass for k in self._env_shape_ids])
        last = self.body_model if self._env_shape_ids is None else self.body_shapes[self._env_shape_ids[-1]]
        # (the reference keeps the gains of the LAST env it built, :382-383)
        self.stiffness = torch.tensor(last.kp, dtype=torch.float32, device=self.device)
        self.damping = torch.tensor(last.kd, dtype=torch.float32, device=self.device)

        # agent back-channels (:325-327): rest joints, parents and first children in SMPL joint order (uhc/smpllib/smpl_parser.py:10-35,
        # 340-350), what the network builder's inverse kinematics indexes with (im_network_builder.py:81-102).  The rest joints are the
        # body origins of the zero pose: the MJCF bodies sit at the SMPL joints they were generated from.
        def rest_joints(m):
            rest = np.zeros((self.num_bodies, 3))
            for b in range(self.num_bodies):
                p = m.parents[b]
                rest[b] = m.local_pos[b] + (rest[p] if p >= 0 else 0.0)
            return rest

        names = list(self.body_model.body_names)
        smpl_named = self.num_bodies == 24 and all(nm in names for nm in SMPL_BONE_ORDER_NAMES)
        if smpl_named:
            to_smpl = np.array([names.index(nm) for nm in SMPL_BONE_ORDER_NAMES])  # SMPL joint i = MJCF body to_smpl[i]
        else:  # (a body model with other bodies: MJCF order)
            to_smpl = np.arange(self.num_bodies)
        from_mjcf = np.full(self.num_bodies, -1)
        from_mjcf[to_smpl] = np.arange(len(to_smpl))
        if self._env_shape_ids is None:
            self.smpl_rest_joints = torch.tensor(rest_joints(self.body_model)[to_smpl], dtype=torch.float32, device=self.device).unsqueeze(0).repeat(self.num_envs, 1, 1)
        else:
            per_shape = np.stack([rest_joints(m)[to_smpl] for m in self.body_shapes])
            self.smpl_rest_joints = torch.tensor(per_shape[self._env_shape_ids], dtype=torch.float32, device=self.device)
        par = np.asarray(self.body_model.parents)
        smpl_par = np.array([(-1 if par[b] < 0 else from_mjcf[par[b]]) for b in to_smpl])
        self.smpl_parents = torch.tensor(smpl_par, dtype=torch.long, device=self.device)
        children = np.full(len(to_smpl), -1)  # SMPL_Parser._parents_to_children: the first child in joint order ...
        for i in range(len(to_smpl)):
            if smpl_par[i] != -1 and children[smpl_par[i]] < 0:
                children[smpl_par[i]] = i
        if smpl_named:
            children[0] = 3                                    # ... except the pelvis -> Torso
            children[9] = SMPL_BONE_ORDER_NAMES.index("Neck")  # and Chest (SPINE3) -> Neck
        self.smpl_children = torch.tensor(children, dtype=torch.long, device=self.device)

        self._create_engine()
        self._sub_rewards_names = "dof_reward,vel_reward,body_pos_reward,body_rot_reward"
        self.extras = {}
        self.actions = None

    # ------------------------------------------------------------------ construction helpers
    def _check_body_shapes(self, env):
        """The reference builds one humanoid asset per sampled clip from the clip's betas (humanoid_smpl_im.py:247-296).  A library whose
        clips carry different betas, simulated with ONE body model, gives targets / observations / termination heights of per-beta
        skeletons against a simulated skeleton that is not theirs: refuse unless the caller says so (cfg env body_shape_mismatch =
        'warn' | 'ignore'); per-clip bodies go in cfg env body_model=[BodyModel, ...] (+ motion_shape_ids)."""
        if self.body_shapes is not None and len(self.body_shapes) > 1:
            return
        mb = self._motion_lib._motion_bodies
        if mb.shape[0] < 2 or bool((mb == mb[0]).all()) or getattr(self._motion_lib, "_single_skeleton", False):
            return
        how = env.get("body_shape_mismatch", "error")
        msg = ("the motion library's clips carry %d different body shapes (gender + betas) but a single body model is simulated; pass one "
               "BodyModel per clip (cfg['env']['body_model'] = [...], optionally 'motion_shape_ids') or set cfg['env']['body_shape_mismatch'] "
               "to 'warn' / 'ignore'" % len(torch.unique(mb, dim=0)))
        if how == "error":
            raise RuntimeError(msg)
        if how == "warn":
            import warnings

            warnings.warn(msg)

    def _load_motion(self, env):
        lib = env.get("motion_lib")
        if isinstance(lib, MotionLib):
            return lib
        if "synthetic_motions" in env:
            from .. import synth

            spec = dict(env["synthetic_motions"])
            clips = synth.make_clips(spec.get("seed", 7), spec.get("num_clips", 64), spec.get("min_frames", 90), spec.get("max_frames", 300),
                                     spec.get("speed", 1.0))
            per_clip = self.body_shapes is not None and len(self.body_shapes) == len(clips) and "motion_shape_ids" not in env
            return MotionLib.from_clips(clips, self.body_shapes if per_clip else self.body_model, self.device)
        path = env.get("motion_file")
        if path and os.path.isfile(path) and path.endswith(".v2pm"):  # the flat, memory-mappable library file (motion_tables.save_flat)
            return MotionLib.from_flat_file(path, self.device)
        if path and os.path.isfile(path) and path.endswith(".npz"):
            with np.load(path) as z:
                return MotionLib({k: z[k] for k in z.files}, self.device)
        if path and (os.path.isdir(path) or (os.path.isfile(path) and path.endswith(".pth"))):
            # the reference's pickled MotionLib parts (humanoid_smpl_im.py:420-440), through the restricted unpickler
            from ..legacy_motion_lib import load_legacy_motion_lib

            return load_legacy_motion_lib(path, self.device, env.get("motion_file_range"))
        raise RuntimeError("no motion library: pass cfg['env']['motion_lib'] (MotionLib), 'synthetic_motions', a flat .npz / .v2pm 'motion_file' or the "
                           "reference's .pth file / directory of mlib_part_*.pth")

    def _allocate_buffers(self):
        n, dev = self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        i = dict(dtype=torch.long, device=dev)
        self.obs_buf = torch.zeros((n, self.num_obs), **f)
        self.states_buf = torch.zeros((n, self.num_states), **f)
        self.rew_buf = torch.zeros(n, **f)
        self.reset_buf = torch.ones(n, **i)
        self.progress_buf = torch.zeros(n, **i)
        self.randomize_buf = torch.zeros(n, **i)
        self._terminate_buf = torch.ones(n, **i)
        self._sub_rewards = torch.zeros((n, 4), **f)
        # the tensors gym.acquire_*_tensor would have returned (humanoid_smpl.py:66-113)
        self._root_states = torch.zeros((n, 13), **f)
        self._humanoid_root_states = self._root_states
        # the actors' start pose (humanoid_smpl.py:348-353: char_h 0.89, identity rotation), velocities zeroed (:89-90): what
        # _reset_default restores
        self._root_states[:, 2] = 0.89
        self._root_states[:, 6] = 1.0
        self._initial_humanoid_root_states = self._root_states.clone()
        self._dof_state = torch.zeros((n * self._num_dof, 2), **f)
        ds = self._dof_state.view(n, self._num_dof, 2)
        self._dof_pos, self._dof_vel = ds[..., 0], ds[..., 1]
        self._rigid_body_state = torch.zeros((n * self.num_bodies, 13), **f)
        rb = self._rigid_body_state.view(n, self.num_bodies, 13)
        # bodies of the freshly created actors: the start pose with every joint at zero (a default-pose reset leaves the rigid-body
        # tensor as it is until the next simulate(), like gym.refresh_rigid_body_state_tensor does: at creation it holds this pose)
        bm0 = self.body_model
        if bm0 is not None:
            pos = np.zeros((self.num_bodies, 3), dtype=np.float32)
            pos[0] = (0.0, 0.0, 0.89)
            for b in range(1, self.num_bodies):
                pos[b] = pos[int(bm0.parents[b])] + np.asarray(bm0.local_pos[b], dtype=np.float32)
            rb[:, :, 0:3] = torch.as_tensor(pos, **f)
            rb[:, :, 6] = 1.0
        self._rigid_body_pos, self._rigid_body_rot = rb[..., 0:3], rb[..., 3:7]
        self._rigid_body_vel, self._rigid_body_ang_vel = rb[..., 7:10], rb[..., 10:13]
        self._contact_forces = torch.zeros((n, self.num_bodies, 3), **f)
        self.dof_force_tensor = torch.zeros((n, self._num_dof), **f)
        self._pd_target = torch.zeros((n, self._num_dof), **f)
        self._cur_ref_motion_times = torch.zeros(n, **f)
        self._reset_ref_motion_times = torch.zeros(n, **f)
        self._target_bufs = [torch.zeros((n, _lib.MOTION_STATE_DIM), **f) for _ in range(2)]
        w = self.context_length + 2 * self.context_padding
        self.context_feat = torch.zeros((n, w, _lib.CONTEXT_DIM), **f)
        self._context_mask_u8 = torch.zeros((n, w), dtype=torch.uint8, device=dev)
        self._humanoid_actor_ids = torch.arange(n, device=dev, dtype=torch.int32)
        self._prev_dof_pos = None

    def _build_termination_heights(self):
        """humanoid_smpl_im.py:217-224"""
        env = self.cfg["env"]
        th = np.array([env.get("terminationBodyHeight", -0.5)] * self.num_bodies, dtype=np.float64)
        head = self.body_names.index("Head")
        self._humanoid_head_id = head
        th[head] = max(env.get("terminationHeadHeight", 1.0), th[head])
        self._termination_heights = torch.tensor(th, dtype=torch.float32, device=self.device)

    def _create_engine(self):
        lib = _lib.load()
        env, sp, bm = self.cfg["env"], self.sim_params, self.body_model
        keep = []

        def farr(x):
            a = np.ascontiguousarray(x, dtype=np.float32)
            keep.append(a)
            return a.ctypes.data_as(_lib.c_f)

        def iarr(x):
            a = np.ascontiguousarray(x, dtype=np.int32)
            keep.append(a)
            return a.ctypes.data_as(_lib.c_i32)

        def create_model(m):
            d = _lib.ModelDesc(num_bodies=m.num_bodies, parents=iarr(m.parents), local_pos=farr(m.local_pos), mass=farr(m.mass),
                               com=farr(m.com), inertia=farr(m.inertia), kp=farr(m.kp), kd=farr(m.kd), armature=farr(m.armature),
                               hull_offsets=iarr(m.hull_offsets), hull_verts=farr(m.hull_verts),
                               limit_lower=farr(m.limit_lower), limit_upper=farr(m.limit_upper))
            h = C.c_void_p()
            _lib.check(lib.v2p_model_create(C.byref(d), self.device_id, C.byref(h)), "v2p_model_create")
            return h

        self._h_models = [create_model(m) for m in (self.body_shapes if self._env_shape_ids is not None else [bm])]
        self._h_model = self._h_models[0]
        c = _lib.SimCfg()
        c.sim_dt = sp.dt
        c.substeps = sp.substeps
        c.control_freq_inv = self.control_freq_inv
        c.enable_contact = int(env.get("enable_contact", True))
        c.freeze_terminated_envs = int(env.get("freeze_terminated_envs", False))  # not the reference's behaviour: see v2p_rollout.h
        c.schedule = {"link_per_lane": 0, "env_per_lane": 1}[env.get("kernel_schedule", "link_per_lane")]
        c.pair_envs_by_load = int(env.get("pair_envs_by_load", True))
        # the sim.physx block, solver choice included (sim.physx.solver_type 1 of amass_im.yaml:41 = TGS; env.contact_solver overrides;
        # the engine's TGS restates the published algorithm with frozen Jacobians, see oracle/phys/v2p_phys_oracle.c)
        self.contact_solver, self.contact_solver_source = fill_physx(c, sp, env, log=lambda m: print(m, flush=True))
        # physics launch cut into (substep, env pair) jobs: finer load balancing, bit-identical results (tests); on by default
        # (True / 1: the engine cuts launches that do not fit the wave slots in one round; 2: always; every solver / contact setting)
        c.substep_jobs = int(env.get("substep_jobs", True)) if c.schedule == 0 else 0
        c.job_mono_permille = int(env.get("job_mono_permille", -1))  # -1: the engine's defaults
        c.pair_mix_permille = int(env.get("pair_mix_permille", -1))
        c.kernel_build = int(env.get("kernel_build", 0))  # 0: the engine chooses by the envs resident on the device
        # A/B and test switches of the substep jobs (v2p_sim_cfg, ABI 13): 0 = the engine's defaults
        c.job_timeout_spins = int(env.get("job_timeout_spins", 0))
        c.job_len = int(env.get("job_len", 0))
        c.job_lead = int(env.get("job_lead", 0))
        c.job_no_interleave = int(env.get("job_no_interleave", 0))
        # tangent frame of the hull x ground friction rows (v2p_sim_cfg.friction_frame, ABI 14): "world" (x / y, the default) | "velocity"
        ff = env.get("friction_frame", "world")
        if ff not in ("world", "velocity"):
            raise ValueError("env.friction_frame = %r: 'world' or 'velocity'" % (ff,))
        c.friction_frame = {"world": 0, "velocity": 1}[ff]
        self.friction_frame = ff
        # joint ranges of the MJCF enforced as limit rows (Isaac Gym always enforces them; only the racket arm of the player MJCFs has
        # DOFs narrower than a full turn, the amass MJCF has none)
        c.joint_limits = int(env.get("joint_limits", False))
        c.limit_margin = float(env.get("limit_margin", -1.0))  # radians within which a limit row exists (-1: the engine's default, 0.05)
        c.debug_contacts = int(env.get("debug_contacts", 0))  # 0 off, 1 last substep's contact vertices kept, 2 every substep's
        hold = env.get("residual_force_hold", "first_sim")
        c.residual_hold_sims = 1 if hold == "first_sim" else self.control_freq_inv
        c.gravity_z = sp.gravity[2]
        c.friction = 0.5 * (env["plane"]["staticFriction"] + env["plane"]["dynamicFriction"]) if "plane" in env else 1.0
        c.erp = env.get("contact_erp", 0.2)
        c.angular_damping = 0.01
        c.max_angular_velocity = 100.0
        c.pd_tar_lim = self.pd_tar_lim
        c.residual_force_scale = self.residual_force_scale
        c.residual_torque_scale = self.residual_torque_scale
        c.ground_tolerance = self.ground_tolerance
        c.max_episode_length = self.max_episode_length
        c.enable_early_termination = int(self._enable_early_termination)
        c.context_length = self.context_length
        c.context_padding = self.context_padding
        th = self._termination_heights.cpu().numpy().copy()
        th[self._contact_body_ids.cpu().numpy()] = -np.inf  # fall_height[:, contact_body_ids] = False (:970)
        c.term_heights[:] = th.tolist()
        c.body_pos_weights[:] = self.body_pos_weights.cpu().numpy().tolist()
        specs = {"k_dof": 60, "k_vel": 0.2, "k_pos": 100, "k_rot": 40, "w_dof": 0.6, "w_vel": 0.1, "w_pos": 0.2, "w_rot": 0.1}
        specs.update(env.get("reward_specs", dict()))
        c.reward_specs[:] = [specs[k] for k in ("k_dof", "k_vel", "k_pos", "k_rot", "w_dof", "w_vel", "w_pos", "w_rot")]
        self.reward_specs = specs
        b = _lib.EnvBuffers()
        b.root_states = self._root_states.data_ptr()
        b.dof_state = self._dof_state.data_ptr()
        b.rb_state = self._rigid_body_state.data_ptr()
        b.contact_force = self._contact_forces.data_ptr()
        b.dof_force = self.dof_force_tensor.data_ptr()
        b.pd_target = self._pd_target.data_ptr()
        b.obs = self.obs_buf.data_ptr()
        b.rew = self.rew_buf.data_ptr()
        b.sub_rewards = self._sub_rewards.data_ptr()
        b.reset = self.reset_buf.data_ptr()
        b.terminate = self._terminate_buf.data_ptr()
        b.progress = self.progress_buf.data_ptr()
        b.cur_time = self._cur_ref_motion_times.data_ptr()
        b.reset_time = self._reset_ref_motion_times.data_ptr()
        b.target[0] = self._target_bufs[0].data_ptr()
        b.target[1] = self._target_bufs[1].data_ptr()
        b.context_feat = self.context_feat.data_ptr()
        b.context_mask = self._context_mask_u8.data_ptr()
        self._h_env = C.c_void_p()
        if self._env_shape_ids is None:
            _lib.check(lib.v2p_env_create(self._h_model, self._motion_lib.handle(), C.byref(c), _lib.ptr(self._reset_ref_motion_ids),
                                          self.num_envs, C.byref(b), self.device_id, C.byref(self._h_env)), "v2p_env_create")
        else:
            hs = (C.c_void_p * len(self._h_models))(*[h.value for h in self._h_models])
            _lib.check(lib.v2p_env_create_shapes(hs, len(self._h_models), iarr(self._env_shape_ids), self._motion_lib.handle(), C.byref(c),
                                                 _lib.ptr(self._reset_ref_motion_ids), self.num_envs, C.byref(b), self.device_id,
                                                 C.byref(self._h_env)), "v2p_env_create_shapes")
        self._lib = lib
        self._cur = 0
        self._motion_lib._borrowed = True  # the batch keeps the table pointers: no merge from now on

    def close(self):
        if getattr(self, "_h_env", None):
            self._lib.v2p_env_destroy(self._h_env)
            self._h_env = None
        for h in getattr(self, "_h_models", None) or []:
            self._lib.v2p_model_destroy(h)
        self._h_models = []
        self._h_model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference surface
    def get_obs_size(self):
        return self._num_obs

    def get_action_size(self):
        return self._num_actions

    def get_states(self):
        return self.states_buf

    def register_model(self, model):
        self.model = model

    def pre_epoch(self, epoch):
        return

    def render_vis(self, init=False):
        return

    def render(self, sync_frame_time=False):
        return

    def get_aux_losses(self, model_res_dict):
        """humanoid_smpl_im.py:694-722 with empty aux_loss_specs (amass_im / djokovic_im)."""
        if self.cfg["env"].get("aux_loss_specs"):
            raise NotImplementedError("aux_loss_specs are not built")
        return {}, {}

    @property
    def context_mask(self):
        return self._context_mask_u8.bool()

    def _stream(self):
        return _lib.current_stream(self.device)

    def reset(self, env_ids=None):
        """HumanoidSMPL.reset (humanoid_smpl.py:136-159) with reference-state init."""
        if env_ids is None:
            # once per epoch, without a wait: substep jobs that had to be recomputed (never, normally) are noticed and warned about
            _lib.check(self._lib.v2p_env_check_async(self._h_env, self._stream()), "v2p_env_check_async")
            self._warn_job_recoveries()
            n = self.num_envs
            ids_t, motion_ids = None, self._reset_ref_motion_ids
        else:
            ids_t = env_ids.to(device=self.device, dtype=torch.long).contiguous()
            n = ids_t.shape[0]
            if n == 0:
                return
            motion_ids = self._reset_ref_motion_ids[ids_t]
        # _reset_actors (humanoid_smpl_im.py:470-480): default pose, reference state, or a Bernoulli mix of the two (:638-651)
        if self._state_init == HumanoidSMPLIM.StateInit.Default:
            self._reset_default(ids_t)
            return
        if self._state_init == HumanoidSMPLIM.StateInit.Hybrid and self._hybrid_init_prob < 1.0:
            all_ids = torch.arange(self.num_envs, device=self.device, dtype=torch.long) if ids_t is None else ids_t
            ref_mask = torch.bernoulli(torch.full((n,), float(self._hybrid_init_prob), device=self.device)) == 1.0
            ref_ids, def_ids = all_ids[ref_mask].contiguous(), all_ids[~ref_mask].contiguous()
            if def_ids.numel() > 0:
                self._reset_default(def_ids)
            if ref_ids.numel() == 0:
                return
            ids_t, n, motion_ids = ref_ids, ref_ids.shape[0], self._reset_ref_motion_ids[ref_ids]
        if self._state_init == HumanoidSMPLIM.StateInit.Start:
            times = torch.zeros(n, device=self.device)
        else:
            trunc = self.context_length * self.dt if self.truncate_time else None
            times = self._motion_lib.sample_time(motion_ids, truncate_time=trunc).to(self.device)
        self.reset_with_times(ids_t, times)

    def _reset_default(self, env_ids):
        """_reset_default (humanoid_smpl_im.py:482-487) + _reset_env_tensors (humanoid_smpl.py:161-173) + the observation of the reset
        envs (:153-158): root at the actor's start pose, joints and velocities at zero, pushed into the engine; progress / reset /
        terminate flags cleared.  Like the reference, nothing else moves: the clip time, the target and the context of these envs stay,
        and the rigid-body tensor keeps showing the bodies of before the reset until the next physics step (Isaac Gym only refreshes it
        from the last simulate()), so the observation of a default-reset env mixes new joint coordinates with old body poses."""
        ids = slice(None) if env_ids is None else env_ids
        self._humanoid_root_states[ids] = self._initial_humanoid_root_states[ids]
        self._dof_pos[ids] = 0.0
        self._dof_vel[ids] = 0.0
        self._reset_env_tensors(env_ids)
        self.progress_buf[ids] = 0
        self.reset_buf[ids] = 0
        self._terminate_buf[ids] = 0
        self._reset_default_env_ids = env_ids
        # _compute_humanoid_obs on the reset envs (:653-668; layout :198): plain copies of the exposed tensors
        k = self.num_envs if env_ids is None else env_ids.shape[0]
        self.obs_buf[ids] = torch.cat([self._rigid_body_pos[ids].reshape(k, -1), self._rigid_body_rot[ids].reshape(k, -1), self._dof_pos[ids],
                                       self._dof_vel[ids], self._rigid_body_vel[ids].reshape(k, -1), self._rigid_body_ang_vel[ids].reshape(k, -1),
                                       self._reset_ref_motion_bodies[ids][:, :11]], dim=1)

    def reset_with_times(self, env_ids, motion_times):
        """Reference-state init at explicit clip times (parity tests; _reset_ref_state_init :489-528)."""
        times = motion_times.to(device=self.device, dtype=torch.float32).contiguous()
        n = self.num_envs if env_ids is None else env_ids.shape[0]
        _lib.check(self._lib.v2p_env_reset(self._h_env, _lib.ptr(env_ids), n, _lib.ptr(times), self._stream()), "v2p_env_reset")
        self._reset_ref_env_ids = env_ids
        self._forward_context()

    def _forward_context(self):
        """The tail of the reference's _init_context (humanoid_smpl_im.py:557-563): a registered model is told the new window."""
        if self.model is not None:
            if not self.is_env_dim_setup:
                self.model.a2c_network.setup_env_named_dims(self.obs_names, self.obs_shapes, self.obs_dims, self.context_names,
                                                            self.context_shapes, self.context_dims)
                self.is_env_dim_setup = True
            with torch.no_grad():
                self.model.a2c_network.forward_context(self.context_feat, self.context_mask)

    def _init_context(self, motion_ids, motion_times):
        """HumanoidSMPLIM._init_context (humanoid_smpl_im.py:530-563) on its own: context_feat / context_mask rebuilt around
        `motion_times` [num_envs] (frames motion_times + dt * (1 - padding ... length + padding)), nothing else touched.  The reference's
        player calls it every context_length steps with (task._reset_ref_motion_ids, task._cur_ref_motion_times) (players/im_player.py:
        238-240).  Like the reference's (its `.view(self.num_envs, ...)`) it takes all envs; `motion_ids` must be the envs' own clips -
        the engine samples each env from the clip it was created with."""
        if motion_ids is not self._reset_ref_motion_ids:
            ids = motion_ids.to(device=self.device, dtype=torch.long)
            if ids.shape != self._reset_ref_motion_ids.shape or not torch.equal(ids, self._reset_ref_motion_ids):
                raise RuntimeError("_init_context: motion_ids must be the task's _reset_ref_motion_ids (one entry per env)")
        times = motion_times.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(times.shape) != (self.num_envs,):
            raise RuntimeError("_init_context: motion_times must have one entry per env")
        _lib.check(self._lib.v2p_env_context(self._h_env, None, self.num_envs, _lib.ptr(times), self._stream()), "v2p_env_context")
        self._forward_context()

    def step(self, actions):
        """BaseTask.step (base_task.py:147-165).  `actions` [N,75] fp32 on this device; rows of envs whose
        reset flag is set are zeroed in place like the reference does (humanoid_smpl_im.py:126)."""
        if self.record_pd_torque or type(self).pre_physics_step is not HumanoidSMPLIM.pre_physics_step or \
                type(self)._physics_step is not HumanoidSMPLIM._physics_step or type(self).post_physics_step is not HumanoidSMPLIM.post_physics_step:
            self.pre_physics_step(actions)  # a subclass hooks a stage (or the PD torque is being logged): run the stages one by one
            self._physics_step()
            self.post_physics_step()
        else:
            self.step_fused(actions)  # one C call: pre-physics inside the physics kernel, then post-physics

    def _check_actions(self, actions):
        if actions.dtype != torch.float32 or not actions.is_contiguous() or str(actions.device) != self.device or tuple(actions.shape) != (self.num_envs, self.num_actions):
            raise RuntimeError("actions must be a contiguous float32 [%d,%d] tensor on %s" % (self.num_envs, self.num_actions, self.device))

    def pre_physics_step(self, actions):
        self._check_actions(actions)
        if self.record_pd_torque:
            self._prev_dof_pos = self._dof_pos.clone()
        _lib.check(self._lib.v2p_env_pre_physics(self._h_env, _lib.ptr(actions), self._stream()), "v2p_env_pre_physics")
        self.actions = actions

    def _physics_step(self):
        _lib.check(self._lib.v2p_env_physics(self._h_env, self._stream()), "v2p_env_physics")
        _lib.check(self._lib.v2p_env_export(self._h_env, self._stream()), "v2p_env_export")

    def post_physics_step(self):
        _lib.check(self._lib.v2p_env_post_physics(self._h_env, self._stream()), "v2p_env_post_physics")
        self._cur = 1 - self._cur
        self.extras["terminate"] = self._terminate_buf
        self.extras["sub_rewards"] = self._sub_rewards
        self.extras["sub_rewards_names"] = self._sub_rewards_names

    def step_fused(self, actions):
        """One C call for the whole step (what bench.py times)."""
        self._check_actions(actions)
        _lib.check(self._lib.v2p_env_step(self._h_env, _lib.ptr(actions), self._stream()), "v2p_env_step")
        self._cur = 1 - self._cur
        self.actions = actions
        self.extras["terminate"] = self._terminate_buf
        self.extras["sub_rewards"] = self._sub_rewards
        self.extras["sub_rewards_names"] = self._sub_rewards_names

    def _reset_env_tensors(self, env_ids=None, with_rb_state=False):
        """set_actor_root_state_tensor_indexed + set_dof_state_tensor_indexed (humanoid_smpl.py:161-173):
        push edits made through `_humanoid_root_states/_dof_pos/_dof_vel` into the engine."""
        n = self.num_envs if env_ids is None else env_ids.shape[0]
        _lib.check(self._lib.v2p_env_push_state(self._h_env, _lib.ptr(env_ids), n, int(with_rb_state), self._stream()), "v2p_env_push_state")

    def set_schedule(self, name):
        """'link_per_lane' (default) or 'env_per_lane': two GPU schedules of the same physics model."""
        kind = {"link_per_lane": 0, "env_per_lane": 1}[name]
        _lib.check(self._lib.v2p_env_set_schedule(self._h_env, kind), "v2p_env_set_schedule")

    def debug_contacts(self):
        out = torch.empty((self.num_envs, self.num_bodies, 4), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.v2p_env_debug_contacts(self._h_env, _lib.ptr(out), self._stream()), "v2p_env_debug_contacts")
        return out

    def debug_contacts_substeps(self):
        """[N, substeps of a control step, 24, 4] contact vertex ids of every substep of the last step (cfg env debug_contacts = 2)."""
        nsub = self.sim_params.substeps * self.control_freq_inv
        out = torch.empty((self.num_envs, nsub, self.num_bodies, 4), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.v2p_env_debug_contacts_substeps(self._h_env, _lib.ptr(out), self._stream()), "v2p_env_debug_contacts_substeps")
        return out

    def check(self):
        """Synchronise (v2p_env_check: raises on a HIP error) and fetch the substep jobs' recovery counter."""
        _lib.check(self._lib.v2p_env_check(self._h_env, self._stream()), "v2p_env_check")
        self._warn_job_recoveries()

    def kernel_build(self):
        """Which of the library's two builds of the physics kernel this batch runs (v2p_sim_cfg.kernel_build)."""
        return {1: "lds-parked, 3 waves per SIMD", 2: "registers, 2 waves per SIMD"}[int(self._lib.v2p_env_kernel_build(self._h_env))]

    def job_recoveries(self):
        """Substep jobs that gave up waiting for their predecessor and recomputed the earlier substeps themselves (as last fetched by
        check() or by the per-epoch reset()).  Results are unaffected; a count that grows means the launch loses time."""
        n = C.c_int64(0)
        _lib.check(self._lib.v2p_env_job_recoveries(self._h_env, C.byref(n)), "v2p_env_job_recoveries")
        return int(n.value)

    def jobs_skipped(self):
        """Late substep jobs that found their pair's step complete and were skipped (as last fetched): the per-call records those jobs own
        were not published for that step.  Never observed; check() raises on it, the per-epoch reset() too."""
        n = C.c_int64(0)
        _lib.check(self._lib.v2p_env_jobs_skipped(self._h_env, C.byref(n)), "v2p_env_jobs_skipped")
        return int(n.value)

    def _warn_job_recoveries(self):
        k = self.jobs_skipped()
        if k > getattr(self, "_jobs_skipped_seen", 0):
            self._jobs_skipped_seen = k
            raise RuntimeError("%d substep job(s) of the physics launches started after their env pair's step was complete and were skipped: exposed PD targets, "
                               "in-place action masking and the ball's per-call records of those steps are missing (cfg env substep_jobs=False avoids it)" % k)
        n = self.job_recoveries()
        if n > getattr(self, "_job_recoveries_seen", 0):
            import warnings
            warnings.warn("%d substep jobs of the physics launches were recomputed after waiting in vain for their predecessor (workgroups not "
                          "dispatched in index order?): results are unaffected, the launches lose time; cfg env substep_jobs=False avoids it" % n)
            self._job_recoveries_seen = n

    def profile_begin(self, max_launches, stride=1, period=1):
        """HIP events around the physics-kernel launches from now on (engine side, on the launch stream): all of them, or launch L when
        L % stride == (L // period) % stride (period = steps per epoch: every position of the epoch once in `stride` epochs)."""
        _lib.check(self._lib.v2p_env_profile_begin_sampled(self._h_env, int(max_launches), int(stride), int(period)), "v2p_env_profile_begin_sampled")

    def profile_end(self):
        """(summed physics-kernel milliseconds, launches measured) since profile_begin; synchronises."""
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        _lib.check(self._lib.v2p_env_profile_end(self._h_env, C.byref(ms), C.byref(cnt)), "v2p_env_profile_end")
        return ms.value, cnt.value

    def debug_pairing(self):
        """(perm, key): wave-slot -> env order of the last physics launch and the contact-load key of each env after it."""
        perm = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        key = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.v2p_env_debug_pairing(self._h_env, _lib.ptr(perm), _lib.ptr(key), self._stream()), "v2p_env_debug_pairing")
        return perm, key

    # ------------------------------------------------------------------ target views
    _SLICES = {"root_pos": (0, 3, None), "root_rot": (3, 7, None), "dof_pos": (7, 76, None), "root_vel": (76, 79, None),
               "root_ang_vel": (79, 82, None), "dof_vel": (82, 151, None), "key_pos": (151, 163, (4, 3)), "rb_pos": (163, 235, (24, 3)),
               "rb_rot": (235, 331, (24, 4))}

    def _tview(self, which, name):
        a, b, shp = self._SLICES[name]
        t = self._target_bufs[which][:, a:b]
        return t.view(self.num_envs, *shp) if shp else t

    @property
    def pd_torque(self):
        if self._prev_dof_pos is None:
            raise RuntimeError("set cfg['env']['record_pd_torque']=True to keep the pre-step dof positions")
        return (self._pd_target - self._prev_dof_pos) * self.stiffness


def _add_target_properties():
    for name in HumanoidSMPLIM._SLICES:
        setattr(HumanoidSMPLIM, "_target_" + name, property(lambda self, n=name: self._tview(self._cur, n)))
        setattr(HumanoidSMPLIM, "_prev_target_" + name, property(lambda self, n=name: self._tview(1 - self._cur, n)))


_add_target_properties()
