"""The imitation task with vid2player's racket and ball in the simulation (BASELINE config 4's "racket+ball contacts").

`vid2player/env/tasks/humanoid_smpl_im_mvae.py` puts two actors in every env: the SMPL humanoid with a racket welded to the right
wrist (`smpl_mesh_humanoid_djokovic.xml:188-190`) and a free tennis ball (`tennis_ball.urdf`), applies the aerodynamic force of
`apply_external_force_to_ball` (:711-739) before every `simulate()` call and polls the net contact forces after it (:752-783).  What
this class takes from it is that PHYSICS: the observation / reward / MVAE machinery of the vid2player task is out of scope,
the task logic stays the imitation task's (`HumanoidSMPLIM`).  Exposed like the reference's tensors:

    _ball_root_states [N,13]     the ball actor's root state; write it (+ nothing else) to launch a ball, as `_reset_balls` does (:503-522)
    _racket_rb_state  [N,13]     rigid body 24
    _ball_states_per_sim [N,2,13], _racket_ball_contact_per_sim [N,2]   what the reference sees after each of the 2 simulate() calls
    _has_bounce / _has_bounce_now / _bounce_pos, _has_racket_ball_contact(_now)   the flags of :731-737 and :773-779, same rules
    _contact_forces_sum [N,24,3]  net contact forces summed over the simulate() calls of a control step (:781)
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib, racket
from ..model import load_baked_model
from .humanoid_smpl_im import HumanoidSMPLIM

BALL_R = racket.BALL["radius"]


class HumanoidSMPLIMRacketBall(HumanoidSMPLIM):
    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True):
        # (work on a copy: the caller's cfg stays what it was, a second task built from it starts from the plain body model again)
        cfg = dict(cfg)
        env = cfg["env"] = dict(cfg["env"])
        base = env.get("body_model") or load_baked_model(default_humanoid_mass=env.get("default_humanoid_mass", 90.0), kp_scale=env.get("kp_scale", 1.0),
                                                         kd_scale=env.get("kd_scale", env.get("kp_scale", 1.0)))
        if env.get("has_racket_collision", False):
            # (humanoid_smpl_im_mvae.py:42, 397-401: racket shapes with collision filter 0 = the racket collides with the links of its own
            # humanoid; default False in every config)
            raise NotImplementedError("has_racket_collision=True (racket x link contacts) is not built; the reference default is False")
        # the player asset: djokovic / federer (right hand) or nadal (left hand); cfg_v2p righthand = False selects the left-handed one
        # like the reference (humanoid_smpl_im_mvae.py:73-78).  The exposed rigid-body order is the canonical one either way (racket =
        # rigid body 24: the reference permutes the left-handed asset's tensor into it, :67, 197-201).
        v2p = dict(cfg.get("v2p") or {})
        player = env.get("player", "djokovic" if v2p.get("righthand", True) else "nadal")
        if isinstance(base, (list, tuple)):
            # one body shape per clip: the racket is the same object in every hand - welded at the same offset of the wrist frame - so
            # every shape gets it folded in, and the ball's cylinders (given in the wrist frame) are shared
            folded = [racket.with_racket(b, player=player) for b in base]
            model, self.racket_geometry = [m for m, _ in folded], folded[0][1]
        else:
            model, self.racket_geometry = racket.with_racket(base, player=player)
        env["body_model"] = model
        # the player MJCF's racket-arm ranges (R_Wrist +-10 / +-45 / +-90 deg, R_Elbow_x <= 90 deg) are enforced like Isaac Gym does
        env.setdefault("joint_limits", True)
        # (the solver is the one the files name: vid2player/cfg/im/tennis_im.yaml:39, embodied_pose/cfg/djokovic_im.yaml:41 state
        # `solver_type: 1`, TGS - the racket-arm limit rows and the ball's rows are solved inside its slices like the hull rows;
        # `env.contact_solver` overrides as in the base task)
        self.cfg_v2p = dict(cfg.get("v2p") or {})
        super().__init__(cfg, sim_params, physics_engine, device_type, device_id, headless)
        n, dev = self.num_envs, self.device
        nsim = self.control_freq_inv
        f = dict(dtype=torch.float32, device=dev)
        self._ball_root_states = torch.zeros((n, 13), **f)
        self._ball_root_states[:, 2] = 1.0   # start_pose of the ball actor (:430-431)
        self._ball_root_states[:, 6] = 1.0
        self._racket_rb_state = torch.zeros((n, 13), **f)
        self._ball_states_per_sim = torch.zeros((n, nsim, 13), **f)
        self._racket_ball_contact_per_sim = torch.zeros((n, nsim), dtype=torch.int32, device=dev)
        self._ball_contact_forces = torch.zeros((n, 2, 3), **f)
        self._ball_body_contact_force = torch.zeros((n, 3), **f)  # on the ball from the humanoid's links (ball x hull contacts)
        # `_contact_forces_sum` (:186, 690, 781): net contact forces of the links after each simulate() call, summed over the control step
        # (opt-in, cfg env contact_forces_sum: nothing in the reference reads it, and keeping it costs the launch 2 %)
        self._contact_forces_sum = torch.zeros((n, 24, 3), **f) if env.get("contact_forces_sum", False) else None
        self._has_bounce = torch.zeros(n, dtype=torch.bool, device=dev)
        self._has_bounce_now = torch.zeros(n, dtype=torch.bool, device=dev)
        self._bounce_pos = torch.zeros((n, 3), **f)
        self._has_racket_ball_contact = torch.zeros(n, dtype=torch.bool, device=dev)
        self._has_racket_ball_contact_now = torch.zeros(n, dtype=torch.bool, device=dev)
        mat = dict(racket.BALL_MATERIAL)
        if "restitution" in self.cfg_v2p:  # cfg_v2p.restitution sets ball AND racket head (:414, :436); the plane keeps 0
            mat["rest_ground"], mat["rest_racket"] = 0.5 * self.cfg_v2p["restitution"], self.cfg_v2p["restitution"]
            mat["rest_body"] = 0.5 * self.cfg_v2p["restitution"]  # ball x a link's hull: the humanoid's shapes keep restitution 0
        if "ball_friction" in self.cfg_v2p or "racket_friction" in self.cfg_v2p:
            bf, rf = self.cfg_v2p.get("ball_friction", 0.8), self.cfg_v2p.get("racket_friction", 0.8)
            mat["fric_ground"], mat["fric_racket"], mat["fric_body"] = 0.5 * (bf + 1.0), 0.5 * (bf + rf), 0.5 * (bf + 1.0)
        g = self.racket_geometry
        c = _lib.BallCfg(radius=racket.BALL["radius"], mass=racket.BALL["mass"], inertia=racket.BALL["inertia"],
                         restitution_ground=mat["rest_ground"], friction_ground=mat["fric_ground"], restitution_racket=mat["rest_racket"],
                         friction_racket=mat["fric_racket"], bounce_threshold_velocity=self.sim_params.physx.bounce_threshold_velocity,
                         angular_damping=mat["ang_damp"], max_angular_velocity=mat["max_ang_vel"], spin_scale=self.cfg_v2p.get("spin_scale", 1.0),
                         racket_link=g["racket_link"], num_cylinders=len(g["cylinders"]),
                         restitution_body=mat["rest_body"], friction_body=mat["fric_body"], body_contacts=int(env.get("ball_body_contacts", True)),
                         bounce_height=BALL_R * (6 if self.sim_params.substeps > 2 else 4), poll_racket_hits=int(self.sim_params.substeps <= 2))
        for k, cy in enumerate(g["cylinders"]):
            c.cylinders[k][:] = [float(x) for x in list(cy["center"]) + list(cy["axis"]) + [cy["half_len"], cy["radius"]]]
        c.racket_offset[:] = [float(x) for x in g["racket_offset"]]
        b = _lib.BallBuffers(ball_state=self._ball_root_states.data_ptr(), racket_state=self._racket_rb_state.data_ptr(),
                             ball_per_sim=self._ball_states_per_sim.data_ptr(), racket_hit_per_sim=self._racket_ball_contact_per_sim.data_ptr(),
                             ball_contact=self._ball_contact_forces.data_ptr(), ball_body_contact=self._ball_body_contact_force.data_ptr(),
                             has_bounce=self._has_bounce.data_ptr(), has_bounce_now=self._has_bounce_now.data_ptr(), bounce_pos=self._bounce_pos.data_ptr(),
                             has_racket_contact=self._has_racket_ball_contact.data_ptr(), has_racket_contact_now=self._has_racket_ball_contact_now.data_ptr(),
                             contact_force_sum=None if self._contact_forces_sum is None else self._contact_forces_sum.data_ptr())
        _lib.check(self._lib.v2p_env_attach_ball(self._h_env, C.byref(c), C.byref(b)), "v2p_env_attach_ball")
        self.ball_material = mat

    # ------------------------------------------------------------------ the reference's flag bookkeeping around the physics step
    def reset_balls(self, env_ids, launch_pos, launch_vel, launch_ang_vel):
        """`_reset_balls` (:503-522) with the launch state given by the caller (the reference draws it from its trajectory generator)."""
        ids = torch.as_tensor(env_ids, device=self.device, dtype=torch.long)
        self._ball_root_states[ids, 0:3] = launch_pos
        self._ball_root_states[ids, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=self.device)
        self._ball_root_states[ids, 7:10] = launch_vel
        self._ball_root_states[ids, 10:13] = launch_ang_vel
        self._has_bounce[ids] = False
        self._bounce_pos[ids] = 0
        self._has_racket_ball_contact[ids] = False

    # The reference's flag bookkeeping around every simulate() call - the bounce test on the ball height at the START of the call
    # (apply_external_force_to_ball, :731-737: threshold 4 ball radii, 6 with more than 2 substeps) and the contact-force poll after it
    # (:773-779, only with sim.substeps <= 2) - runs inside the physics launch: the flag tensors above are the buffers it writes.
