"""The reference's `sim` block at the drop-in boundary (no GPU): every key of sim.physx either reaches v2p_sim_cfg or is refused there;
the solver the file names is the solver the engine is told to run."""
import pytest
import yaml

from vid2player3d_amd import _lib
from vid2player3d_amd.tasks import SimParams, default_cfg
from vid2player3d_amd.tasks.humanoid_smpl_im import fill_physx, resolve_contact_solver

# embodied_pose/cfg/amass_im.yaml:37-52, verbatim (djokovic_im.yaml's sim block is the same)
AMASS_IM_SIM = """
sim:
  substeps: 2
  physx:
    num_threads: 4
    solver_type: 1
    num_position_iterations: 4
    num_velocity_iterations: 0
    contact_offset: 0.02
    rest_offset: 0.0
    bounce_threshold_velocity: 0.2
    max_depenetration_velocity: 10.0
    default_buffer_size_multiplier: 10.0

  flex:
    num_inner_iterations: 10
    warm_start: 0.25
"""


def test_the_references_sim_block_selects_tgs():
    sim = yaml.safe_load(AMASS_IM_SIM)["sim"]
    sp = SimParams.from_cfg(sim)
    c = _lib.SimCfg()
    name, source = fill_physx(c, sp, env={})
    assert (name, source) == ("tgs", "sim.physx.solver_type") and c.solver_type == 1
    assert c.num_solver_iterations == 4 and c.num_velocity_iterations == 0
    assert abs(c.contact_offset - 0.02) < 1e-9 and c.rest_offset == 0.0 and abs(c.bounce_threshold_velocity - 0.2) < 1e-7
    assert c.max_depenetration_velocity == 10.0 and sp.substeps == 2
    sim["physx"]["solver_type"] = 0
    assert fill_physx(_lib.SimCfg(), SimParams.from_cfg(sim), env={})[0] == "pgs"


def test_env_contact_solver_overrides_and_says_so():
    sp = SimParams.from_cfg(yaml.safe_load(AMASS_IM_SIM)["sim"])
    said = []
    c = _lib.SimCfg()
    assert fill_physx(c, sp, env={"contact_solver": "pgs"}, log=said.append) == ("pgs", "env.contact_solver") and c.solver_type == 0
    assert len(said) == 1 and "overrides sim.physx.solver_type = 1" in said[0]
    said.clear()
    assert resolve_contact_solver({"contact_solver": "tgs"}, sp, log=said.append)[0] == "tgs" and not said  # no disagreement, nothing to say
    with pytest.raises(ValueError):
        resolve_contact_solver({"contact_solver": "jacobi"}, sp)


def test_a_block_that_names_no_solver_gets_the_engine_default():
    """default_cfg() (this package's own defaults, the bench's configuration) names no solver type: PGS, the solver BASELINE config 3 names."""
    sp = SimParams.from_cfg(default_cfg(4)["sim"])
    assert "solver_type" not in sp.given and resolve_contact_solver({}, sp) == ("pgs", "engine default")
    assert resolve_contact_solver({}, SimParams()) == ("pgs", "engine default")


def test_a_foreign_sim_params_object_is_taken_at_its_word():
    """What `parse_sim_params` (utils/config.py:190-222) returns is a gymapi.SimParams: no record of which keys a file named, and its
    solver_type is 1 before any yaml is read (config.py:203)."""
    class Physx:
        solver_type, num_position_iterations, num_velocity_iterations = 1, 4, 0
        contact_offset, rest_offset, bounce_threshold_velocity, max_depenetration_velocity = 0.02, 0.0, 0.2, 10.0

    class Foreign:
        dt, substeps, physx = 1.0 / 60.0, 2, Physx()

    c = _lib.SimCfg()
    assert fill_physx(c, Foreign(), env={}) == ("tgs", "sim.physx.solver_type") and c.solver_type == 1


def test_velocity_iterations_are_refused_by_the_library_not_dropped():
    """num_velocity_iterations != 0 travels to v2p_env_create, which refuses it (V2P_ERR_UNSUPPORTED); the check itself needs no GPU work
    but sits behind the buffer checks of v2p_env_create, so here only the forwarding is asserted (the refusal: tests/test_gpu_vec_task.py)."""
    sim = yaml.safe_load(AMASS_IM_SIM)["sim"]
    sim["physx"]["num_velocity_iterations"] = 1
    sim["physx"]["rest_offset"] = 0.005
    c = _lib.SimCfg()
    fill_physx(c, SimParams.from_cfg(sim), env={})
    assert c.num_velocity_iterations == 1 and abs(c.rest_offset - 0.005) < 1e-9


def test_racket_ball_task_keeps_the_solver_the_file_names():
    """vid2player's tennis configs state solver_type: 1 (tennis_im.yaml:39, djokovic_im.yaml:41).  Until round 5 the racket + ball task
    overrode that to PGS; the override is gone (the limit rows and the ball's rows are solved under TGS too): the task module resolves the
    solver exactly like the base task, and nothing in it rewrites env.contact_solver."""
    import inspect

    from vid2player3d_amd.tasks import humanoid_racket_ball as rb

    assert not hasattr(rb, "racket_ball_solver_override")
    src = inspect.getsource(rb.HumanoidSMPLIMRacketBall.__init__)
    assert 'env["contact_solver"]' not in src and "contact_solver\"] =" not in src
    sp = SimParams.from_cfg(yaml.safe_load(AMASS_IM_SIM)["sim"])
    assert resolve_contact_solver({"joint_limits": True}, sp) == ("tgs", "sim.physx.solver_type")


def test_zero_initialised_cfg_keeps_the_job_defaults():
    """ABI 13: the substep-job switches are cfg fields (no environment variables); zero = the engine's defaults."""
    c = _lib.SimCfg()
    assert (c.job_timeout_spins, c.job_len, c.job_lead, c.job_no_interleave, c.kernel_build) == (0, 0, 0, 0, 0)
