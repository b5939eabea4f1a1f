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


def test_racket_ball_task_falls_back_to_pgs_when_the_file_says_tgs():
    """vid2player's tennis configs state solver_type: 1; the racket-arm limit rows and the ball exist in the PGS solver only: the task
    runs PGS and says so (it used to fail in v2p_env_create) - unless the caller names a solver by the engine's own key."""
    from vid2player3d_amd.tasks.humanoid_racket_ball import racket_ball_solver_override

    sp = SimParams.from_cfg(yaml.safe_load(AMASS_IM_SIM)["sim"])
    env, said = {}, []
    assert racket_ball_solver_override(env, sp, log=said.append) and env["contact_solver"] == "pgs"
    assert len(said) == 1 and "TGS" in said[0] and "PGS" in said[0]
    env = {"contact_solver": "tgs"}  # an explicit engine-side choice is left alone (and refused by the engine with its own message)
    assert not racket_ball_solver_override(env, sp) and env["contact_solver"] == "tgs"
    assert not racket_ball_solver_override({}, SimParams.from_cfg(default_cfg(4)["sim"]))  # no solver named: PGS anyway, nothing to say


def test_zero_initialised_cfg_keeps_the_job_defaults():
    """ABI 13: the substep-job switches are cfg fields (no environment variables); zero = the engine's defaults."""
    c = _lib.SimCfg()
    assert (c.job_timeout_spins, c.job_len, c.job_lead, c.job_no_interleave, c.kernel_build) == (0, 0, 0, 0, 0)
