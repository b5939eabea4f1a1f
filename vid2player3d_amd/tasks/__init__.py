from .humanoid_smpl_im import HumanoidSMPLIM, SimParams, default_cfg  # noqa: F401
