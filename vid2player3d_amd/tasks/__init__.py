from .humanoid_smpl_im import HumanoidSMPLIM, SimParams, default_cfg  # noqa: F401
from .humanoid_racket_ball import HumanoidSMPLIMRacketBall  # noqa: F401
