"""The reference's own `ImitatorBuilder.Network` (embodied_pose/models/im_network_builder.py:28-245) with small MLPs, built without rl_games'
A2CBuilder (absent): the attributes its __init__ would set for cfg/amass_im.yaml are set by hand.  Shared by the golden generators
(oracle/gen_golden_ppo.py, oracle/gen_golden_player.py).  TEST INFRASTRUCTURE; needs ref_shim.install + rl_games_restated.register first."""
import numpy as np
import torch
import torch.nn as nn


def mlp(inp, units):
    layers, d = [], inp
    for u in units:
        layers += [nn.Linear(d, u), nn.ReLU()]
        d = u
    return nn.Sequential(*layers)


def build_reference_network(units, pad, ops):
    """`ops`: tests/golden/task_ops.npz (running statistics of a model that has trained for a while)."""
    import models.im_network_builder as NB
    from models.running_norm import RunningNorm

    t32 = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)  # noqa: E731
    net = NB.ImitatorBuilder.Network.__new__(NB.ImitatorBuilder.Network)
    nn.Module.__init__(net)
    net.context_padding, net.humanoid_obs_dim, net.residual_action = pad, 734, True
    net.use_running_obs, net.running_obs_type, net.use_ik = True, "ours", False
    net.running_obs = RunningNorm(734)
    net.is_continuous, net.is_discrete, net.is_multi_discrete = True, False, False
    net.space_config = {"fixed_sigma": True, "learn_sigma": False}
    net.actor_cnn, net.critic_cnn = nn.Sequential(), nn.Sequential()
    net.actor_mlp, net.critic_mlp = mlp(734, units), mlp(734, units)
    net.mu, net.value = nn.Linear(units[-1], 75), nn.Linear(units[-1], 1)
    net.mu_act, net.sigma_act, net.value_act = nn.Identity(), nn.Identity(), nn.Identity()
    net.sigma = nn.Parameter(torch.full((75,), -1.756), requires_grad=False)
    with torch.no_grad():  # (outputs of a size that makes the clipped / unclipped branches of the losses both occur)
        net.mu.weight.mul_(0.3)
        net.value.weight.mul_(3.0)
    nb = 24
    shape_dict = {"body_pos": (nb, 3), "body_pos_gt": (nb, 3), "body_rot": (nb, 4), "dof_pos": (69,), "dof_pos_gt": (69,), "dof_vel": (69,),
                  "body_vel": (nb, 3), "body_ang_vel": (nb, 3), "motion_bodies": (11,)}
    obs_names = ["body_pos", "body_rot", "dof_pos", "dof_vel", "body_vel", "body_ang_vel", "motion_bodies"]  # humanoid_smpl_im.py:198
    ctx_names = ["body_pos", "body_rot", "dof_pos", "body_pos_gt", "dof_pos_gt"]                              # :202
    net.setup_env_named_dims(obs_names, [shape_dict[x] for x in obs_names], [int(np.prod(shape_dict[x])) for x in obs_names],
                             ctx_names, [shape_dict[x] for x in ctx_names], [int(np.prod(shape_dict[x])) for x in ctx_names])
    # running statistics of a model that has trained for a while (n > 0: the eval-mode forward of the rollout normalises)
    net.running_obs.n += 1000
    net.running_obs.mean[:] = t32(ops["rn_mean"])
    net.running_obs.std[:] = t32(ops["rn_std"])
    net.running_obs.var[:] = net.running_obs.std ** 2
    return net
