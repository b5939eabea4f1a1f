"""The handful of rl-games helpers the reference's PPO code calls, RESTATED (test infrastructure only).

The reference pins `rl-games==1.1.4` (/root/reference/install.sh:2).  The package is third-party, absent from /root/reference and from
this image (no network), so the reference's `ImitatorAgent` methods cannot run against the real thing here.  What follows restates the
published algorithm of the few functions those methods call, from the package's public source as I know it [from memory - PARITY OF
THESE HELPERS IS UNPINNED]; each one is anchored on a call site or a twin inside /root/reference:

  apply_masks       embodied_pose/agents/im_agent.py:537 `torch_ext.apply_masks([...], mask=alive)`; the divisor is the NUMBER OF ELEMENTS
                    of the mask, not its sum - the reference's own masked KL two lines further down uses exactly that form
                    (`(kl_dist * alive).sum() / alive.numel()`, im_agent.py:573), mirroring rl_games' a2c_continuous.
  policy_kl         im_agent.py:572; KL(N(p0) || N(p1)) per sample with rl_games' 1e-5 guards.
  mean_list         learning/common_agent.py:192.
  RunningMeanStd    the value normaliser `self.value_mean_std` (im_agent.py:292, 301, 426-429): count starts at 1, mean 0, var 1,
                    epsilon 1e-5 INSIDE the square root, the batch variance is torch's default (unbiased) one, `unnorm=True` clamps its
                    INPUT to +-5 before scaling back.
  ExperienceBuffer  update_data / tensor_dict / get_transformed_list as play_steps uses them (im_agent.py:320-400).
  ModelA2CContinuousLogStd.Network.neglogp    called by the reference's own models/im_models.py:31, 46.
  network_builder.A2CBuilder(.Network)        base classes of models/im_network_builder.py:28-33 (only `nn.Module` behaviour is used by the
                                              methods the goldens call: eval_actor / eval_critic / forward, :200-245).

`register()` places them in sys.modules under their rl_games names BEFORE the reference modules are imported (the generic attribute-sink
stub of `install.py` serves every other rl_games name).
"""
import sys
import types

import numpy as np
import torch
import torch.nn as nn


def apply_masks(losses, mask=None):
    sum_mask = None
    if mask is not None:
        mask = mask.unsqueeze(1)
        sum_mask = mask.numel()
        res_losses = [(l * mask).sum() / sum_mask for l in losses]
    else:
        res_losses = [torch.mean(l) for l in losses]
    return res_losses, sum_mask


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma, reduce=True):
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    c3 = -1.0 / 2.0
    kl = c1 + c2 + c3
    kl = kl.sum(dim=-1)
    return kl.mean() if reduce else kl


def mean_list(val):
    return torch.mean(torch.stack(val))


def shape_whc_to_cwh(shape):
    if len(shape) == 3:
        return (shape[2], shape[0], shape[1])
    return shape


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        self.insize, self.epsilon, self.norm_only, self.per_channel = insize, epsilon, norm_only, per_channel
        self.axis = [0]
        in_size = insize
        self.register_buffer("running_mean", torch.zeros(in_size, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(in_size, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))

    @staticmethod
    def _update_mean_var_count_from_moments(mean, var, count, batch_mean, batch_var, batch_count):
        delta = batch_mean - mean
        tot_count = count + batch_count
        new_mean = mean + delta * batch_count / tot_count
        m_a = var * count
        m_b = batch_var * batch_count
        M2 = m_a + m_b + delta ** 2 * count * batch_count / tot_count
        new_var = M2 / tot_count
        return new_mean, new_var, tot_count

    def forward(self, input, unnorm=False):
        if self.training:
            mean = input.mean(self.axis)
            var = input.var(self.axis)
            self.running_mean, self.running_var, self.count = self._update_mean_var_count_from_moments(
                self.running_mean, self.running_var, self.count, mean, var, input.size()[0])
        current_mean, current_var = self.running_mean, self.running_var
        if unnorm:
            y = torch.clamp(input, min=-5.0, max=5.0)
            y = torch.sqrt(current_var.float() + self.epsilon) * y + current_mean.float()
        elif self.norm_only:
            y = input / torch.sqrt(current_var.float() + self.epsilon)
        else:
            y = (input - current_mean.float()) / torch.sqrt(current_var.float() + self.epsilon)
            y = torch.clamp(y, min=-5.0, max=5.0)
        return y


class ExperienceBuffer:
    """only what play_steps touches: a dict of [T, N, ...] tensors written row by row"""

    def __init__(self, tensor_dict):
        self.tensor_dict = tensor_dict

    def update_data(self, name, index, val):
        if type(val) is dict:
            for k, v in val.items():
                self.tensor_dict[name][k][index, :] = v
        else:
            self.tensor_dict[name][index, :] = val

    def get_transformed_list(self, transform_op, tensor_list):
        res_dict = {}
        for k in tensor_list:
            v = self.tensor_dict.get(k)
            if v is None:
                continue
            res_dict[k] = transform_op(v)
        return res_dict


class _BaseModel:
    def __init__(self, network=None):
        self.network_builder = network


class ModelA2CContinuousLogStd(_BaseModel):
    class Network(nn.Module):
        def __init__(self, a2c_network):
            nn.Module.__init__(self)
            self.a2c_network = a2c_network

        def neglogp(self, x, mean, std, logstd):
            return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * np.log(2.0 * np.pi) * x.size()[-1] + logstd.sum(dim=-1)


class _A2CBuilder:
    def __init__(self, **kwargs):
        pass

    class Network(nn.Module):
        def __init__(self, params=None, **kwargs):
            nn.Module.__init__(self)


def register():
    """sys.modules entries for the rl_games names above; call after ref_shim.install.install() and before importing the reference."""
    import importlib

    def mod(name):
        importlib.import_module(name)  # (the sink stub creates the package chain)
        return sys.modules[name]

    te = mod("rl_games.algos_torch.torch_ext")
    for f in (apply_masks, policy_kl, mean_list, shape_whc_to_cwh):
        setattr(te, f.__name__, f)
    setattr(mod("rl_games.algos_torch.running_mean_std"), "RunningMeanStd", RunningMeanStd)
    setattr(mod("rl_games.algos_torch.models"), "ModelA2CContinuousLogStd", ModelA2CContinuousLogStd)
    setattr(mod("rl_games.algos_torch.network_builder"), "A2CBuilder", _A2CBuilder)
    setattr(mod("rl_games.common.experience"), "ExperienceBuffer", ExperienceBuffer)
    return types.SimpleNamespace(torch_ext=te)
