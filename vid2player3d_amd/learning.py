"""Host-side mirrors of the two learner-side ops that sit right after the env boundary, with the reference's
argument meaning, forwarding to the HIP library (no CPU path):

  * `ImitationObs` — `ImitatorBuilder.Network.preprocess_input` + `compute_humanoid_obs` + `running_obs`
    (models/im_network_builder.py:150-189, env/tasks/humanoid_smpl_im.py:773-850, models/running_norm.py:32-43): the 734-d
    in-network observation computed straight from the packed 461-d obs rows and the context frames, RunningNorm (eval) fused.
  * `RunningNorm` — models/running_norm.py:5-43 including the training-mode update of the statistics.
  * `discount_values` — `CommonAgent.discount_values` (learning/common_agent.py:423-435), the GAE reverse scan.
"""
import torch

from . import _lib

OBS_IMITATION_DIM = 734


def _chk(t, shape_tail, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda or tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise RuntimeError("%s must be a contiguous float32 CUDA tensor [..., %s]" % (name, ", ".join(map(str, shape_tail))))


class RunningNorm:
    """models/running_norm.py:5-43: y = clamp((x - mean) / (std + 1e-8)) with running estimates.  Same buffers (n mean var std), same
    update rule and the same order (training mode updates the statistics with the batch BEFORE normalising it; nothing is normalised
    while n == 0).  Plain torch on whatever device the buffers live on: the statistics of a batch depend on the whole batch, so the
    training-mode pass cannot be fused into the per-row observation kernel; eval mode is what `ImitationObs` fuses."""

    def __init__(self, dim, demean=True, destd=True, clip=5.0, device=None):
        self.dim, self.demean, self.destd, self.clip = int(dim), demean, destd, clip
        self.n = torch.zeros((), dtype=torch.long, device=device)
        self.mean = torch.zeros(dim, device=device)
        self.var = torch.zeros(dim, device=device)
        self.std = torch.zeros(dim, device=device)
        self.training = True

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    @torch.no_grad()
    def update(self, x):
        var_x, mean_x = torch.var_mean(x, dim=0, unbiased=False)
        m = x.shape[0]
        w = self.n.to(x.dtype) / (m + self.n).to(x.dtype)
        self.var[:] = w * self.var + (1 - w) * var_x + w * (1 - w) * (mean_x - self.mean).pow(2)
        self.mean[:] = w * self.mean + (1 - w) * mean_x
        self.std[:] = torch.sqrt(self.var)
        self.n += m

    def normalize(self, x):
        if int(self.n) > 0:
            if self.demean:
                x = x - self.mean
            if self.destd:
                x = x / (self.std + 1e-8)
            if self.clip:
                x = torch.clamp(x, -self.clip, self.clip)
        return x

    def __call__(self, x):
        if self.training:
            self.update(x)
        return self.normalize(x)


class ImitationObs:
    """context_padding as in cfg (amass_im.yaml: 8).  `mean` / `std` [734] are the buffers of the policy's RunningNorm
    (`Network.running_obs.mean/.std`; None = raw features, like `use_running_obs=False`), `clip` its clamp (5.0)."""

    def __init__(self, context_padding=8, mean=None, std=None, clip=5.0):
        self.context_padding = int(context_padding)
        # RunningNorm.forward clamps only `if self.clip` (running_norm.py:41-42): None / 0 = no clamp
        self.clip = float(clip) if clip else float("inf")
        self.set_running_stats(mean, std)
        self._lib = _lib.load()

    @classmethod
    def from_running_norm(cls, running_norm, context_padding=8):
        """running_norm: the reference's models.running_norm.RunningNorm in eval mode (or anything with .mean .std .clip [.n]).
        A fresh model (n == 0) does not normalise at all (running_norm.py:36): raw features, like mean = std = None."""
        n = getattr(running_norm, "n", None)
        if n is not None and int(n) == 0:
            return cls(context_padding, None, None, running_norm.clip)
        if not (getattr(running_norm, "demean", True) and getattr(running_norm, "destd", True)):
            raise NotImplementedError("RunningNorm with demean=False or destd=False is not built (the reference's networks use both)")
        return cls(context_padding, running_norm.mean, running_norm.std, running_norm.clip)

    def set_running_stats(self, mean, std):
        if (mean is None) != (std is None):
            raise ValueError("mean and std go together")
        for name, t in (("mean", mean), ("std", std)):
            if t is not None and tuple(t.shape) != (OBS_IMITATION_DIM,):
                raise ValueError("%s must have shape [%d], got %s" % (name, OBS_IMITATION_DIM, tuple(t.shape)))
        self._mean = None if mean is None else mean.detach().float().contiguous()
        self._std = None if std is None else std.detach().float().contiguous()

    def _stats_on(self, device):
        """the statistics on the device of the observations (RunningNorm buffers of a model still on the CPU are moved, never passed
        to the kernel as host pointers)"""
        if self._mean is not None and self._mean.device != device:
            self._mean, self._std = self._mean.to(device), self._std.to(device)
        return self._mean, self._std

    def _run(self, obs, context_feat, steps, first_frame):
        rows = obs.shape[0]
        out = torch.empty((rows, OBS_IMITATION_DIM), dtype=torch.float32, device=obs.device)
        stream = torch.cuda.current_stream(obs.device).cuda_stream
        mean, std = self._stats_on(obs.device)
        _lib.check(self._lib.v2p_obs_imitation_packed(rows, steps, _lib.ptr(obs), _lib.ptr(context_feat), context_feat.shape[1], first_frame,
                                                      _lib.ptr(mean), _lib.ptr(std), self.clip, _lib.ptr(out), stream),
                   "v2p_obs_imitation_packed")
        return out

    def rollout(self, obs, context_feat, t):
        """eval / rollout flavour (flatten=False): obs [N,461], context_feat [N,L,378], step t of the epoch -> [N,734]."""
        _chk(obs, (_lib.NUM_OBS,), "obs")
        _chk(context_feat, (378,), "context_feat")
        return self._run(obs, context_feat, 1, self.context_padding + int(t))

    def training(self, obs, context_feat, running_norm=None):
        """training flavour (flatten=True): obs [N,T,461] (or [N*T,461]), context_feat [N,L,378] -> [N*T,734].
        running_norm: a `RunningNorm` (or the reference's module) in TRAINING mode - its statistics are updated with this batch's raw
        features first, then the batch is normalised with them (running_norm.py:32-43); the kernel then produces the raw features and
        the normalisation is the module's own.  None: the statistics given at construction, fused into the kernel (eval mode)."""
        _chk(obs, (_lib.NUM_OBS,), "obs")
        _chk(context_feat, (378,), "context_feat")
        n = context_feat.shape[0]
        flat = obs.reshape(-1, _lib.NUM_OBS)
        if flat.shape[0] % n:
            raise RuntimeError("obs rows %d are not a multiple of the %d envs of context_feat" % (flat.shape[0], n))
        if running_norm is None:
            return self._run(flat, context_feat, flat.shape[0] // n, self.context_padding)
        keep = self._mean, self._std
        self._mean = self._std = None
        try:
            raw = self._run(flat, context_feat, flat.shape[0] // n, self.context_padding)
        finally:
            self._mean, self._std = keep
        return running_norm(raw)


def discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values, gamma, tau):
    """CommonAgent.discount_values (learning/common_agent.py:423-435) with the agent's gamma / tau passed in (the method reads
    self.gamma, self.tau, self.horizon_length): mb_fdones [T,N], the others [T,N,1] (or [T,N]) float32 CUDA tensors; returns
    mb_advs like mb_rewards."""
    for name, t in (("mb_fdones", mb_fdones), ("mb_values", mb_values), ("mb_rewards", mb_rewards), ("mb_next_values", mb_next_values)):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise RuntimeError("%s must be a contiguous float32 CUDA tensor" % name)
    horizon = mb_rewards.shape[0]
    n = mb_rewards[0].numel()
    if mb_fdones.numel() != horizon * n or mb_values.numel() != horizon * n or mb_next_values.numel() != horizon * n:
        raise RuntimeError("discount_values: tensors disagree on [T,N]")
    advs = torch.empty_like(mb_rewards)
    lib = _lib.load()
    stream = torch.cuda.current_stream(mb_rewards.device).cuda_stream
    _lib.check(lib.v2p_gae(horizon, n, _lib.ptr(mb_fdones), _lib.ptr(mb_values), _lib.ptr(mb_rewards), _lib.ptr(mb_next_values),
                           float(gamma), float(tau), _lib.ptr(advs), stream), "v2p_gae")
    return advs


class DiscountValuesMixin:
    """Drop-in for CommonAgent.discount_values on an agent object that has .gamma and .tau (same 4-argument signature)."""

    def discount_values(self, mb_fdones, mb_values, mb_rewards, mb_next_values):
        return discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values, self.gamma, self.tau)
