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
    """The policy's observation normaliser (models/running_norm.py:5-43): y = clamp((x - mean) / (std + 1e-8)) with running estimates;
    buffers `n mean var std` as in the reference's module (so its checkpoints load), training mode takes the batch into the
    statistics BEFORE normalising it, nothing is normalised while n == 0.

    The statistics are kept the way a data-parallel learner needs them: a batch enters as its sufficient statistics (count, sum, sum of
    squares in float64) - one all-reduce merges the batches of all ranks - and is folded into the running moments with the pairwise
    update of Chan et al. (mean += delta m / (n + m); M2 += M2_batch + delta^2 n m / (n + m)).  In exact arithmetic this is the
    reference's weighted form (running_norm.py:22-31: biased batch variance, weight n / (n + m)); pinned to vectors recorded from the
    reference's module (tests/test_running_norm.py).  Plain torch on whatever device the buffers live on: the statistics of a batch
    depend on the whole batch, so the training-mode pass cannot be fused into the per-row observation kernel; eval mode is what
    `ImitationObs` fuses."""

    def __init__(self, dim, demean=True, destd=True, clip=5.0, device=None):
        self.dim, self.demean, self.destd, self.clip = int(dim), demean, destd, clip
        self.n = torch.zeros((), dtype=torch.long, device=device)
        self.mean = torch.zeros(dim, device=device)
        self.var = torch.zeros(dim, device=device)
        self.std = torch.zeros(dim, device=device)
        self.training = True
        self._seen = False  # host-side "n > 0" (None = unknown: ask the device once)

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        return {"n": self.n, "mean": self.mean, "var": self.var, "std": self.std}

    def load_state_dict(self, sd):
        for k in ("n", "mean", "var", "std"):
            getattr(self, k).copy_(torch.as_tensor(sd[k]).to(getattr(self, k).device))
        self._seen = None

    @torch.no_grad()
    def update(self, x, group=None):
        """x [m, dim]: the batch of this rank; with `group` (or an initialised default group of more than one rank) the batches of all
        ranks enter as one."""
        import torch.distributed as dist

        x64 = x.detach().double()
        stats = torch.cat([x64.sum(0), (x64 * x64).sum(0), torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(stats, group=group)
        d = self.dim
        m = stats[2 * d]
        ok = (m > 0).double()
        msafe = torch.clamp(m, min=1.0)
        mean_x = stats[:d] / msafe
        m2_x = torch.clamp(stats[d:2 * d] - msafe * mean_x * mean_x, min=0.0)  # sum of squared deviations of the batch
        n = self.n.double()
        tot = torch.clamp(n + m, min=1.0)
        delta = mean_x - self.mean.double()
        mean = self.mean.double() + ok * delta * (m / tot)
        m2 = self.var.double() * n + ok * (m2_x + delta * delta * (n * m / tot))
        self.mean.copy_(mean.float())
        self.var.copy_((m2 / tot).float())
        self.std.copy_(torch.sqrt(self.var))
        self.n += m.long()
        # (from the REDUCED count, like the reference's `n > 0` test: a rank whose own batch was empty must normalise like its peers;
        # None = re-read from n at the next normalize(), no host sync here)
        self._seen = True if x.shape[0] > 0 else None

    def normalize(self, x):
        if self._seen is None:
            self._seen = int(self.n) > 0
        if self._seen:
            if self.demean:
                x = x - self.mean
            if self.destd:
                x = x / (self.std + 1e-8)
            if self.clip:
                x = torch.clamp(x, -self.clip, self.clip)
        return x

    def __call__(self, x, group=None):
        if self.training:
            self.update(x, group)
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
