"""The PPO loop of the imitation task around the rollout engine, method for method what the reference's agent does:

    get_action_values / _eval_critic    embodied_pose/agents/im_agent.py:271-303
    play_steps                          :305-409
    prepare_dataset / _calc_advs        :411-473
    calc_gradients                      :475-587  (losses: learning/common_agent.py:442-450, 491-520)
    train_epoch                         learning/common_agent.py:146-216, minibatches as learning/amp_datasets.py:17-35

with the network of cfg/amass_im.yaml (`ImitatorNetwork` = models/im_network_builder.py:28-245 for that config + models/im_models.py:20-58)
and these differences in HOW, none in WHAT (pinned to vectors recorded from the reference's own methods: oracle/gen_golden_ppo.py,
tests/test_gpu_ppo_reference.py):

  * the experience buffer is resident on the GPU as [T, N, ...] tensors written in place; nothing inside the 32-step rollout reads a value
    back to the host: the reference's `.nonzero()` bookkeeping of finished episodes (:366-386) and its `torch.all(self.dones == 1)` early
    exit (:388) become masked device reductions read once per epoch (finished envs are masked out of every statistic and of the loss by
    `alive`, so running the remaining steps changes no result.  Where the reference leaves the loop early, the rows it did not reach keep
    whatever the previous epoch wrote, `dones` included; here they hold dones = 1),
  * between the actor MLP and the env step sits ONE kernel (`v2p_policy_head`): residual action (mu[:, :69] += the context's target DOF
    positions, im_network_builder.py:226-228), sample, neglogp; the 734-d in-network observation + RunningNorm (eval) is the fused HIP
    kernel (`learning.ImitationObs`), the GAE reverse scan the HIP kernel `v2p_gae`,
  * the critic runs ONCE per step: the reference evaluates it on the observation after step n for `next_values` (`_eval_critic`, t = n + 1)
    and again on the very same observation, same t, same eval-mode weights for `values` of step n + 1 (`get_action_values`); the second
    evaluation is the first one's result (`reuse_next_values`; bit-identical, tested).  And it runs BESIDE the physics (`overlap_critic`):
    the value of an observation goes to the buffer only - the next env step needs the actor alone - so the critic pass of step n + 1 is
    issued on a side stream and fills the SIMDs the latency-bound physics launch of step n + 1 leaves idle (same numbers, tested),
  * ROLLOUT GROUPS: the agent takes one task or several (env batches of one GPU).  Each group runs its steps on its own HIP stream: while
    the physics launch of one group is in flight the MLP passes of the other run, and each fills the tail of the other's launches.  Envs
    are independent, so the buffer an N-env rollout fills is the same whether N envs form one group or two (tested bit for bit),
  * data parallel over env shards (one process per GPU): advantages are normalised with GLOBAL masked statistics (3-number all-reduce;
    the reference's Horovod ranks normalise locally), the observation / value normalisers merge the batches of all ranks, gradients are
    averaged in one flat all-reduce over RCCL.

The policy / value networks are dense MLPs (cfg/amass_im.yaml:86-88: [1024, 1024, 512], relu; fixed sigma exp(-1.756), :76-81) through
torch (rocBLAS): they are NOT part of the accelerated path.  rl_games itself is not rebuilt; the pieces of it the reference's methods call
(value normaliser, masked means, policy KL, neglogp) are restated below from its published source [rl-games 1.1.4, from memory].
"""
import math
import time

import os

import torch
import torch.nn as nn

from . import _lib
from . import dist as vdist
from .learning import OBS_IMITATION_DIM, ImitationObs, RunningNorm, discount_values

LOG_2PI = math.log(2.0 * math.pi)
NUM_DOF = 69
CTX_DOF_POS = 168  # offset of `dof_pos` inside a 378-d context frame (body_pos 72 | body_rot 96 | dof_pos 69 | ..., humanoid_smpl_im.py:202)


def _world(group):
    return vdist.dist.get_world_size(group) if vdist.dist.is_available() and vdist.dist.is_initialized() else 1


def _mlp(inp, units):
    layers, d = [], inp
    for u in units:
        layers += [nn.Linear(d, u), nn.ReLU()]
        d = u
    return nn.Sequential(*layers)


def _mlp_forward(seq, x):
    """The MLP of `_mlp`.  Inference on the GPU (no autograd): every Linear + ReLU pair as ONE GEMM with bias + ReLU in its epilogue
    (`torch._addmm_activation` -> hipBLASLt): bit-identical to Linear followed by ReLU, one elementwise kernel per layer less
    (6 x 10 us per rollout step at 8192 envs, profiles/r05_play_steps_kernels.txt)."""
    if torch.is_grad_enabled() or not x.is_cuda or x.dim() != 2:
        return seq(x)
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and m.bias is not None:
            x = torch._addmm_activation(m.bias, x, m.weight.t())
            i += 2
        else:
            x = m(x)
            i += 1
    return x


def neglogp(x, mu, sigma, logstd):
    """rl_games ModelA2CContinuousLogStd.neglogp (called at models/im_models.py:31, 46)"""
    return 0.5 * (((x - mu) / sigma) ** 2).sum(dim=-1) + 0.5 * LOG_2PI * x.shape[-1] + logstd.sum(dim=-1)


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    """rl_games torch_ext.policy_kl(..., reduce=False) (im_agent.py:572): KL(N(p0) || N(p1)) per sample, with its 1e-5 guards"""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    return (c1 + c2 - 0.5).sum(dim=-1)


def masked_mean(x, mask):
    """rl_games torch_ext.apply_masks (im_agent.py:537): sum of the alive entries over the NUMBER OF ELEMENTS of the mask (not its sum) -
    the form the reference's own masked KL uses two lines further down (:573)."""
    return (x * mask).sum() / mask.numel()


class ValueMeanStd:
    """The value normaliser `self.value_mean_std` (rl_games RunningMeanStd((1,)); im_agent.py:292, 301, 426-429): count starts at 1 with
    mean 0 / var 1, epsilon 1e-5 inside the square root, a batch enters with torch's default (unbiased) variance, `unnorm` clamps its
    input to +-5 before scaling back.  Data parallel: the moments of a batch are taken over the batches of all ranks (sum / sum of
    squares / count in one all-reduce), so every rank keeps the same normaliser."""

    def __init__(self, device, epsilon=1e-5):
        self.running_mean = torch.zeros(1, dtype=torch.float64, device=device)
        self.running_var = torch.ones(1, dtype=torch.float64, device=device)
        self.count = torch.ones((), dtype=torch.float64, device=device)
        self.epsilon = epsilon
        self.training = False

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    @torch.no_grad()
    def update(self, x, group=None):
        x64 = x.detach().double().reshape(-1)
        s = torch.stack([x64.sum(), (x64 * x64).sum(), torch.tensor(float(x64.numel()), dtype=torch.float64, device=x.device)])
        if _world(group) > 1:
            vdist.dist.all_reduce(s, group=group)
        m = s[2]
        mean = s[0] / m
        var = torch.clamp(s[1] - m * mean * mean, min=0.0) / torch.clamp(m - 1.0, min=1.0)
        delta = mean - self.running_mean
        tot = self.count + m
        self.running_mean = self.running_mean + delta * m / tot
        self.running_var = (self.running_var * self.count + var * m + delta * delta * self.count * m / tot) / tot
        self.count = tot

    def __call__(self, x, unnorm=False, group=None):
        if self.training:
            self.update(x, group)
        scale = torch.sqrt(self.running_var.float() + self.epsilon)
        if unnorm:
            return scale * torch.clamp(x, min=-5.0, max=5.0) + self.running_mean.float()
        return torch.clamp((x - self.running_mean.float()) / scale, min=-5.0, max=5.0)


class ImitatorNetwork(nn.Module):
    """`ImitatorBuilder.Network` for cfg/amass_im.yaml (separate actor / critic MLPs over the 734-d in-network observation, RunningNorm
    "ours", residual action, fixed sigma) with the parameter names of the reference's module, so that `load_reference_state_dict` takes a
    reference checkpoint's `model` entry (`a2c_network.actor_mlp.0.weight` ...)."""

    def __init__(self, num_actions=75, units=(1024, 1024, 512), sigma_init=-1.756, residual_action=True, device=None):
        super().__init__()
        self.actor_mlp, self.critic_mlp = _mlp(OBS_IMITATION_DIM, units), _mlp(OBS_IMITATION_DIM, units)
        self.mu, self.value = nn.Linear(units[-1], num_actions), nn.Linear(units[-1], 1)
        self.sigma = nn.Parameter(torch.full((num_actions,), float(sigma_init)), requires_grad=False)  # fixed_sigma, learn_sigma False
        self.residual_action = residual_action
        self.to(device)
        self.running_obs = RunningNorm(OBS_IMITATION_DIM, device=device)

    def actor(self, x):
        return self.mu(_mlp_forward(self.actor_mlp, x))

    def critic(self, x):
        return self.value(_mlp_forward(self.critic_mlp, x))

    def load_reference_state_dict(self, sd, prefix="a2c_network."):
        own = {k: v for k, v in sd.items() if k.startswith(prefix)}
        self.load_state_dict({k[len(prefix):]: torch.as_tensor(v) for k, v in own.items() if "running_obs" not in k}, strict=True)
        self.running_obs.load_state_dict({k.split("running_obs.")[1]: v for k, v in own.items() if "running_obs." in k})

    def forward_train(self, feat_raw, target_dof_pad, prev_actions, group=None):
        """ImitatorModel.Network.forward with is_train (im_models.py:27-41) on the minibatch's RAW in-network features: the running
        statistics take them in (training mode, running_norm.py:33-34), then normalise them; residual mean; neglogp of the old actions."""
        x = self.running_obs(feat_raw, group) if self.training else self.running_obs.normalize(feat_raw)
        mu = self.actor(x)
        if self.residual_action:
            mu = mu + target_dof_pad
        logstd = mu * 0.0 + self.sigma
        sigma = torch.exp(logstd)
        entropy = (0.5 + 0.5 * LOG_2PI + logstd).sum(dim=-1)  # Normal(mu, sigma).entropy().sum(-1)
        return {"prev_neglogp": neglogp(prev_actions, mu, sigma, logstd), "values": self.critic(x), "entropy": entropy, "mus": mu, "sigmas": sigma}


class ExperienceBuffer:
    """[T, N, ...] device tensors, written in place step by step (rl_games ExperienceBuffer.update_data / tensor_dict)."""

    def __init__(self, horizon, num_envs, obs_dim, act_dim, device):
        f = dict(dtype=torch.float32, device=device)
        # obses[n + 1] and next_obses[n] are the same observation (no env resets inside an epoch): ONE tensor of horizon + 1 rows, two
        # views - the rollout writes every observation once (15 MB per step at 8192 envs) and the buffer is 0.5 GB smaller
        self._obs_all = torch.zeros((horizon + 1, num_envs, obs_dim), **f)
        self.tensor_dict = {
            "obses": self._obs_all[:horizon], "next_obses": self._obs_all[1:],
            "dones": torch.zeros((horizon, num_envs), **f),
            "rewards": torch.zeros((horizon, num_envs, 1), **f), "values": torch.zeros((horizon, num_envs, 1), **f),
            "next_values": torch.zeros((horizon, num_envs, 1), **f), "actions": torch.zeros((horizon, num_envs, act_dim), **f),
            "neglogpacs": torch.zeros((horizon, num_envs), **f), "mus": torch.zeros((horizon, num_envs, act_dim), **f),
            "sigmas": torch.zeros((horizon, num_envs, act_dim), **f),
        }

    def update_data(self, name, n, value, sl=slice(None)):
        dst = self.tensor_dict[name][n, sl]
        dst.copy_(value.reshape(dst.shape))


class _Group:
    """one env batch of the rollout: a task, its env range in the buffer, its stream"""

    def __init__(self, task, lo, stream, side):
        self.task, self.lo, self.hi, self.stream, self.side = task, lo, lo + task.num_envs, stream, side  # side: the critic's stream
        self.sl = slice(self.lo, self.hi)


class PolicyInference:
    """The network side of one rollout step, shared by the training agent and the player: needs self.model (ImitatorNetwork),
    self.obs_enc (ImitationObs), self.value_mean_std, self.normalize_value, self._lib."""

    def _features(self, task, obs, t):
        """734-d in-network observation of step t (preprocess_input in eval mode: context frame context_padding + t, running statistics
        applied inside the kernel)"""
        return self.obs_enc.rollout(obs, task.context_feat, t)

    def _sync_obs_norm(self):
        rn = self.model.running_obs
        if rn._seen is None:
            rn._seen = int(rn.n) > 0
        if rn._seen:  # (a fresh model does not normalise, running_norm.py:36)
            self.obs_enc.set_running_stats(rn.mean, rn.std)
        else:
            self.obs_enc.set_running_stats(None, None)

    def _value(self, feat):
        value = self.model.critic(feat)
        if self.normalize_value:
            value = self.value_mean_std(value, True)
        return value

    def _policy_head(self, task, mu, noise, t, rows=None):
        """models/im_network_builder.py:226-228 + models/im_models.py:42-46 in one kernel; mu is updated in place.  rows = (actions, mus,
        sigmas, neglogpacs) rows of the experience buffer: written by the kernel itself (the returned sigma / neglogp ARE those rows)."""
        n = mu.shape[0]
        action = torch.empty_like(mu)
        ctx = task.context_feat
        frame = (task.context_padding + int(t)) if self.model.residual_action else -1
        if frame < 0:
            raise NotImplementedError("residual_action = False is not built (the reference's default and both configs use True)")
        if rows is not None:
            a_row, m_row, s_row, l_row = rows
            _lib.check(self._lib.v2p_policy_head_record(n, _lib.ptr(mu), _lib.ptr(ctx), ctx.shape[1], frame, _lib.ptr(self.model.sigma), _lib.ptr(noise), _lib.ptr(action),
                                                        _lib.ptr(s_row), _lib.ptr(l_row), _lib.ptr(a_row), _lib.ptr(m_row), _lib.current_stream(mu.device)), "v2p_policy_head_record")
            return action, s_row, l_row
        sigma = torch.empty_like(mu)
        nlp = torch.empty(n, dtype=torch.float32, device=mu.device)
        _lib.check(self._lib.v2p_policy_head(n, _lib.ptr(mu), _lib.ptr(ctx), ctx.shape[1], frame, _lib.ptr(self.model.sigma), _lib.ptr(noise),
                                             _lib.ptr(action), _lib.ptr(sigma), _lib.ptr(nlp), _lib.current_stream(mu.device)), "v2p_policy_head")
        return action, sigma, nlp


class PPOAgent(PolicyInference):
    def __init__(self, task, horizon_length=32, gamma=0.99, tau=0.95, learning_rate=2e-5, e_clip=0.2, critic_coef=5.0, mini_epochs=6,
                 minibatch_envs=512, grad_norm=50.0, units=(1024, 1024, 512), sigma_init=-1.756, seed=0, group=None,
                 normalize_value=True, normalize_advantage=True, entropy_coef=0.0, residual_action=True, reuse_next_values=True,
                 truncate_grads=True, overlap_critic=True, mixed_precision=False):
        tasks = list(task) if isinstance(task, (list, tuple)) else [task]
        self.tasks, self.task = tasks, tasks[0]
        self.horizon_length, self.gamma, self.tau = horizon_length, gamma, tau
        self.e_clip, self.critic_coef, self.entropy_coef, self.mini_epochs = e_clip, critic_coef, entropy_coef, mini_epochs
        self.grad_norm, self.truncate_grads = grad_norm, truncate_grads
        self.normalize_value, self.normalize_advantage = normalize_value, normalize_advantage
        self.reuse_next_values, self.overlap_critic = reuse_next_values, overlap_critic
        self.fused_record = True  # (tests switch it off to compare the launch with the torch statement of the same bookkeeping)
        # cfg `mixed_precision` (amass_im.yaml: False): the reference wraps the forward pass and the losses of calc_gradients in
        # torch.cuda.amp.autocast and scales the loss (im_agent.py:509, 548-563); the rollout stays float32 there as well
        self.mixed_precision = bool(mixed_precision)
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.mixed_precision) if self.mixed_precision else None
        self.group = group
        self.device = torch.device(self.task.device)
        if any(str(t.device) != str(self.task.device) or t.context_padding != self.task.context_padding for t in tasks):
            raise ValueError("the rollout groups of one agent live on one GPU and share the context layout")
        self.num_actors = sum(t.num_envs for t in tasks)
        self.num_actions = self.task.num_actions
        self.minibatch_envs = min(minibatch_envs, self.num_actors)
        lo, self.groups = 0, []
        for k, t in enumerate(tasks):
            self.groups.append(_Group(t, lo, None if len(tasks) == 1 else torch.cuda.Stream(device=self.device),
                                      torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None))
            lo += t.num_envs
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)  # the same initial weights on every rank
            self.model = ImitatorNetwork(self.num_actions, units, sigma_init, residual_action, self.device)
        self.model.eval()
        self.last_lr = float(learning_rate)
        self.optimizer = torch.optim.Adam(self.model.parameters(), self.last_lr, eps=1e-08, weight_decay=0.0)
        self.value_mean_std = ValueMeanStd(self.device)
        self.obs_enc = ImitationObs(self.task.context_padding)
        self._lib = _lib.load()
        self.experience_buffer = ExperienceBuffer(horizon_length, self.num_actors, self.task.num_obs, self.num_actions, self.device)
        rank = vdist.dist.get_rank(group) if _world(group) > 1 else 0
        self.action_gen = torch.Generator(device=self.device)
        self.action_gen.manual_seed(seed + 1000 * rank)
        self.idx_gen = torch.Generator(device="cpu")
        self.idx_gen.manual_seed(seed)  # the same minibatch order on every rank (AMPDataset._idx_buf)
        self._idx_buf = torch.randperm(self.num_actors, generator=self.idx_gen).to(self.device)
        self.dones = torch.zeros(self.num_actors, device=self.device)
        self.current_rewards = torch.zeros(self.num_actors, device=self.device)
        self.current_lengths = torch.zeros(self.num_actors, device=self.device)
        self.epoch_num = 0
        self.frame = 0
        self.dataset = None
        self.noise_fn = None  # tests: noise_fn(n) -> [N, num_actions] standard-normal draws of step n
        self.broadcast_state()  # world > 1: every rank starts from rank 0's weights, whatever its seed or RNG history

    def broadcast_state(self, src=0):
        """Rank `src`'s model, normalisers, optimizer state and counters to every rank of the group (RCCL broadcast, one flat buffer per
        dtype).  The reference's Horovod path does this in `hvd.setup_algo` (im_agent.py:174-175: broadcast_parameters of the model's
        state_dict + broadcast_optimizer_state): replica identity must not rest on equal seeds or on every rank reading the same file."""
        if _world(self.group) <= 1:
            return
        tensors = [p.data for p in self.model.parameters()] + list(self.model.buffers()) + list(self.model.running_obs.state_dict().values())
        v = self.value_mean_std
        tensors += [v.running_mean, v.running_var, v.count]
        for st in self.optimizer.state.values():  # (empty before the first step; after restore(): exp_avg, exp_avg_sq, step)
            tensors += [t for t in st.values() if torch.is_tensor(t)]
        counters = torch.tensor([self.epoch_num, self.frame], dtype=torch.float64, device=self.device)
        tensors.append(counters)
        # one flat buffer per DTYPE, staged on this rank's device whatever device a tensor lives on: Adam's `step` counters are CPU tensors
        # on a rank that materialised them and device tensors on a rank that loaded them with map_location (advisor r5) - grouped by
        # (dtype, device) the ranks would issue different numbers of collectives of different sizes
        seen, by_dtype = set(), {}
        for t in tensors:
            key = (str(t.device), t.data_ptr())
            if key in seen and t.numel() > 0:
                continue
            seen.add(key)
            by_dtype.setdefault(str(t.dtype), []).append(t)
        names = sorted(by_dtype)
        # the layout every rank is about to broadcast, compared BEFORE any data moves: a mismatch raises on every rank instead of hanging
        # in (or silently corrupting) mismatched collectives
        K = 8
        assert len(names) <= K
        man = torch.full((K, 3), -1, dtype=torch.int64, device=self.device)
        for i, nm in enumerate(names):
            man[i, 0] = sum(ord(c) * (k + 1) for k, c in enumerate(nm)) % (1 << 31)
            man[i, 1] = len(by_dtype[nm])
            man[i, 2] = sum(t.numel() for t in by_dtype[nm])
        allman = [torch.empty_like(man) for _ in range(_world(self.group))]
        vdist.dist.all_gather(allman, man, group=self.group)
        if any(not torch.equal(m, man) for m in allman):
            raise RuntimeError("broadcast_state: the ranks hold different state layouts (dtype hash, tensors, elements per dtype): %s"
                               % [m.cpu().tolist()[:len(names) + 1] for m in allman])
        for nm in names:
            ts = by_dtype[nm]
            flat = torch.cat([t.reshape(-1).to(self.device) for t in ts])
            vdist.dist.broadcast(flat, src=vdist.dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape).to(t.device))
                off += n
        self.epoch_num, self.frame = int(counters[0].item()), int(counters[1].item())
        self.model.running_obs._seen = None  # (host-side cache of n > 0: ask the device again)

    # compatibility with the round-2 attribute names
    @property
    def actor(self):
        return nn.Sequential(self.model.actor_mlp, self.model.mu)

    @property
    def critic(self):
        return nn.Sequential(self.model.critic_mlp, self.model.value)

    def set_eval(self):
        self.model.eval()
        self.model.running_obs.eval()
        self.value_mean_std.eval()

    def set_train(self):
        self.model.train()
        self.model.running_obs.train()
        self.value_mean_std.train()

    @staticmethod
    def _fused_record_ok(task):
        """v2p_rollout_record reads the task's buffers through raw pointers with the layouts of HumanoidSMPLIM hard-coded (int64 reset /
        terminate flags, float32 [N] rewards, float32 [N,4] sub-rewards, contiguous float32 observations): any other CUDA task - bool or
        int32 flags, another number of sub-rewards - takes the torch statement of the bookkeeping instead of being misread (advisor r5)."""
        ex = getattr(task, "extras", None)
        if not isinstance(ex, dict):
            return False
        n = task.num_envs
        # (the task fills `extras` in its first step: before it, the buffers the entries will alias)
        term, sub = ex.get("terminate", getattr(task, "_terminate_buf", None)), ex.get("sub_rewards", getattr(task, "_sub_rewards", None))
        checks = ((task.obs_buf, torch.float32, (n, task.num_obs)), (task.rew_buf, torch.float32, (n,)), (task.reset_buf, torch.int64, (n,)),
                  (term, torch.int64, (n,)), (sub, torch.float32, (n, 4)))
        return all(torch.is_tensor(x) and x.is_cuda and x.dtype == dt and tuple(x.shape) == shp and x.is_contiguous() for x, dt, shp in checks)

    def get_action_values(self, obs, t, task=None, feat=None, noise=None, value=None, rows=None):
        """im_agent.py:271-294 (`obs['t']` = t).  feat / value: what `_eval_critic` of the step before has already computed for this
        very observation (reuse_next_values)."""
        task = task or self.task
        if feat is None:
            feat = self._features(task, obs, t)
        mu = self.model.actor(feat)
        if noise is None:
            noise = torch.randn(mu.shape, device=mu.device, generator=self.action_gen)
        action, sigma, nlp = self._policy_head(task, mu, noise.contiguous(), t, rows)
        if value is None:
            value = self._value(feat)
        return {"actions": action, "mus": mu, "sigmas": sigma, "neglogpacs": nlp, "values": value}  # (value False: the caller has it)

    def _eval_critic(self, obs, t, task=None):
        """im_agent.py:296-303; also returns the features it was computed from"""
        feat = self._features(task or self.task, obs, t)
        return self._value(feat), feat

    # ------------------------------------------------------------------ rollout (im_agent.py:305-409), no host syncs in the loop
    @torch.no_grad()
    def play_steps(self, reset_fn=None):
        """reset_fn(task, group_index): tests reset the envs at given clip times instead of `task.reset()`."""
        buf = self.experience_buffer
        self.set_eval()
        self._sync_obs_norm()
        T, A = self.horizon_length, self.num_actions
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        multi = len(self.groups) > 1
        self.dones.zero_()
        self.current_rewards.zero_()
        self.current_lengths.zero_()
        # the epoch's action noise in one draw, so that the env -> noise assignment does not depend on how the envs are grouped
        noise_all = None if self.noise_fn else torch.randn((T, self.num_actors, A), device=self.device, generator=self.action_gen)
        st = []
        for gi, g in enumerate(self.groups):
            # (built BEFORE the start event is recorded: the zero fills run on the main stream, the group streams wait for the event)
            s = dict(prev_dones=torch.zeros(g.task.num_envs, device=self.device),
                     # device accumulators of what the reference collects through .nonzero() / AverageMeter on the host:
                     # finished episodes: n, sum reward, sum length | steps of envs not done before: n, sum reward
                     acc=torch.zeros(8, dtype=torch.float64, device=self.device), sub=None, feat=None, value=None)
            st.append(s)
        start = torch.cuda.Event() if multi else None
        if multi:
            start.record(main)
        cuda = self.device.type == "cuda"
        overlap = self.overlap_critic and self.reuse_next_values and cuda
        # the after-step bookkeeping as one HIP launch (device tensors, the default rollout); the torch statement of it below stays for CPU
        # tensors (the gloo tests' stub tasks) and for the reference's two-critic-passes variant, and is what the kernel is tested against
        fused = cuda and self.reuse_next_values and self.fused_record and all(self._fused_record_ok(g.task) for g in self.groups)
        if fused:
            for s in st:
                s["sub"] = torch.zeros(4, dtype=torch.float64, device=self.device)

        def on(g):
            return torch.cuda.stream(g.stream) if multi else _NullCtx()

        def critic_of(g, s, feat, n, terminated):
            """value of the observation BEFORE step n (features `feat`): `values` of step n and, masked by the terminations of the step
            that produced the observation, `next_values` of step n - 1.  With overlap_critic on the group's SIDE stream: the critic pass is
            needed by the buffer only, never by the next physics launch, so it runs while the actor pass and the physics of step n do."""
            def work():
                if fused:
                    # critic output -> un-normalised -> `values` of step n and the masked `next_values` of step n - 1: one launch
                    raw = self.model.critic(feat)
                    vms, td = self.value_mean_std, buf.tensor_dict
                    _lib.check(self._lib.v2p_value_record(
                        raw.shape[0], _lib.ptr(raw), _lib.ptr(vms.running_mean) if self.normalize_value else None,
                        _lib.ptr(vms.running_var) if self.normalize_value else None, float(vms.epsilon), _lib.ptr(terminated) if n > 0 else None,
                        _lib.ptr(td["values"][n, g.sl]) if n < T else None, _lib.ptr(td["next_values"][n - 1, g.sl]) if n > 0 else None,
                        _lib.current_stream(self.device)), "v2p_value_record")
                    return
                v = self._value(feat)
                if n < T:
                    buf.update_data("values", n, v, g.sl)
                if n > 0:
                    buf.update_data("next_values", n - 1, v * (1.0 - terminated), g.sl)  # end_value_type 'next'
            if not overlap:
                return work()
            cur = torch.cuda.current_stream(self.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            g.side.wait_event(ev)
            for x in (feat, terminated):
                if x is not None:
                    x.record_stream(g.side)
            with torch.cuda.stream(g.side):
                work()

        for gi, g in enumerate(self.groups):
            with on(g):
                if multi:
                    g.stream.wait_event(start)
                if overlap:
                    g.side.wait_stream(torch.cuda.current_stream(self.device))
                (reset_fn(g.task, gi) if reset_fn else g.task.reset())  # per-epoch reset of all envs + context window
        for n in range(T):
            for gi, g in enumerate(self.groups):
                s, task, sl = st[gi], g.task, g.sl
                with on(g):
                    obs = task.obs_buf
                    if not (fused and n > 0):  # (fused: obses[n] IS next_obses[n - 1], one tensor of T + 1 rows - the record of step n - 1 wrote it)
                        buf.update_data("obses", n, obs, sl)
                    noise = self.noise_fn(n)[sl] if self.noise_fn else noise_all[n, sl]
                    if self.reuse_next_values:
                        if n == 0:
                            s["feat"], s["term"] = self._features(task, obs, 0), None
                        td = buf.tensor_dict
                        rows = (td["actions"][n, sl], td["mus"][n, sl], td["sigmas"][n, sl], td["neglogpacs"][n, sl]) if fused else None
                        res = self.get_action_values(obs, n, task, s["feat"], noise, value=False, rows=rows)
                    else:
                        res = self.get_action_values(obs, n, task, None, noise)
                        buf.update_data("values", n, res["values"], sl)
                    if not fused:
                        for k in ("actions", "mus", "sigmas", "neglogpacs"):
                            buf.update_data(k, n, res[k], sl)
                    if self.reuse_next_values:
                        # issued here, right in front of the physics launch it is to run beside (issued any earlier it would share the GPU
                        # with the actor pass instead: two GEMM streams gain nothing from each other)
                        critic_of(g, s, s["feat"], n, s["term"])
                    task.step(res["actions"])  # masks the rows of finished envs in place, like the reference
                    obs, rewards = task.obs_buf, task.rew_buf
                    if fused:
                        # rewards / next_obses / dones rows, dones and terminate as floats, episode returns / lengths and the statistics
                        # below: ONE launch (v2p_rollout_record) instead of ~25 elementwise / reduction kernels per step
                        terminated = torch.empty((task.num_envs, 1), device=self.device)
                        td = buf.tensor_dict
                        _lib.check(self._lib.v2p_rollout_record(
                            task.num_envs, _lib.ptr(obs), obs.shape[1], _lib.ptr(rewards), _lib.ptr(task.reset_buf), _lib.ptr(task.extras["terminate"]),
                            _lib.ptr(task.extras["sub_rewards"]), _lib.ptr(td["next_obses"][n, sl]), _lib.ptr(td["rewards"][n, sl]), _lib.ptr(td["dones"][n, sl]),
                            _lib.ptr(self.dones[sl]), _lib.ptr(terminated), _lib.ptr(s["prev_dones"]), _lib.ptr(self.current_rewards[sl]),
                            _lib.ptr(self.current_lengths[sl]), _lib.ptr(s["acc"]), _lib.ptr(s["sub"]), _lib.current_stream(self.device)), "v2p_rollout_record")
                        s["feat"], s["term"] = self._features(task, obs, n + 1), terminated
                        if n == T - 1:
                            critic_of(g, s, s["feat"], T, terminated)
                        continue
                    dones = task.reset_buf.float()
                    self.dones[sl] = dones
                    buf.update_data("rewards", n, rewards, sl)  # rewards_shaper scale_value 1
                    buf.update_data("next_obses", n, obs, sl)
                    buf.update_data("dones", n, dones, sl)
                    terminated = task.extras["terminate"].float().unsqueeze(-1)
                    if self.reuse_next_values:
                        s["feat"], s["term"] = self._features(task, obs, n + 1), terminated
                        if n == T - 1:
                            critic_of(g, s, s["feat"], T, terminated)
                    else:
                        value_next, _ = self._eval_critic(obs, n + 1, task)
                        buf.update_data("next_values", n, value_next * (1.0 - terminated), sl)
                    self.current_rewards[sl] += rewards
                    self.current_lengths[sl] += 1
                    step_dones = dones * (1.0 - s["prev_dones"])  # envs that finished at this step
                    alive_before = 1.0 - s["prev_dones"]
                    acc = s["acc"]
                    acc[0] += step_dones.sum()
                    acc[1] += (self.current_rewards[sl] * step_dones).sum()
                    acc[2] += (self.current_lengths[sl] * step_dones).sum()
                    acc[3] += alive_before.sum()
                    acc[4] += (rewards * alive_before).sum()
                    sub = (task.extras["sub_rewards"] * alive_before.unsqueeze(-1)).sum(0).double()
                    s["sub"] = sub if s["sub"] is None else s["sub"] + sub
                    s["prev_dones"] = dones
            # (the reference leaves the loop when every env is done - a host sync per step; see the module docstring)
        for gi, g in enumerate(self.groups):
            s, sl = st[gi], g.sl
            with on(g):
                still = 1.0 - self.dones[sl]
                s["acc"][0] += still.sum()
                s["acc"][1] += (self.current_rewards[sl] * still).sum()
                s["acc"][2] += (self.current_lengths[sl] * still).sum()
                if overlap:
                    torch.cuda.current_stream(self.device).wait_stream(g.side)
        if multi:
            for g in self.groups:
                main.wait_stream(g.stream)
        acc = torch.stack([s["acc"] for s in st]).sum(0)
        sub_acc = torch.stack([s["sub"] for s in st]).sum(0)
        td = buf.tensor_dict
        mb_advs = discount_values(td["dones"], td["values"], td["rewards"], td["next_values"], self.gamma, self.tau)
        mb_returns = mb_advs + td["values"]
        batch = {k: v.transpose(0, 1) for k, v in td.items()}  # swap01: [N, T, ...] views, no copy
        batch["returns"] = mb_returns.transpose(0, 1)
        batch["alive"] = 1.0 - batch["dones"]  # the `dones` of step n as overwritten AFTER the step (im_agent.py:343, 403)
        batch["played_frames"] = self.num_actors * self.horizon_length
        if multi:
            batch["context_feat"] = torch.cat([t.context_feat for t in self.tasks], dim=0)
            batch["context_mask"] = torch.cat([t.context_mask for t in self.tasks], dim=0)
        else:
            batch["context_feat"], batch["context_mask"] = self.task.context_feat, self.task.context_mask
        batch["stats"] = (acc, sub_acc)
        return batch

    # ------------------------------------------------------------------ update
    def _calc_advs(self, batch):
        """im_agent.py:461-473; the statistics of the alive entries are global over the ranks"""
        adv = (batch["returns"] - batch["values"]).sum(-1)
        if self.normalize_advantage:
            adv = vdist.normalize_advantages(adv, batch["alive"], self.group)
        return adv

    @torch.no_grad()
    def prepare_dataset(self, batch, feat_raw=None):
        """im_agent.py:411-459: advantages from the un-normalised values; the value normaliser (training mode) takes the values in and
        normalises them, then the returns.  Added: what the network's training-mode forward needs of the context, computed once per
        epoch instead of once per minibatch - the raw 734-d features of every (env, step) (HIP kernel; `feat_raw` [N,T,734] given: the
        caller's, e.g. a learner process that receives them) and the residual-action term."""
        values, returns = batch["values"], batch["returns"]
        advantages = self._calc_advs(batch)
        if self.normalize_value:
            values = self.value_mean_std(values.reshape(-1, 1), group=self.group).view(values.shape)
            returns = self.value_mean_std(returns.reshape(-1, 1), group=self.group).view(returns.shape)
        n, t = batch["alive"].shape
        ctx, pad = batch["context_feat"], self.task.context_padding
        if feat_raw is None:
            feat_raw = ImitationObs(pad).training(batch["obses"].contiguous(), ctx).view(n, t, -1)  # preprocess_input, flatten=True
        tgt = torch.zeros((n, t, self.num_actions), device=self.device)
        tgt[:, :, :NUM_DOF] = ctx[:, pad:pad + t, CTX_DOF_POS:CTX_DOF_POS + NUM_DOF]
        self.dataset = {"old_values": values, "old_logp_actions": batch["neglogpacs"], "advantages": advantages, "returns": returns,
                        "actions": batch["actions"], "mu": batch["mus"], "sigma": batch["sigmas"], "alive": batch["alive"],
                        "feat_raw": feat_raw, "target_dof_pad": tgt}
        return self.dataset

    def _get_item(self, i):
        """AMPDataset._get_item (learning/amp_datasets.py:17-31): minibatch i = envs _idx_buf[i m : (i + 1) m]; reshuffled at the end"""
        start, end = i * self.minibatch_envs, (i + 1) * self.minibatch_envs
        idx = self._idx_buf[start:end]
        item = {k: v[idx] for k, v in self.dataset.items()}
        if end >= self.num_actors:
            self._idx_buf = torch.randperm(self.num_actors, generator=self.idx_gen).to(self.device)
        return item

    def calc_gradients(self, input_dict):
        """im_agent.py:475-587 on one minibatch of envs ([m, T, ...] tensors, flattened like :478-481)."""
        self.set_train()
        d = {k: v.reshape((-1, v.shape[-1]) if v.dim() > 2 else (-1,)) for k, v in input_dict.items()}
        alive, advantage = d["alive"], d["advantages"]
        with torch.autocast(device_type=self.device.type, dtype=torch.float16, enabled=self.mixed_precision):
            res = self.model.forward_train(d["feat_raw"], d["target_dof_pad"], d["actions"], self.group)
            # _actor_loss / _critic_loss (common_agent.py:491-520; clip_value False), bound_loss with bounds_loss_coef None = 0
            ratio = torch.exp(d["old_logp_actions"] - res["prev_neglogp"])
            a_loss = torch.max(-advantage * ratio, -advantage * torch.clamp(ratio, 1.0 - self.e_clip, 1.0 + self.e_clip))
            c_loss = (d["returns"] - res["values"]) ** 2
            mask = alive.unsqueeze(1)
            a_l, c_l, ent = masked_mean(a_loss.unsqueeze(1), mask), masked_mean(c_loss, mask), masked_mean(res["entropy"].unsqueeze(1), mask)
            loss = a_l + self.critic_coef * c_l - self.entropy_coef * ent
        params = [p for p in self.model.parameters() if p.requires_grad]
        for p in params:
            p.grad = None
        (self.scaler.scale(loss) if self.mixed_precision else loss).backward()
        world = _world(self.group)
        if world > 1:  # data parallel over env shards: one flat all-reduce of the gradients over RCCL, averaged (Horovod's DistributedOptimizer)
            # (the SCALED gradients are reduced, unscale_ comes after - the order of the reference's Horovod path, optimizer.synchronize()
            # before scaler.unscale_(): every rank then sees the same inf / NaN verdict and takes the same step-or-skip decision)
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            vdist.dist.all_reduce(flat, group=self.group)
            flat /= world
            o = 0
            for p in params:
                p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        if self.mixed_precision:
            self.scaler.unscale_(self.optimizer)
        if self.truncate_grads:
            nn.utils.clip_grad_norm_(params, self.grad_norm)
        if self.mixed_precision:
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            self.optimizer.step()
        with torch.no_grad():
            kl = (policy_kl(res["mus"].detach(), res["sigmas"].detach(), d["mu"], d["sigma"]) * alive).sum() / alive.numel()
            clip_frac = (torch.abs(ratio.detach() - 1.0) > self.e_clip).float().mean()
        self.train_result = {"actor_loss": a_l.detach(), "critic_loss": c_l.detach(), "entropy": ent.detach(), "kl": kl, "actor_clip_frac": clip_frac}
        return self.train_result

    def train_epoch(self):
        """learning/common_agent.py:146-216"""
        sync = torch.cuda.synchronize if self.device.type == "cuda" else (lambda: None)
        sync()
        t0 = time.perf_counter()
        batch = self.play_steps()
        sync()
        t1 = time.perf_counter()
        self.set_train()
        frames = batch.pop("played_frames")
        self.prepare_dataset(batch)
        info = {"actor_loss": [], "critic_loss": [], "kl": [], "actor_clip_frac": []}
        for _ in range(self.mini_epochs):
            for i in range(self.num_actors // self.minibatch_envs):
                r = self.calc_gradients(self._get_item(i))
                for k in info:
                    info[k].append(r[k])
        sync()
        t2 = time.perf_counter()
        acc, sub_acc = batch["stats"]
        alive = batch["alive"]
        tail = torch.stack([torch.stack(info[k]).mean().double() for k in ("actor_loss", "critic_loss", "kl", "actor_clip_frac")] + [alive.double().mean()])
        host = torch.cat([acc, sub_acc, tail]).tolist()  # the ONE read-back of the epoch's statistics
        acc_h, sub_h, tail_h = host[:8], host[8:8 + sub_acc.numel()], host[8 + sub_acc.numel():]
        for t in self.tasks:
            if hasattr(t, "check"):
                t.check()  # a substep job that timed out surfaces here, once per epoch (the streams are idle: no extra wait)
        self.epoch_num += 1
        self.frame += frames
        play_time, update_time = t1 - t0, t2 - t1
        return {"play_time": play_time, "update_time": update_time, "total_time": t2 - t0, "frames": frames,
                "fps_step": frames / play_time, "fps_total": frames / (t2 - t0),
                "mean_rewards": acc_h[1] / max(acc_h[0], 1.0), "mean_lengths": acc_h[2] / max(acc_h[0], 1.0),
                "step_rewards": acc_h[4] / max(acc_h[3], 1.0), "step_sub_rewards": [s / max(acc_h[3], 1.0) for s in sub_h],
                "alive_ratio": tail_h[4], "a_loss": tail_h[0], "c_loss": tail_h[1], "kl": tail_h[2], "clip_frac": tail_h[3]}

    # ------------------------------------------------------------------ the reference agent's configuration and checkpoint surface
    @classmethod
    def from_config(cls, task, params, group=None, **overrides):
        """`params`: the `params` block of the reference's yaml (cfg/amass_im.yaml:55-143: `config` = the PPO hyper-parameters rl_games'
        A2CBase reads, `network` = what ImitatorBuilder reads).  Options this agent does not build are refused, not ignored."""
        cfg, net = params.get("config", {}), params.get("network", {})
        space = net.get("space", {}).get("continuous", {})
        if not space.get("fixed_sigma", True) or space.get("learn_sigma", False):
            raise NotImplementedError("a learned sigma is not built (cfg/amass_im.yaml: fixed_sigma True, learn_sigma False)")
        for key, want in (("use_ik", False), ("kinematic_pretrained", False)):
            if net.get(key, want) != want:
                raise NotImplementedError("network.%s = %r is not built" % (key, net.get(key)))
        if not net.get("use_running_obs", False) or net.get("running_obs_type", "rl_game") != "ours":
            raise NotImplementedError("the in-network observation normaliser is RunningNorm ('use_running_obs: True, running_obs_type: ours', cfg/amass_im.yaml:67-68)")
        if cfg.get("normalize_input", False) or cfg.get("clip_value", False) or cfg.get("bounds_loss_coef") is not None or cfg.get("lr_schedule", "constant") != "constant":
            raise NotImplementedError("normalize_input / clip_value / bounds_loss_coef / a learning-rate schedule are not built (off in both reference configs)")
        kw = dict(horizon_length=cfg.get("horizon_length", 32), gamma=cfg.get("gamma", 0.99), tau=cfg.get("tau", 0.95),
                  learning_rate=float(cfg.get("learning_rate", 2e-5)), e_clip=cfg.get("e_clip", 0.2), critic_coef=cfg.get("critic_coef", 5.0),
                  mini_epochs=cfg.get("mini_epochs", 6), minibatch_envs=cfg.get("minibatch_size", 512), grad_norm=cfg.get("grad_norm", 50.0),
                  truncate_grads=cfg.get("truncate_grads", True), entropy_coef=cfg.get("entropy_coef", 0.0),
                  normalize_value=cfg.get("normalize_value", True), normalize_advantage=cfg.get("normalize_advantage", True),
                  mixed_precision=cfg.get("mixed_precision", False), units=tuple(net.get("mlp", {}).get("units", (1024, 1024, 512))),
                  sigma_init=float(space.get("sigma_init", {}).get("val", -1.756)), residual_action=net.get("residual_action", True),
                  seed=params.get("seed", 0), group=group)
        kw.update(overrides)
        agent = cls(task, **kw)
        agent.max_epochs, agent.save_freq = int(cfg.get("max_epochs", 10000)), int(cfg.get("save_frequency", 0))
        agent.config_name = cfg.get("name", "Humanoid")
        return agent

    def get_full_state_weights(self):
        """The checkpoint dict of the reference's agent (rl_games A2CBase.get_full_state_weights [1.1.4, from memory] through
        CommonAgent / ImitatorAgent.restore, im_agent.py:109-112): `model` with the `a2c_network.` names, RunningNorm buffers included,
        `reward_mean_std` = the value normaliser, optimizer, epoch, frame."""
        model = {"a2c_network." + k: v.detach().clone() for k, v in self.model.state_dict().items()}
        model.update({"a2c_network.running_obs." + k: v.detach().clone() for k, v in self.model.running_obs.state_dict().items()})
        v = self.value_mean_std
        return {"model": model, "reward_mean_std": {"running_mean": v.running_mean.clone(), "running_var": v.running_var.clone(), "count": v.count.clone()},
                "optimizer": self.optimizer.state_dict(), "epoch": self.epoch_num, "frame": self.frame, "last_mean_rewards": getattr(self, "last_mean_rewards", -100500)}

    def set_full_state_weights(self, weights, optimizer=True):
        self.model.load_reference_state_dict(weights["model"])
        r = weights.get("reward_mean_std")
        if r is not None:
            v = self.value_mean_std
            v.running_mean = torch.as_tensor(r["running_mean"], dtype=torch.float64, device=self.device).reshape(1).clone()
            v.running_var = torch.as_tensor(r["running_var"], dtype=torch.float64, device=self.device).reshape(1).clone()
            v.count = torch.as_tensor(r["count"], dtype=torch.float64, device=self.device).reshape(()).clone()
        if optimizer and "optimizer" in weights:
            self.optimizer.load_state_dict(weights["optimizer"])
        self.epoch_num, self.frame = int(weights.get("epoch", 0)), int(weights.get("frame", 0))
        self.last_mean_rewards = weights.get("last_mean_rewards", -100500)

    def save(self, fn):
        """rl_games torch_ext.save_checkpoint: `<fn>.pth`"""
        torch.save(self.get_full_state_weights(), fn + ".pth")
        return fn + ".pth"

    def restore(self, path):
        """world > 1: what rank 0 loaded is what every rank continues from (the file need not be visible to, or the same on, every rank)"""
        rank0 = _world(self.group) <= 1 or vdist.dist.get_rank(self.group) == 0
        if rank0 or os.path.exists(path):
            self.set_full_state_weights(torch.load(path, map_location=self.device, weights_only=False))
        self._materialise_optimizer_state()
        self.broadcast_state()

    def _materialise_optimizer_state(self):
        """Adam's per-parameter state exists only after a step or a load: a rank that could not read the checkpoint gets zero-filled slots
        of the right shapes so that the broadcast has something to fill."""
        if _world(self.group) <= 1:
            return
        flag = torch.tensor([1.0 if len(self.optimizer.state) else 0.0], device=self.device)
        vdist.dist.all_reduce(flag, op=vdist.dist.ReduceOp.MAX, group=self.group)
        if flag.item() > 0 and not len(self.optimizer.state):
            for g in self.optimizer.param_groups:
                for p in g["params"]:
                    self.optimizer.state[p] = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}

    def load_pretrained(self, path):
        """ImitatorAgent.load_pretrained (im_agent.py:114-155) for EmbodyPose checkpoints: the weights and normalisers, not the optimizer"""
        self.set_full_state_weights(torch.load(path, map_location=self.device, weights_only=False), optimizer=False)
        self.broadcast_state()

    def train(self, max_epochs=None, log=print, network_path=None):
        """ImitatorAgent.train (im_agent.py:164-269): epochs of train_epoch with the reference's log line; checkpoints `<name>_latest` /
        `<name>_epoch%05d` every save_frequency epochs and at the end."""
        import os

        last = getattr(self, "max_epochs", 10000) if max_epochs is None else self.epoch_num + int(max_epochs)
        name = getattr(self, "config_name", "Humanoid")
        r = None
        while self.epoch_num < last:
            for t in self.tasks:
                if hasattr(t, "pre_epoch"):
                    t.pre_epoch(self.epoch_num)  # (im_agent.py:180-181)
            r = self.train_epoch()
            self.last_mean_rewards = r["mean_rewards"]
            if log:
                log(self.format_epoch_line(r))
            if network_path and getattr(self, "save_freq", 0) > 0 and self.epoch_num % self.save_freq == 0:
                self.save(os.path.join(network_path, "%s_latest" % name))
                self.save(os.path.join(network_path, "%s_epoch%05d" % (name, self.epoch_num)))
        if network_path:
            self.save(os.path.join(network_path, "%s_latest" % name))
        return r

    def format_epoch_line(self, r):
        """the reference's per-epoch line (im_agent.py:211-214)"""
        return ("%d\tT_play %.2f\tT_update %.2f\tstep_rewards %.4f %s\teps_len %.2f\talive %.2f\tfps step %d\tfps total %d"
                % (self.epoch_num, r["play_time"], r["update_time"], r["step_rewards"], "[" + ",".join("%.4f" % s for s in r["step_sub_rewards"]) + "]",
                   r["mean_lengths"], r["alive_ratio"], r["fps_step"], r["fps_total"]))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
