"""PPO rollout bookkeeping on the device: what `embodied_pose/agents/im_agent.py:305-409` (`play_steps`),
`:412-473` (`prepare_dataset`, `_calc_advs`) and `learning/common_agent.py:146-216` (`train_epoch`) do around the VecTask, with

  * the experience buffer resident on the GPU as [T, N, ...] tensors written in place (rl_games' ExperienceBuffer.update_data),
  * NO host synchronisation inside the 32-step rollout: the reference's `.nonzero()` bookkeeping of finished episodes
    (`im_agent.py:366-386`) and its `torch.all(self.dones == 1)` early exit (`:388`) become masked device reductions accumulated
    over the epoch and read once at its end,
  * the GAE reverse scan as the HIP kernel (`v2p_gae`), the advantage statistics as a 3-number all-reduce over RCCL
    (`dist.global_advantage_stats`; the reference normalises rank-locally), gradients averaged over the ranks,
  * the 734-d in-network observation + RunningNorm (eval) as the fused HIP kernel (`learning.ImitationObs`).

The policy / value networks are the dense MLPs of `cfg/amass_im.yaml:86-88` (units [1024, 1024, 512], relu; fixed sigma
exp(-1.756), `:76-81`) through torch (rocBLAS): they are NOT part of the accelerated path, and the reference's context encoder
(`im_network_builder.py`, pose_im_rnn) is not rebuilt - the actor and critic here read the 734-d observation only.  This is what
BASELINE config 5 ("full PPO train loop") needs to run and to print the reference's `fps step / fps total`
(`im_agent.py:204-214`); it is not a reimplementation of rl_games.
"""
import math
import time

import torch
import torch.nn as nn

from . import dist as vdist
from .learning import OBS_IMITATION_DIM, ImitationObs, discount_values


class MLP(nn.Module):
    def __init__(self, inp, units, out):
        super().__init__()
        layers, d = [], inp
        for u in units:
            layers += [nn.Linear(d, u), nn.ReLU()]
            d = u
        layers.append(nn.Linear(d, out))
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


class RunningMeanStd:
    """rl_games' value normaliser / the reference's RunningNorm update rule (models/running_norm.py:22-31) on flat tensors; statistics
    can be merged across ranks (sum, sum of squares, count) before the update."""

    def __init__(self, dim, device, clip=5.0):
        self.n = torch.zeros((), dtype=torch.float64, device=device)
        self.mean = torch.zeros(dim, device=device)
        self.var = torch.zeros(dim, device=device)
        self.std = torch.zeros(dim, device=device)
        self.clip = clip

    def update(self, x, mask=None, group=None):
        """x [M, dim]; rows with mask == 0 are ignored.  One all-reduce of (count, sum, sum of squares) when distributed."""
        x = x.double()
        w = torch.ones(x.shape[0], 1, dtype=torch.float64, device=x.device) if mask is None else mask.double().reshape(-1, 1)
        stats = torch.cat([w.sum().reshape(1), (x * w).sum(0), (x * x * w).sum(0)])
        if vdist.dist.is_initialized() and vdist.dist.get_world_size(group) > 1:
            vdist.dist.all_reduce(stats, group=group)
        d = x.shape[1]
        m = stats[0]
        mean_x = stats[1:1 + d] / torch.clamp(m, min=1.0)
        var_x = torch.clamp(stats[1 + d:] / torch.clamp(m, min=1.0) - mean_x * mean_x, min=0.0)
        wgt = self.n / torch.clamp(m + self.n, min=1.0)
        new_var = wgt * self.var.double() + (1 - wgt) * var_x + wgt * (1 - wgt) * (mean_x - self.mean.double()) ** 2
        new_mean = wgt * self.mean.double() + (1 - wgt) * mean_x
        self.var, self.mean = new_var.float(), new_mean.float()
        self.std = self.var.sqrt()
        self.n = self.n + m

    def normalize(self, x):
        return torch.clamp((x - self.mean) / (self.std + 1e-8), -self.clip, self.clip)

    def denormalize(self, y):
        return y * (self.std + 1e-8) + self.mean


class ExperienceBuffer:
    """[T, N, ...] device tensors, written in place step by step (rl_games ExperienceBuffer.update_data / tensor_dict)."""

    def __init__(self, horizon, num_envs, obs_dim, act_dim, device):
        f = dict(dtype=torch.float32, device=device)
        self.tensor_dict = {
            "obses": torch.zeros((horizon, num_envs, obs_dim), **f), "next_obses": torch.zeros((horizon, num_envs, obs_dim), **f),
            "dones": torch.zeros((horizon, num_envs), **f),
            "rewards": torch.zeros((horizon, num_envs, 1), **f), "values": torch.zeros((horizon, num_envs, 1), **f),
            "next_values": torch.zeros((horizon, num_envs, 1), **f), "actions": torch.zeros((horizon, num_envs, act_dim), **f),
            "neglogpacs": torch.zeros((horizon, num_envs), **f), "mus": torch.zeros((horizon, num_envs, act_dim), **f),
            "sigmas": torch.zeros((horizon, num_envs, act_dim), **f),
        }

    def update_data(self, name, n, value):
        self.tensor_dict[name][n].copy_(value.reshape(self.tensor_dict[name][n].shape))


def neglogp(x, mu, logstd):
    """rl_games ModelA2CContinuousLogStd.neglogp"""
    return 0.5 * (((x - mu) / torch.exp(logstd)) ** 2).sum(-1) + 0.5 * math.log(2.0 * math.pi) * x.shape[-1] + logstd.sum(-1)


class PPOAgent:
    def __init__(self, task, horizon_length=32, gamma=0.99, tau=0.95, learning_rate=2e-5, e_clip=0.2, critic_coef=5.0, mini_epochs=6,
                 minibatch_envs=512, grad_norm=50.0, units=(1024, 1024, 512), sigma_init=-1.756, seed=0, group=None,
                 normalize_value=True, normalize_advantage=True):
        self.task, self.horizon_length, self.gamma, self.tau = task, horizon_length, gamma, tau
        self.e_clip, self.critic_coef, self.mini_epochs, self.grad_norm = e_clip, critic_coef, mini_epochs, grad_norm
        self.normalize_value, self.normalize_advantage = normalize_value, normalize_advantage
        self.group = group
        self.device = torch.device(task.device)
        self.num_actors = task.num_envs
        self.minibatch_envs = min(minibatch_envs, self.num_actors)
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)  # the same initial weights on every rank
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            self.actor = MLP(OBS_IMITATION_DIM, units, task.num_actions).to(self.device)
            self.critic = MLP(OBS_IMITATION_DIM, units, 1).to(self.device)
        self.logstd = torch.full((task.num_actions,), float(sigma_init), device=self.device)  # fixed_sigma, learn_sigma False
        self.optimizer = torch.optim.Adam(list(self.actor.parameters()) + list(self.critic.parameters()), lr=learning_rate, eps=1e-8)
        self.obs_norm = RunningMeanStd(OBS_IMITATION_DIM, self.device)
        self.value_norm = RunningMeanStd(1, self.device)
        self.obs_enc = ImitationObs(task.context_padding)
        self.experience_buffer = ExperienceBuffer(horizon_length, self.num_actors, task.num_obs, task.num_actions, self.device)
        self.action_gen = torch.Generator(device=self.device)
        self.action_gen.manual_seed(seed + 1000 * (vdist.dist.get_rank(group) if vdist.dist.is_initialized() else 0))
        self.dones = torch.zeros(self.num_actors, device=self.device)
        self.current_rewards = torch.zeros(self.num_actors, device=self.device)
        self.current_lengths = torch.zeros(self.num_actors, device=self.device)
        self.epoch_num = 0
        self.frame = 0
        self._obs_norm_ready = self._value_norm_ready = False

    # ------------------------------------------------------------------ network side
    def _features(self, obs, t):
        """734-d in-network observation of step t (rollout flavour); the running statistics are applied inside the kernel."""
        return self.obs_enc.rollout(obs, self.task.context_feat, t)

    def _sync_obs_norm(self):
        if self._obs_norm_ready:  # (host-side flags: no device read in the rollout loop)
            self.obs_enc.set_running_stats(self.obs_norm.mean, self.obs_norm.std)

    def get_action_values(self, obs, t):
        feat = self._features(obs, t)
        mu = self.actor(feat)
        sigma = torch.exp(self.logstd).expand_as(mu)
        action = mu + sigma * torch.randn(mu.shape, device=mu.device, generator=self.action_gen)
        value = self.critic(feat)
        if self.normalize_value and self._value_norm_ready:
            value = self.value_norm.denormalize(value)
        return {"actions": action, "mus": mu, "sigmas": sigma, "neglogpacs": neglogp(action, mu, self.logstd), "values": value}

    def eval_critic(self, obs, t):
        value = self.critic(self._features(obs, t))
        if self.normalize_value and self._value_norm_ready:
            value = self.value_norm.denormalize(value)
        return value

    # ------------------------------------------------------------------ rollout (im_agent.py:305-409), no host syncs in the loop
    @torch.no_grad()
    def play_steps(self):
        task, buf = self.task, self.experience_buffer
        self._sync_obs_norm()
        task.reset()  # per-epoch reset of all envs + context window (env_reset)
        obs = task.obs_buf
        self.dones.zero_()
        prev_dones = torch.zeros_like(self.dones)
        self.current_rewards.zero_()
        self.current_lengths.zero_()
        # device accumulators of what the reference collects through .nonzero() / AverageMeter on the host
        acc = torch.zeros(8, dtype=torch.float64, device=self.device)  # finished: n, sum reward, sum length | step: n, sum reward | spare
        sub_acc = torch.zeros(4, dtype=torch.float64, device=self.device)
        for n in range(self.horizon_length):
            buf.update_data("obses", n, obs)
            res = self.get_action_values(obs, n)
            for k in ("actions", "mus", "sigmas", "neglogpacs", "values"):
                buf.update_data(k, n, res[k])
            actions = res["actions"].contiguous()
            task.step(actions)  # masks the rows of finished envs in place, like the reference
            obs, rewards = task.obs_buf, task.rew_buf
            self.dones = task.reset_buf.float()
            buf.update_data("rewards", n, rewards)  # rewards_shaper scale_value 1
            buf.update_data("next_obses", n, obs)
            buf.update_data("dones", n, self.dones)
            terminated = task.extras["terminate"].float().unsqueeze(-1)
            next_vals = self.eval_critic(obs, n + 1) * (1.0 - terminated)  # end_value_type 'next'
            buf.update_data("next_values", n, next_vals)
            self.current_rewards += rewards
            self.current_lengths += 1
            step_dones = self.dones * (1.0 - prev_dones)  # envs that finished at this step
            alive_before = 1.0 - prev_dones
            acc[0] += step_dones.sum()
            acc[1] += (self.current_rewards * step_dones).sum()
            acc[2] += (self.current_lengths * step_dones).sum()
            acc[3] += alive_before.sum()
            acc[4] += (rewards * alive_before).sum()
            sub_acc += (task.extras["sub_rewards"] * alive_before.unsqueeze(-1)).sum(0).double()
            prev_dones = self.dones.clone()
            # (the reference leaves the loop when every env is done - a host sync per step; finished envs are masked out of every
            # statistic and of the loss by `alive`, so running the remaining steps changes no result)
        still = 1.0 - self.dones
        acc[0] += still.sum()
        acc[1] += (self.current_rewards * still).sum()
        acc[2] += (self.current_lengths * still).sum()
        td = buf.tensor_dict
        mb_advs = discount_values(td["dones"], td["values"], td["rewards"], td["next_values"], self.gamma, self.tau)
        mb_returns = mb_advs + td["values"]
        batch = {k: v.transpose(0, 1) for k, v in td.items()}  # swap01: [N, T, ...] views, no copy
        batch["returns"] = mb_returns.transpose(0, 1)
        batch["alive"] = 1.0 - batch["dones"]  # the `dones` of step n as overwritten AFTER the step (im_agent.py:343, 403)
        batch["played_frames"] = self.num_actors * self.horizon_length
        batch["context_feat"], batch["context_mask"] = task.context_feat, task.context_mask
        batch["stats"] = (acc, sub_acc)
        return batch

    # ------------------------------------------------------------------ update (im_agent.py:412-473 + a2c_common train_actor_critic)
    def _calc_advs(self, batch):
        adv = (batch["returns"] - batch["values"]).sum(-1)
        if self.normalize_advantage:
            adv = vdist.normalize_advantages(adv, batch["alive"], self.group)  # global masked mean / std over all ranks
        return adv

    def train_epoch(self):
        sync = torch.cuda.synchronize if self.device.type == "cuda" else (lambda: None)
        sync()
        t0 = time.perf_counter()
        batch = self.play_steps()
        sync()
        t1 = time.perf_counter()
        adv = self._calc_advs(batch)
        alive = batch["alive"]
        n, t = alive.shape
        # in-network features of the whole batch (training flavour of the obs kernel) and running statistics (RunningNorm.update)
        raw_enc = ImitationObs(self.task.context_padding)
        feats_raw = raw_enc.training(batch["obses"].contiguous(), batch["context_feat"])
        self.obs_norm.update(feats_raw, alive.reshape(-1), self.group)
        self._obs_norm_ready = True
        self._sync_obs_norm()
        feats = self.obs_norm.normalize(feats_raw).view(n, t, -1)
        values, returns = batch["values"], batch["returns"]
        if self.normalize_value:
            self.value_norm.update(returns.reshape(-1, 1), alive.reshape(-1), self.group)
            self._value_norm_ready = True
            values, returns = self.value_norm.normalize(values), self.value_norm.normalize(returns)
        old_nlp, actions = batch["neglogpacs"], batch["actions"]
        world = vdist.dist.get_world_size(self.group) if vdist.dist.is_initialized() else 1
        params = list(self.actor.parameters()) + list(self.critic.parameters())
        info = {"a_loss": [], "c_loss": [], "kl": []}
        for _ in range(self.mini_epochs):
            perm = torch.randperm(n, device=self.device, generator=self.action_gen)
            for i in range(0, n - self.minibatch_envs + 1, self.minibatch_envs):
                idx = perm[i:i + self.minibatch_envs]
                f, a_ = feats[idx].reshape(-1, feats.shape[-1]), actions[idx].reshape(-1, actions.shape[-1])
                m = alive[idx].reshape(-1)
                mu = self.actor(f)
                nlp = neglogp(a_, mu, self.logstd)
                ratio = torch.exp(old_nlp[idx].reshape(-1) - nlp)
                ad = adv[idx].reshape(-1)
                a_loss = torch.max(-ad * ratio, -ad * torch.clamp(ratio, 1.0 - self.e_clip, 1.0 + self.e_clip))
                v = self.critic(f)
                c_loss = (v - returns[idx].reshape(-1, 1)) ** 2
                denom = torch.clamp(m.sum(), min=1.0)
                a_l, c_l = (a_loss * m).sum() / denom, (c_loss.squeeze(-1) * m).sum() / denom
                loss = a_l + self.critic_coef * c_l
                for p in params:
                    p.grad = None
                loss.backward()
                if world > 1:  # data-parallel over env shards: one flat all-reduce of the gradients over RCCL, averaged
                    flat = torch.cat([p.grad.reshape(-1) for p in params])
                    vdist.dist.all_reduce(flat, group=self.group)
                    flat /= world
                    o = 0
                    for p in params:
                        p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                        o += p.numel()
                nn.utils.clip_grad_norm_(params, self.grad_norm)
                self.optimizer.step()
                info["a_loss"].append(a_l.detach())
                info["c_loss"].append(c_l.detach())
                info["kl"].append((0.5 * ((mu.detach() - batch["mus"][idx].reshape(-1, mu.shape[-1])) ** 2 / torch.exp(2 * self.logstd)).sum(-1) * m).sum() / denom)
        sync()
        t2 = time.perf_counter()
        acc, sub_acc = batch["stats"]
        acc_h, sub_h = acc.tolist(), sub_acc.tolist()  # the ONE read-back of the epoch's statistics
        frames = batch["played_frames"]
        self.epoch_num += 1
        self.frame += frames
        play_time, update_time = t1 - t0, t2 - t1
        return {"play_time": play_time, "update_time": update_time, "total_time": t2 - t0, "frames": frames,
                "fps_step": frames / play_time, "fps_total": frames / (t2 - t0),
                "mean_rewards": acc_h[1] / max(acc_h[0], 1.0), "mean_lengths": acc_h[2] / max(acc_h[0], 1.0),
                "step_rewards": acc_h[4] / max(acc_h[3], 1.0), "step_sub_rewards": [s / max(acc_h[3], 1.0) for s in sub_h],
                "alive_ratio": float(alive.sum().item() / alive.numel()),
                "a_loss": float(torch.stack(info["a_loss"]).mean()), "c_loss": float(torch.stack(info["c_loss"]).mean()),
                "kl": float(torch.stack(info["kl"]).mean())}

    def format_epoch_line(self, r):
        """the reference's per-epoch line (im_agent.py:211-214)"""
        return ("%d\tT_play %.2f\tT_update %.2f\tstep_rewards %.4f %s\teps_len %.2f\talive %.2f\tfps step %d\tfps total %d"
                % (self.epoch_num, r["play_time"], r["update_time"], r["step_rewards"], "[" + ",".join("%.4f" % s for s in r["step_sub_rewards"]) + "]",
                   r["mean_lengths"], r["alive_ratio"], r["fps_step"], r["fps_total"]))
