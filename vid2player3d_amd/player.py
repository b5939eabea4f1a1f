"""The evaluation rollout of the imitation policy around the rollout engine: what `run.py --test` runs in the reference,

    ImitatorPlayer.get_action / env_step / run / restore / load_pretrained     embodied_pose/players/im_player.py:13-311
    CommonPlayer (net config, env_reset)                                      embodied_pose/learning/common_player.py:10-194

method for method, with the network side shared with the training agent (`ppo.PolicyInference`: the fused 734-d observation kernel, the
actor MLP, the policy-head kernel with the residual action).  Unlike a training epoch the rollout runs PAST the 32-step context window
without a reset: every `task.context_length` steps the player has the task rebuild the window around the current clip times
(`task._init_context(task._reset_ref_motion_ids, task._cur_ref_motion_times)`, im_player.py:238-240 -> `v2p_env_context`).

The bookkeeping of finished episodes is the reference's (`step_dones = done * (1 - prev_dones)`, sums over the envs that finished in this
step, the round ends when enough games were played or when env 0 is done), computed on the device and read back as ONE 4-number tensor
per step (the reference's `.nonzero()`, two `.sum().item()` and `done[0]` are 4 host reads): the loop's exits depend on those values,
so one read per step is what the control flow needs.  rl_games' BasePlayer [1.1.4, from memory] supplies the defaults `games_num` 2000,
`determenistic` True (its spelling), `n_game_life` 1, `print_stats` True, `max_steps` 27000 from the `player` block of the yaml.
"""
import os

import torch

from . import _lib
from .learning import ImitationObs
from .ppo import ImitatorNetwork, PolicyInference, ValueMeanStd


class ImitatorPlayer(PolicyInference):
    def __init__(self, task, units=(1024, 1024, 512), sigma_init=-1.756, residual_action=True, games_num=2000, deterministic=True,
                 n_game_life=1, print_stats=True, max_steps=108000 // 4, normalize_value=True, seed=0, name="Humanoid", network_path=None,
                 model=None, value_mean_std=None, log=print):
        self.task = task
        self.device = torch.device(task.device)
        self.num_actors, self.num_actions = task.num_envs, task.num_actions
        self.games_num, self.is_determenistic, self.n_game_life = int(games_num), bool(deterministic), int(n_game_life)
        self.print_stats, self.max_steps = bool(print_stats), int(max_steps)
        self.normalize_value = normalize_value
        self.config_name, self.network_path = name, network_path
        self.log = log or (lambda *a: None)
        if model is None:
            with torch.random.fork_rng(devices=[]):
                torch.manual_seed(seed)
                model = ImitatorNetwork(self.num_actions, units, sigma_init, residual_action, self.device)
        self.model = model
        self.model.eval()
        self.model.running_obs.eval()
        self.value_mean_std = value_mean_std or ValueMeanStd(self.device)
        self.value_mean_std.eval()
        self.obs_enc = ImitationObs(task.context_padding)
        self._lib = _lib.load()
        self.action_gen = torch.Generator(device=self.device)
        self.action_gen.manual_seed(seed)
        self._zero_noise = torch.zeros((self.num_actors, self.num_actions), device=self.device)

    @classmethod
    def from_config(cls, task, params, **overrides):
        """`params`: the `params` block of the reference's yaml (network as in PPOAgent.from_config; `config.player` = rl_games' player
        options)."""
        cfg, net = params.get("config", {}), params.get("network", {})
        space = net.get("space", {}).get("continuous", {})
        pc = cfg.get("player", {}) or {}
        if cfg.get("normalize_input", False):
            raise NotImplementedError("normalize_input is not built (off in both reference configs: the network normalises inside, RunningNorm)")
        kw = dict(units=tuple(net.get("mlp", {}).get("units", (1024, 1024, 512))), sigma_init=float(space.get("sigma_init", {}).get("val", -1.756)),
                  residual_action=net.get("residual_action", True), games_num=pc.get("games_num", 2000),
                  deterministic=pc.get("determenistic", pc.get("deterministic", True)), n_game_life=pc.get("n_game_life", 1),
                  print_stats=pc.get("print_stats", True), normalize_value=cfg.get("normalize_value", True), seed=params.get("seed", 0),
                  name=cfg.get("name", "Humanoid"))
        kw.update(overrides)
        return cls(task, **kw)

    @classmethod
    def from_agent(cls, agent, **kw):
        """A player over the training agent's own network (the weights are shared, not copied)."""
        return cls(agent.task, model=agent.model, value_mean_std=agent.value_mean_std, normalize_value=agent.normalize_value,
                   name=getattr(agent, "config_name", "Humanoid"), **kw)

    # ------------------------------------------------------------------ checkpoints (im_player.py:43-100)
    def restore(self, cp_name):
        """`<network_path>/<name>_<cp_name>.pth` like the reference, or a path; None / 'base' = nothing to load."""
        if cp_name is None or cp_name == "base":
            self.log("No checkpoint provided.")
            return
        path = cp_name if os.path.exists(cp_name) else os.path.join(self.network_path or ".", "%s_%s.pth" % (self.config_name, cp_name))
        self.set_weights(torch.load(path, map_location=self.device, weights_only=False))

    def load_pretrained(self, path):
        self.set_weights(torch.load(path, map_location=self.device, weights_only=False))

    def set_weights(self, weights):
        self.model.load_reference_state_dict(weights["model"])
        r = weights.get("reward_mean_std")
        if r is not None:  # (the player of the reference does not read values; get_action_values does)
            v = self.value_mean_std
            v.running_mean = torch.as_tensor(r["running_mean"], dtype=torch.float64, device=self.device).reshape(1).clone()
            v.running_var = torch.as_tensor(r["running_var"], dtype=torch.float64, device=self.device).reshape(1).clone()
            v.count = torch.as_tensor(r["count"], dtype=torch.float64, device=self.device).reshape(()).clone()
        self.model.running_obs._seen = None

    # ------------------------------------------------------------------ one step (im_player.py:120-190)
    @torch.no_grad()
    def get_action_values(self, obs, t):
        obs = obs["obs"] if isinstance(obs, dict) else obs
        self._sync_obs_norm()
        feat = self._features(self.task, obs, t)
        mu = self.model.actor(feat)
        noise = torch.randn(mu.shape, device=mu.device, generator=self.action_gen)
        action, sigma, nlp = self._policy_head(self.task, mu, noise, t)
        return {"actions": action, "mus": mu, "sigmas": sigma, "neglogpacs": nlp, "values": self._value(feat)}

    @torch.no_grad()
    def get_action(self, obs_dict, is_determenistic=False):
        """obs_dict: {'obs': [N,461], 't': step inside the context window}; the mean (residual included) or a sample."""
        obs, t = obs_dict["obs"], obs_dict["t"]
        self._sync_obs_norm()
        mu = self.model.actor(self._features(self.task, obs, t))
        noise = self._zero_noise if is_determenistic else torch.randn(mu.shape, device=mu.device, generator=self.action_gen)
        action, _, _ = self._policy_head(self.task, mu, noise, t)
        return mu if is_determenistic else action

    def env_reset(self, env_ids=None):
        self.task.reset(env_ids)
        return {"obs": self.task.obs_buf}

    def env_step(self, actions):
        t = self.task
        t.step(actions)
        return {"obs": t.obs_buf}, t.rew_buf, t.reset_buf, t.extras

    def _post_step(self, info):
        return

    # ------------------------------------------------------------------ the evaluation loop (im_player.py:192-311)
    def run(self):
        n_games = self.games_num * self.n_game_life
        task, dev = self.task, self.device
        sum_rewards = sum_steps = 0.0
        games_played = rounds = total_steps = 0
        for _ in range(n_games):
            if games_played >= n_games:
                break
            obs_dict = self.env_reset()
            rounds += 1
            cr = torch.zeros(self.num_actors, device=dev)
            steps = torch.zeros(self.num_actors, device=dev)
            prev_dones = torch.zeros(self.num_actors, device=dev)
            task.render_vis(init=True)
            for n in range(self.max_steps):
                t = n % task.context_length
                if n > 0 and t == 0:
                    task._init_context(task._reset_ref_motion_ids, task._cur_ref_motion_times)
                obs_dict["t"], obs_dict["global_t_offset"] = t, n - t
                action = self.get_action(obs_dict, self.is_determenistic)
                obs_dict, r, done, info = self.env_step(action)
                total_steps += 1
                done = done.to(torch.float32)
                cr += r
                steps += 1.0
                task.render_vis()
                self._post_step(info)
                step_dones = done * (1.0 - prev_dones)
                done_count, cur_rewards, cur_steps, done0 = torch.stack([step_dones.sum(), (cr * step_dones).sum(), (steps * step_dones).sum(), done[0]]).tolist()
                done_count = int(done_count)
                games_played += done_count
                # (the reference clears these only in steps that finish a game; the entries of envs that are done are never read again)
                cr *= 1.0 - done
                steps *= 1.0 - done
                if done_count > 0:
                    sum_rewards += cur_rewards
                    sum_steps += cur_steps
                    if self.print_stats:
                        self.log("reward: %s steps: %s" % (cur_rewards / done_count, cur_steps / done_count))
                    if self.num_actors == 1 or games_played >= n_games:
                        break
                prev_dones = done
                if done0:
                    break
        res = {"sum_rewards": sum_rewards, "sum_steps": sum_steps, "games_played": games_played, "rounds": rounds, "env_steps": total_steps * self.num_actors,
               "av_reward": sum_rewards / max(games_played, 1) * self.n_game_life, "av_steps": sum_steps / max(games_played, 1) * self.n_game_life}
        self.log(sum_rewards)
        self.log("av reward: %s av steps: %s" % (res["av_reward"], res["av_steps"]))
        return res
