"""VecTask wrappers: the surface rl_games sees (embodied_pose/env/tasks/vec_task.py:16-63, 120-138,
vec_task_wrappers.py:22-28, run.py:93-137).  gym.spaces is not required: `Box` is a minimal stand-in
with the attributes rl_games reads (`shape`, `low`, `high`, `dtype`)."""
import numpy as np
import torch


class Box:
    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32


class VecTask:
    def __init__(self, task, rl_device, clip_observations=5.0, clip_actions=1.0):
        self.task = task
        self.num_environments = task.num_envs
        self.num_agents = 1
        self.num_observations = task.num_obs
        self.num_states = task.num_states
        self.num_actions = task.num_actions
        self.obs_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.act_space = Box(np.ones(self.num_actions) * -1.0, np.ones(self.num_actions) * 1.0)
        self.clip_obs = clip_observations
        self.clip_actions = clip_actions
        self.rl_device = rl_device

    def get_number_of_agents(self):
        return self.num_agents

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self):
        return self.num_environments

    @property
    def num_acts(self):
        return self.num_actions

    @property
    def num_obs(self):
        return self.num_observations


class VecTaskPython(VecTask):
    def get_state(self):
        return torch.clamp(self.task.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def step(self, actions):
        # clamp makes a fresh tensor, so the task's in-place masking never touches the policy's output (vec_task.py:126)
        actions_tensor = torch.clamp(actions, -self.clip_actions, self.clip_actions).to(self.task.device).contiguous()
        self.task.step(actions_tensor)
        return (torch.clamp(self.task.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device), self.task.rew_buf.to(self.rl_device),
                self.task.reset_buf.to(self.rl_device), self.task.extras)


class VecTaskPythonWrapper(VecTaskPython):
    """vec_task_wrappers.py:22-28: reset(env_ids) resets the task and returns clamped observations."""

    def __init__(self, task, rl_device, clip_observations=5.0, clip_actions=1.0):
        super().__init__(task, rl_device, clip_observations, clip_actions)

    def reset(self, env_ids=None):
        self.task.reset(env_ids)
        return torch.clamp(self.task.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)


class RLGPUEnv:
    """run.py:93-137: the vecenv rl_games instantiates ('RLGPU')."""

    def __init__(self, vec_task):
        self.env = vec_task
        self.use_global_obs = vec_task.num_states > 0
        self.full_state = {"obs": self.env.reset()}
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()

    def step(self, action):
        next_obs, reward, is_done, info = self.env.step(action)
        self.full_state["obs"] = next_obs
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state, reward, is_done, info
        return self.full_state["obs"], reward, is_done, info

    def reset(self, env_ids=None):
        self.full_state["obs"] = self.env.reset(env_ids)
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state
        return self.full_state["obs"]

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if self.use_global_obs:
            info["state_space"] = self.env.state_space
        return info
