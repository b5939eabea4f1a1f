"""The adapter rl_games talks to, so that this repository runs stand-alone (a maintainer keeps the reference's own wrappers: the task
object has every attribute they read).  One class does what the reference spreads over `VecTask` / `VecTaskPython` /
`VecTaskPythonWrapper` (embodied_pose/env/tasks/vec_task.py:16-63, 120-138; vec_task_wrappers.py:22-28): clip what goes in and out, move
it to the RL device, hand the rest through to the task.  `RLGPUEnv` is the vecenv rl_games instantiates as 'RLGPU' (run.py:93-137).
gym.spaces is not required: `Box` carries the four attributes rl_games reads."""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class Box:
    low: np.ndarray
    high: np.ndarray

    @classmethod
    def symmetric(cls, size, bound):
        edge = np.full(size, bound, dtype=np.float32)
        return cls(-edge, edge)

    @property
    def shape(self):
        return self.low.shape

    dtype = np.float32


class VecTaskPythonWrapper:
    num_agents = 1

    def __init__(self, task, rl_device, clip_observations=5.0, clip_actions=1.0):
        self.task, self.rl_device = task, rl_device
        self.clip_obs, self.clip_actions = clip_observations, clip_actions
        self.num_envs = self.num_environments = task.num_envs
        self.num_obs = self.num_observations = task.num_obs
        self.num_acts = self.num_actions = task.num_actions
        self.num_states = task.num_states
        self.observation_space = self.obs_space = Box.symmetric(self.num_obs, np.inf)
        self.state_space = Box.symmetric(self.num_states, np.inf)
        self.action_space = self.act_space = Box.symmetric(self.num_actions, 1.0)

    def _out(self, t, clip=True):
        return (torch.clamp(t, -self.clip_obs, self.clip_obs) if clip else t).to(self.rl_device)

    def get_number_of_agents(self):
        return self.num_agents

    def get_state(self):
        return self._out(self.task.states_buf)

    def reset(self, env_ids=None):
        self.task.reset(env_ids)
        return self._out(self.task.obs_buf)

    def step(self, actions):
        # (clamp makes a fresh tensor: the task's in-place masking of finished envs never touches the policy's own output, vec_task.py:126)
        self.task.step(torch.clamp(actions, -self.clip_actions, self.clip_actions).to(self.task.device).contiguous())
        t = self.task
        return self._out(t.obs_buf), self._out(t.rew_buf, False), self._out(t.reset_buf, False), t.extras


VecTask = VecTaskPython = VecTaskPythonWrapper  # the reference's class names


class RLGPUEnv:
    def __init__(self, vec_task):
        self.env = vec_task
        self.use_global_obs = vec_task.num_states > 0
        self.full_state = {}
        self.reset()

    def _view(self, obs):
        self.full_state["obs"] = obs
        if not self.use_global_obs:
            return obs
        self.full_state["states"] = self.env.get_state()
        return self.full_state

    def step(self, action):
        obs, reward, is_done, info = self.env.step(action)
        return self._view(obs), reward, is_done, info

    def reset(self, env_ids=None):
        return self._view(self.env.reset(env_ids))

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if self.use_global_obs:
            info["state_space"] = self.env.state_space
        return info
