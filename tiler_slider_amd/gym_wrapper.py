"""Gymnasium-style five-tuple view of VecTilerSliderEnv.

The reference's step() returns (obs, done, info) with no reward and no terminated /
truncated split (ref: explainrl/environment/environment.py:100-143).  Learners written
against the Gymnasium vector API expect
    reset(seed=None) -> (obs, info)
    step(actions)    -> (obs, reward, terminated, truncated, info)
This adapter derives them from the kernel's flag byte: terminated = success (the board was
solved by this step), truncated = timeout without success; reward is the build-defined
Manhattan reward (include/tiler_slider.h) plus an optional bonus on success.  No dependency
on the gymnasium package.
"""
import torch

from .vec_env import VecTilerSliderEnv


class GymVecTilerSlider:
    def __init__(self, env: VecTilerSliderEnv, success_bonus=0.0):
        if env._reward is None:
            raise ValueError("build the VecTilerSliderEnv with with_reward=True")
        self.env = env
        self.num_envs = env.num_envs
        self.success_bonus = float(success_bonus)
        self.single_observation_shape = env.observation_shape
        self.n_actions = 4

    def reset(self, seed=None):
        """Levels are fixed at construction (as in the reference); `seed` is accepted and ignored."""
        return self.env.reset(), {}

    def step(self, actions):
        obs, done, info = self.env.step(actions)
        terminated = info["success"]
        truncated = info["timeout"] & ~terminated
        reward = info["reward"].to(torch.float32)
        if self.success_bonus:
            reward = reward + self.success_bonus * terminated.to(torch.float32)
        return obs, reward, terminated, truncated, info

    def action_masks(self):
        """bool [N, 4] legality mask of the current boards (environment.py:149-171)."""
        return self.env.get_valid_moves()

    def close(self):
        self.env.close()
