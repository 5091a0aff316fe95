"""A tiny pure-torch env with the Rodent surface, so the PPO / wrapper logic can be tested on CPU."""
import numpy as np
import torch

from rodent_amd.envs.base import State


class PointEnv:
    """2-D point mass; reward = -|x - target|; done when |x| > 3."""

    def __init__(self, num_envs=8, device="cpu"):
        self.num_envs = num_envs
        self.device = torch.device(device)
        self.observation_size = 4
        self.action_size = 2
        self.dt = 0.1

    def with_num_envs(self, n, device=None):
        return PointEnv(n, device or self.device)

    def _obs(self, x, v):
        return torch.cat([x, v], dim=-1)

    def reset(self, rng):
        keys = np.asarray(rng, dtype=np.uint32).reshape(self.num_envs, 2)
        g = torch.Generator().manual_seed(int(keys[0, 0]) + int(keys[0, 1]))
        x = (torch.rand(self.num_envs, 2, generator=g) - 0.5).to(self.device)
        v = torch.zeros_like(x)
        z = torch.zeros(self.num_envs, device=self.device)
        return State({"x": x, "v": v}, self._obs(x, v), z, z.clone(), {"dist": z.clone()},
                     {"cur_frame": torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)})

    def step(self, state, action):
        ps = state.pipeline_state
        v = 0.9 * ps["v"] + 0.1 * action
        x = ps["x"] + self.dt * v
        dist = x.norm(dim=-1)
        info = dict(state.info)
        info["cur_frame"] = info["cur_frame"] + 1
        return state.replace(pipeline_state={"x": x, "v": v}, obs=self._obs(x, v), reward=-dist,
                             done=(dist > 3).float(), metrics={"dist": dist}, info=info)
