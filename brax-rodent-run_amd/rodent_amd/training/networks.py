"""PPO networks of `brax.training.agents.ppo.networks.make_ppo_networks` (defaults): policy MLP
obs -> 32 x4 -> 2*action_size, value MLP obs -> 256 x5 -> 1, swish on hidden layers,
lecun_uniform kernels, zero biases; `NormalTanhDistribution(min_std=1e-3)`.
On a GPU with these default shapes the forward passes run on the hand-written f32-MFMA kernel (`fused_mlp`, C ABI
`rr_mlp_forward`); other shapes / CPU tensors use the nn.Linear path (rocBLAS/hipBLASLt)."""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class MLP(nn.Module):
    def __init__(self, in_size: int, layer_sizes: Sequence[int]):
        super().__init__()
        sizes = [in_size] + list(layer_sizes)
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:]))
        for lin in self.layers:
            bound = math.sqrt(3.0 / lin.in_features)          # lecun_uniform: variance 1/fan_in
            nn.init.uniform_(lin.weight, -bound, bound)
            nn.init.zeros_(lin.bias)

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = lin(x)
            if i != len(self.layers) - 1:
                x = F.silu(x)
        return x


class NormalTanhDistribution:
    """Normal followed by tanh; parameters = concat(loc, pre-softplus scale)."""

    def __init__(self, event_size: int, min_std: float = 0.001):
        self.event_size = event_size
        self.param_size = 2 * event_size
        self.min_std = min_std

    def _params(self, logits):
        loc, s = torch.chunk(logits, 2, dim=-1)
        return loc, F.softplus(s) + self.min_std

    def sample_no_postprocessing(self, logits, generator=None):
        loc, scale = self._params(logits)
        eps = torch.randn(loc.shape, device=loc.device, dtype=loc.dtype, generator=generator)
        return loc + scale * eps

    def mode(self, logits):
        return torch.tanh(self._params(logits)[0])

    @staticmethod
    def postprocess(x):
        return torch.tanh(x)

    @staticmethod
    def _log_det_jac(x):
        return 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))

    def log_prob(self, logits, raw_actions):
        loc, scale = self._params(logits)
        lp = -0.5 * ((raw_actions - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2 * math.pi)
        return (lp - self._log_det_jac(raw_actions)).sum(-1)

    def entropy(self, logits, generator=None):
        loc, scale = self._params(logits)
        ent = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(scale)
        raw = self.sample_no_postprocessing(logits, generator)
        return (ent + self._log_det_jac(raw)).sum(-1)


class PPONetworks:
    def __init__(self, policy_network: MLP, value_network: MLP, parametric_action_distribution: NormalTanhDistribution):
        self.policy_network = policy_network
        self.value_network = value_network
        self.parametric_action_distribution = parametric_action_distribution


def make_ppo_networks(observation_size: int, action_size: int, policy_hidden_layer_sizes=(32,) * 4,
                      value_hidden_layer_sizes=(256,) * 5, device=None) -> PPONetworks:
    dist = NormalTanhDistribution(event_size=action_size)
    policy = MLP(observation_size, list(policy_hidden_layer_sizes) + [dist.param_size]).to(device)
    value = MLP(observation_size, list(value_hidden_layer_sizes) + [1]).to(device)
    return PPONetworks(policy, value, dist)


def make_inference_fn(ppo_networks: PPONetworks):
    """`make_policy(params, deterministic)` -> `policy(obs, key) -> (action, extras)` [UP ppo.networks]."""
    from . import running_statistics

    import copy
    import os
    from . import fused_mlp

    def make_policy(params, deterministic: bool = False):
        normalizer_params, policy_params = params[0], params[1]
        net = ppo_networks.policy_network
        if isinstance(policy_params, nn.Module):
            net = policy_params                              # the live training network or a snapshot of it
        elif isinstance(policy_params, dict):                # a checkpoint: load into a COPY, never into the training network
            net = copy.deepcopy(ppo_networks.policy_network)
            net.load_state_dict(policy_params)
        elif policy_params is not None:
            raise TypeError(f"policy params must be an nn.Module or a state dict, got {type(policy_params).__name__}")
        dist = ppo_networks.parametric_action_distribution
        fused = os.environ.get("RR_FUSED_MLP", "1") == "1" and fused_mlp.fusable(net, fused_mlp.POLICY_HIDDEN, 64)

        two_launch = (fused and isinstance(dist, NormalTanhDistribution) and type(dist) is NormalTanhDistribution and dist.event_size <= 32
                      and os.environ.get("RR_POLICY_ACT", "1") == "1")

        @torch.no_grad()
        def policy(observations, key_sample=None):
            if two_launch and observations.is_cuda and observations.dtype == torch.float32 and observations.dim() == 2:
                # the whole actor step (normalise, policy MLP with the first layer split over k, tanh-normal head) in two launches
                from .. import hip
                mean, std = (None, None) if normalizer_params is None else (normalizer_params.mean, normalizer_params.std)
                eps = None if deterministic else torch.randn(observations.shape[0], dist.event_size, device=observations.device,
                                                              dtype=observations.dtype, generator=key_sample)
                action, raw, lp, _ = hip.policy_act(observations.contiguous(), mean, std, fused_mlp.net_params(net), eps, dist.min_std)
                return (action, {}) if deterministic else (action, {"log_prob": lp, "raw_action": raw})
            if fused and observations.is_cuda and observations.dtype == torch.float32 and observations.dim() == 2:
                mean, std = (None, None) if normalizer_params is None else (normalizer_params.mean, normalizer_params.std)
                logits = fused_mlp.policy_logits(observations, mean, std, net)        # normalise + MLP: one MFMA launch
            else:
                x = observations if normalizer_params is None else running_statistics.normalize(observations, normalizer_params)
                logits = net(x)
            if deterministic:
                return dist.mode(logits), {}
            if fused and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and type(dist) is NormalTanhDistribution:
                from .. import hip                    # sample + tanh + log-prob: one launch instead of ~20 (same normal draws)
                eps = torch.randn(logits.shape[0], dist.event_size, device=logits.device, dtype=logits.dtype, generator=key_sample)
                action, raw, lp = hip.policy_sample(logits.contiguous(), eps, dist.min_std)
                return action, {"log_prob": lp, "raw_action": raw}
            raw = dist.sample_no_postprocessing(logits, key_sample)
            return dist.postprocess(raw), {"log_prob": dist.log_prob(logits, raw), "raw_action": raw}

        return policy

    return make_policy
