"""PPO losses with the arithmetic of `brax.training.agents.ppo.losses`:
truncation-aware GAE (reverse scan over the unroll) and the clipped surrogate objective with
device-local advantage normalisation (no collective, as upstream)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_: float = 1.0, discount: float = 0.99):
    """All inputs time-major [T, B]; bootstrap_value [B].  Returns (vs [T,B], advantages [T,B])."""
    if values.is_cuda and values.dtype == torch.float32:
        from .... import hip                      # one HIP launch instead of ~6 small kernels per time step
        return hip.compute_gae(truncation, termination, rewards, values.detach(), bootstrap_value.detach(), lambda_, discount)
    truncation_mask = 1 - truncation
    values_t_plus_1 = torch.cat([values[1:], bootstrap_value.unsqueeze(0)], dim=0)
    deltas = (rewards + discount * (1 - termination) * values_t_plus_1 - values) * truncation_mask
    acc = torch.zeros_like(bootstrap_value)
    out = []
    for t in range(values.shape[0] - 1, -1, -1):
        acc = deltas[t] + discount * (1 - termination[t]) * truncation_mask[t] * lambda_ * acc
        out.append(acc)
    vs_minus_v_xs = torch.stack(out[::-1], dim=0)
    vs = vs_minus_v_xs + values
    vs_t_plus_1 = torch.cat([vs[1:], bootstrap_value.unsqueeze(0)], dim=0)
    advantages = (rewards + discount * (1 - termination) * vs_t_plus_1 - values) * truncation_mask
    return vs.detach(), advantages.detach()


def compute_ppo_loss(policy_logits, baseline, bootstrap_value, data: Dict[str, torch.Tensor], dist, entropy_cost=1e-4,
                     discounting=0.9, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.3,
                     normalize_advantage=True, generator=None, advantage_stats_fn=None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """`data` leaves are time-major [T, B(, ...)]: reward, discount, truncation, raw_action, log_prob."""
    rewards = data["reward"] * reward_scaling
    truncation = data["truncation"]
    termination = (1 - data["discount"]) * (1 - truncation)
    target_action_log_probs = dist.log_prob(policy_logits, data["raw_action"])
    behaviour_action_log_probs = data["log_prob"]
    vs, advantages = compute_gae(truncation, termination, rewards, baseline.detach(), bootstrap_value.detach(),
                                 lambda_=gae_lambda, discount=discounting)
    if normalize_advantage:
        if advantage_stats_fn is not None:          # global minibatch statistics (extension; the reference is device-local)
            mean, std = advantage_stats_fn(advantages)
            advantages = (advantages - mean) / (std + 1e-8)
        else:
            advantages = (advantages - advantages.mean()) / (advantages.std(unbiased=False) + 1e-8)
    rho_s = torch.exp(target_action_log_probs - behaviour_action_log_probs)
    surrogate_loss1 = rho_s * advantages
    surrogate_loss2 = torch.clamp(rho_s, 1 - clipping_epsilon, 1 + clipping_epsilon) * advantages
    policy_loss = -torch.mean(torch.minimum(surrogate_loss1, surrogate_loss2))
    v_error = vs - baseline
    v_loss = torch.mean(v_error * v_error) * 0.5 * 0.5
    entropy = torch.mean(dist.entropy(policy_logits, generator))
    entropy_loss = entropy_cost * -entropy
    total_loss = policy_loss + v_loss + entropy_loss
    return total_loss, {"total_loss": total_loss.detach(), "policy_loss": policy_loss.detach(),
                        "v_loss": v_loss.detach(), "entropy_loss": entropy_loss.detach()}
