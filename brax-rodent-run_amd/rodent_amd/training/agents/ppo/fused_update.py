"""One PPO minibatch gradient without an autograd graph: the GPU learner's fast path.

gather (time-major, one index_select) -> `rr_mlp_forward` (both networks, one f32-MFMA launch, pre-activations kept) ->
`rr_ppo_loss` (GAE, normalised advantages, surrogate / value / entropy terms AND d loss / d network outputs, three launches) ->
explicit backward whose matrix products write straight into the flat gradient buffer (`distributed.FlatGrads`).

It computes what `losses.compute_ppo_loss` + `loss.backward()` compute [UP brax.training.agents.ppo.losses /
brax.training.gradients; REF brax_rodent_run_ppo.py:97-114] -- `tests/test_gpu_ppo.py` holds the two paths
against each other and against float64 -- with ~60 launches per minibatch instead of ~190 (no per-leaf gathers, no loss
elementwise chain, no autograd accumulation adds, no gradient zero fill).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .... import hip
from ... import fused_mlp


class FusedUpdate:
    def __init__(self, policy_net, value_net, dist, unroll_length: int, *, entropy_cost, discounting, reward_scaling, gae_lambda,
                 clipping_epsilon, normalize_advantage):
        self.policy_net, self.value_net, self.dist, self.T = policy_net, value_net, dist, unroll_length
        self.cfg = dict(entropy_cost=entropy_cost, discounting=discounting, reward_scaling=reward_scaling, gae_lambda=gae_lambda,
                        clipping_epsilon=clipping_epsilon, normalize_advantage=normalize_advantage, min_std=dist.min_std)
        dev = policy_net.layers[0].weight.device
        self.trange = torch.arange(unroll_length + 1, device=dev)
        self.bufs = {}
        for p in list(policy_net.parameters()) + list(value_net.parameters()):
            if p.grad is None:
                p.grad = torch.zeros_like(p)

    @staticmethod
    def _backward_into_grads(layers, pre, delta, obs, mean, std):
        """dW_l = delta_l' h_{l-1}, db_l = sum delta_l, delta_{l-1} = (delta_l W_l) * silu'(z_{l-1}), written into `.grad`."""
        for l in range(len(layers) - 1, -1, -1):
            W, b = layers[l].weight, layers[l].bias
            torch.sum(delta, 0, out=b.grad)
            if l == 0:
                torch.mm(delta.t(), obs, out=W.grad)
                if mean is not None:         # the kernel normalised on the fly: dW_1 = (delta' obs - (sum delta) mean') / std
                    W.grad.addr_(b.grad, mean, alpha=-1.0).div_(std)
            else:
                torch.mm(delta.t(), F.silu(pre[l - 1]), out=W.grad)
                delta = torch.ops.aten.silu_backward(delta @ W, pre[l - 1])

    @torch.no_grad()
    def __call__(self, data, idx, mean, std, generator=None):
        """Fills every parameter's `.grad` with d total_loss / d parameter of the minibatch `idx`; returns the metrics dict."""
        T = self.T
        B = idx.numel()
        K = data["obs"].shape[-1]
        A = self.dist.event_size
        rows = (idx.unsqueeze(0) * (T + 1) + self.trange.unsqueeze(1)).reshape(-1)         # time-major: row t*B + b
        obs = data["obs"].reshape(-1, K).index_select(0, rows)
        pol, val, ppre, vpre = hip.mlp_forward(obs, mean, std, fused_mlp.net_params(self.policy_net), fused_mlp.net_params(self.value_net),
                                               want_pre=True)
        noise = torch.randn(T * B, A, device=obs.device, dtype=obs.dtype, generator=generator)     # the draw of dist.entropy
        g_pol, g_val, metrics = hip.ppo_loss(pol, val, data, idx, noise, T, out=self.bufs, **self.cfg)
        n = T * B                                                                           # the bootstrap rows carry no policy gradient
        self._backward_into_grads(self.policy_net.layers, ppre[:, :n], g_pol[:n], obs[:n], mean, std)
        self._backward_into_grads(self.value_net.layers, vpre, g_val.unsqueeze(1), obs, mean, std)
        return {"total_loss": metrics[0], "policy_loss": metrics[1], "v_loss": metrics[2], "entropy_loss": metrics[3]}
