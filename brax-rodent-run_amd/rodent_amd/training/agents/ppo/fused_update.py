"""One PPO minibatch gradient without an autograd graph: the GPU learner's fast path.

row indices of the minibatch inside the unroll buffer (no gathered copy of the observations) -> `rr_mlp_forward` (both networks, one f32-MFMA launch, pre-activations kept) ->
`rr_ppo_loss` (GAE, normalised advantages, surrogate / value / entropy terms AND d loss / d network outputs, three launches) ->
explicit backward (`rr_mlp_value_backward` / `rr_policy_backward`: the delta chains of the two networks, one launch each;
`rr_mlp_weight_grad`: every dW as a split-row matrix-core product) writing straight into the flat gradient buffer
(`distributed.FlatGrads`).

It computes what `losses.compute_ppo_loss` + `loss.backward()` compute [UP brax.training.agents.ppo.losses /
brax.training.gradients; REF brax_rodent_run_ppo.py:97-114] -- `tests/test_gpu_ppo.py` holds the two paths
against each other and against float64 -- with ~40 launches per minibatch instead of ~190 (no per-leaf gathers, no loss
elementwise chain, no autograd accumulation adds, no gradient zero fill).
"""
from __future__ import annotations

import torch

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
        # the policy network's backward (35 us chain, small products) runs on a second stream next to the value network's (the
        # matrix-core kernels that fill the GPU): fork / join through events, inside a HIP-graph capture as well.  RR_LEARNER_STREAMS=0: one stream
        import os
        self.side = torch.cuda.Stream(dev) if dev.type == "cuda" and os.environ.get("RR_LEARNER_STREAMS", "1") == "1" else None
        for p in list(policy_net.parameters()) + list(value_net.parameters()):
            if p.grad is None:
                p.grad = torch.zeros_like(p)

    def _policy_backward(self, pre, g, obs, rows, mean, std):
        """Policy network: the delta chain of the 32-wide stack is one launch (`rr_policy_backward`: delta_j, h_j = silu(z_j) over the
        forward's dump, bias gradients).  g [n, P] covers the first n rows of the minibatch (the bootstrap rows behind them carry no
        policy gradient).  Returns the weight-gradient products to be taken (`rr_mlp_weight_grad_batch` items)."""
        layers = self.policy_net.layers
        nh = len(layers) - 1
        n = g.shape[0]
        delta, h = hip.policy_backward(g, layers[nh].weight, [None] + [layers[j].weight for j in range(1, nh)], pre,
                                       [layers[j].bias.grad for j in range(nh)], self.bufs)
        torch.sum(g, 0, out=layers[nh].bias.grad)
        items = [dict(delta=g, act=h[nh - 1, :n], out=layers[nh].weight.grad)]
        items += [dict(delta=delta[j], act=h[j - 1, :n], out=layers[j].weight.grad) for j in range(nh - 1, 0, -1)]
        items.append(dict(delta=delta[0], act=obs, out=layers[0].weight.grad, rows=rows, mean=mean, std=std, delta_colsum=layers[0].bias.grad))
        return items

    def _value_backward(self, pre, g, obs, rows, mean, std):
        """Value network: the delta chain (dX products, silu', h = silu(z), bias gradients) is ONE matrix-core launch
        (`rr_mlp_value_backward`); returns the weight-gradient products of its outputs."""
        layers = self.value_net.layers
        nh = len(layers) - 1
        wt = [None] + [layers[j].weight.t().contiguous() for j in range(1, nh)]
        delta, h = hip.mlp_value_backward(g, layers[nh].weight, wt, pre, [layers[j].bias.grad for j in range(nh)], self.bufs)
        torch.sum(g, 0, keepdim=True, out=layers[nh].bias.grad)
        items = [dict(delta=g.unsqueeze(1), act=h[nh - 1], out=layers[nh].weight.grad)]
        items += [dict(delta=delta[j], act=h[j - 1], out=layers[j].weight.grad) for j in range(nh - 1, 0, -1)]
        items.append(dict(delta=delta[0], act=obs, out=layers[0].weight.grad, rows=rows, mean=mean, std=std, delta_colsum=layers[0].bias.grad))
        return items

    @torch.no_grad()
    def __call__(self, data, idx, mean, std, generator=None):
        """Fills every parameter's `.grad` with d total_loss / d parameter of the minibatch `idx`; returns the metrics dict."""
        T = self.T
        B = idx.numel()
        K = data["obs"].shape[-1]
        A = self.dist.event_size
        # the minibatch is addressed inside the unroll buffer: sample (t, b) = row idx[b] * (T + 1) + t, time-major order
        rows = (idx.unsqueeze(0) * (T + 1) + self.trange.unsqueeze(1)).reshape(-1)
        obs = data["obs"].reshape(-1, K)
        pol, val, ppre, vpre = hip.mlp_forward(obs, mean, std, fused_mlp.net_params(self.policy_net), fused_mlp.net_params(self.value_net),
                                               want_pre=True, rows=rows)
        noise = torch.randn(T * B, A, device=obs.device, dtype=obs.dtype, generator=generator)     # the draw of dist.entropy
        g_pol, g_val, metrics = hip.ppo_loss(pol, val, data, idx, noise, T, out=self.bufs, **self.cfg)
        n = T * B                                                                           # the bootstrap rows carry no policy gradient
        if self.side is None:
            items = self._policy_backward(ppre, g_pol[:n], obs, rows[:n], mean, std) + self._value_backward(vpre, g_val, obs, rows, mean, std)
            hip.mlp_weight_grad_batch(items)        # all eleven dW = delta' h: one launch per tile shape + one reduction launch
        else:
            cur = torch.cuda.current_stream(obs.device)
            self.side.wait_stream(cur)              # fork: the policy branch needs the forward's dumps and the loss gradients
            with torch.cuda.stream(self.side):
                hip.mlp_weight_grad_batch(self._policy_backward(ppre, g_pol[:n], obs, rows[:n], mean, std))
            hip.mlp_weight_grad_batch(self._value_backward(vpre, g_val, obs, rows, mean, std))
            cur.wait_stream(self.side)              # join before anything reads the gradients (all-reduce, Adam) or frees the temporaries
        return {"total_loss": metrics[0], "policy_loss": metrics[1], "v_loss": metrics[2], "entropy_loss": metrics[3]}
