"""Actor / critic forward on the hand-written f32-MFMA kernel (C ABI `rr_mlp_forward`, csrc/rr_mlp.h) with an explicit
backward pass, as one `torch.autograd.Function`.

Replaces, for the `make_ppo_networks` default shapes (policy obs -> 32 x4 -> 2*action_size, value obs -> 256 x5 -> 1, SiLU),
the ~40 library launches of `normalize -> policy_net -> value_net` in `ppo.losses.compute_ppo_loss`'s forward
[UP brax.training.agents.ppo.losses; SURVEY.md a22 / a25] by ONE launch that reads the observation tile once for both
networks.  The kernel also writes the hidden pre-activations, from which `backward` forms the parameter gradients with plain
matrix products (rocBLAS/hipBLASLt): dW_l = delta_l' h_{l-1},
delta_{l-1} = silu_backward(delta_l W_l, z_{l-1}); the first layer's dW uses the raw observations and folds the normaliser
in afterwards, so the normalised copy of the minibatch is never written.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import hip

POLICY_HIDDEN, VALUE_HIDDEN = 32, 256


def net_params(mlp):
    """(weights, biases) of an `MLP` in the order / layout the kernel takes (nn.Linear: [out, in])."""
    return [l.weight for l in mlp.layers], [l.bias for l in mlp.layers]


def fusable(mlp, hidden: int, max_out: int) -> bool:
    ls = list(mlp.layers)
    return (2 <= len(ls) <= 8 and all(l.out_features == hidden for l in ls[:-1]) and ls[-1].out_features <= max_out
            and ls[0].weight.is_cuda and ls[0].weight.dtype == torch.float32)


class _ActorCritic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, mean, std, n_pol, *params):
        pw, pb = list(params[0:2 * n_pol:2]), list(params[1:2 * n_pol:2])
        vw, vb = list(params[2 * n_pol::2]), list(params[2 * n_pol + 1::2])
        need = any(p.requires_grad for p in params)
        with torch.no_grad():
            pol, val, ppre, vpre = hip.mlp_forward(obs, mean, std, ([w.detach() for w in pw], [b.detach() for b in pb]),
                                                   ([w.detach() for w in vw], [b.detach() for b in vb]), want_pre=need)
        ctx.n_pol = n_pol
        ctx.has_norm = mean is not None
        if need:
            ctx.save_for_backward(obs, *([mean, std] if mean is not None else []), ppre, vpre, *params)
        return pol, val

    @staticmethod
    def backward(ctx, g_pol, g_val):
        saved = list(ctx.saved_tensors)
        obs = saved.pop(0)
        mean, std = (saved.pop(0), saved.pop(0)) if ctx.has_norm else (None, None)
        ppre, vpre = saved[0], saved[1]
        params, n_pol = saved[2:], ctx.n_pol
        grads = [None] * len(params)

        def net_backward(ws, pre, delta, base):
            for l in range(len(ws) - 1, -1, -1):
                dsum = delta.sum(0)
                if l == 0 and mean is not None:
                    # dW_1 = delta' ((obs - mean) / std) = (delta' obs - (sum delta) mean') / std: the normalised observations
                    # (114 MB at the learner's shape) are never materialised
                    grads[base] = (delta.t() @ obs).addr_(dsum, mean, alpha=-1.0).div_(std)
                else:
                    grads[base + 2 * l] = delta.t() @ (obs if l == 0 else torch.nn.functional.silu(pre[l - 1]))
                grads[base + 2 * l + 1] = dsum
                if l > 0:
                    delta = torch.ops.aten.silu_backward(delta @ ws[l], pre[l - 1])      # (delta W_l) * silu'(z_{l-1}), one kernel

        if g_pol is not None:
            net_backward(params[0:2 * n_pol:2], ppre, g_pol.contiguous(), 0)
        if g_val is not None:
            net_backward(params[2 * n_pol::2], vpre, g_val.reshape(-1, 1).contiguous(), 2 * n_pol)
        return (None, None, None, None, *grads)


def actor_critic(obs: torch.Tensor, mean: Optional[torch.Tensor], std: Optional[torch.Tensor], policy_net, value_net):
    """(policy logits [M, P], values [M]) of the RAW observations `obs` [M, K] (normalised inside the kernel)."""
    pw, pb = net_params(policy_net)
    vw, vb = net_params(value_net)
    params = [t for wb in zip(pw, pb) for t in wb] + [t for wb in zip(vw, vb) for t in wb]
    return _ActorCritic.apply(obs.contiguous(), mean, std, len(pw), *params)


@torch.no_grad()
def policy_logits(obs: torch.Tensor, mean: Optional[torch.Tensor], std: Optional[torch.Tensor], policy_net) -> torch.Tensor:
    """Inference: normalise + policy MLP in one launch (the rollout's actor step)."""
    pw, pb = net_params(policy_net)
    return hip.mlp_forward(obs.contiguous(), mean, std, policy=(pw, pb))[0]
