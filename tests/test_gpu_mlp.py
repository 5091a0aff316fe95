"""GPU: the fused f32-MFMA actor/critic forward (C ABI rr_mlp_forward) against PyTorch references of the same op:
exact integer data (fragment layouts), float32 / float64 torch MLPs at the rollout ([2048 x 1263]) and learner
([22528 x 1263]) shapes, ragged row counts, single-network calls, and the explicit backward against autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(K, P=60, seed=0, pl=4, vl=5):
    from rodent_amd.training import networks
    torch.manual_seed(seed)
    n = networks.make_ppo_networks(K, P // 2, policy_hidden_layer_sizes=(32,) * pl, value_hidden_layer_sizes=(256,) * vl, device=DEV)
    for net in (n.policy_network, n.value_network):          # non-zero biases so the bias path is exercised
        for lin in net.layers:
            torch.nn.init.uniform_(lin.bias, -0.1, 0.1)
    return n


def _wb(net, dtype=None):
    return [l.weight.detach().to(dtype) if dtype else l.weight.detach() for l in net.layers], \
           [l.bias.detach().to(dtype) if dtype else l.bias.detach() for l in net.layers]


def _ref(x, net, dtype):
    x = x.to(dtype)
    ws, bs = _wb(net, dtype)
    for i, (w, b) in enumerate(zip(ws, bs)):
        x = x @ w.t() + b
        if i < len(ws) - 1:
            x = torch.nn.functional.silu(x)
    return x


def test_integer_data_is_exact():
    """Small-integer observations and first-layer weights (asymmetric): the first-layer pre-activations are exact integers,
    so any lane / row / column mix-up in the MFMA fragment or C/D maps shows as a wrong integer."""
    from rodent_amd import hip
    M, K = 96, 1263
    g = torch.Generator(device=DEV); g.manual_seed(1)
    obs = torch.randint(-3, 4, (M, K), device=DEV, generator=g).float()
    n = _nets(K)
    with torch.no_grad():
        for net in (n.policy_network, n.value_network):
            w = net.layers[0].weight
            w.copy_(torch.randint(-2, 3, w.shape, device=DEV, generator=g).float())
            net.layers[0].bias.copy_(torch.arange(w.shape[0], device=DEV).float() % 7 - 3)
    pol, val, ppre, vpre = hip.mlp_forward(obs, None, None, _wb(n.policy_network), _wb(n.value_network), want_pre=True)
    want_p = obs.double() @ n.policy_network.layers[0].weight.double().t() + n.policy_network.layers[0].bias.double()
    want_v = obs.double() @ n.value_network.layers[0].weight.double().t() + n.value_network.layers[0].bias.double()
    assert torch.equal(ppre[0].double(), want_p) and torch.equal(vpre[0].double(), want_v)


@pytest.mark.parametrize("M", [2048, 22528, 100, 1])
def test_forward_matches_torch(M):
    from rodent_amd import hip
    K = 1263
    g = torch.Generator(device=DEV); g.manual_seed(M)
    obs = torch.randn(M, K, device=DEV, generator=g) * 3 + 0.5
    mean = torch.randn(K, device=DEV, generator=g) * 0.5
    std = torch.rand(K, device=DEV, generator=g) * 2 + 0.1
    n = _nets(K, seed=M)
    pol, val, ppre, vpre = hip.mlp_forward(obs, mean, std, _wb(n.policy_network), _wb(n.value_network), want_pre=True)
    x64 = (obs.double() - mean.double()) / std.double()
    p64, v64 = _ref(x64, n.policy_network, torch.float64), _ref(x64, n.value_network, torch.float64).squeeze(-1)
    x32 = (obs - mean) / std
    p32, v32 = _ref(x32, n.policy_network, torch.float32), _ref(x32, n.value_network, torch.float32).squeeze(-1)
    ep, ev = (pol.double() - p64).abs().max().item(), (val.double() - v64).abs().max().item()
    tp, tv = (p32.double() - p64).abs().max().item(), (v32.double() - v64).abs().max().item()
    print(f"M={M}: |mfma - f64| policy {ep:.2e} value {ev:.2e}; |torch f32 - f64| policy {tp:.2e} value {tv:.2e}")
    assert torch.isfinite(pol).all() and torch.isfinite(val).all()
    # float32 tolerance, stated: as accurate as the float32 library path to within 2x, and 2e-5 absolute (outputs are O(1))
    assert ep <= 2 * tp + 2e-5 and ev <= 2 * tv + 2e-5
    # pre-activations of every hidden layer (what the backward pass consumes)
    h = x64
    for l, lin in enumerate(n.value_network.layers[:-1]):
        z = h @ lin.weight.double().t() + lin.bias.double()
        assert (vpre[l].double() - z).abs().max().item() < 1e-4 * max(1.0, z.abs().max().item())
        h = torch.nn.functional.silu(z)


def test_single_network_calls_and_no_normaliser():
    from rodent_amd import hip
    M, K = 300, 1279                              # rodent_new's observation width
    obs = torch.randn(M, K, device=DEV)
    n = _nets(K, seed=3, pl=2, vl=3)              # other depths
    both = hip.mlp_forward(obs, None, None, _wb(n.policy_network), _wb(n.value_network))
    only_p = hip.mlp_forward(obs, None, None, policy=_wb(n.policy_network))
    only_v = hip.mlp_forward(obs, None, None, value=_wb(n.value_network))
    assert torch.equal(both[0], only_p[0]) and torch.equal(both[1], only_v[1]) and only_p[1] is None and only_v[0] is None
    assert (both[0] - _ref(obs, n.policy_network, torch.float32)).abs().max() < 1e-4
    with pytest.raises(RuntimeError, match="hidden width"):
        from rodent_amd.training import networks
        bad = networks.make_ppo_networks(K, 30, policy_hidden_layer_sizes=(64, 64), device=DEV)
        hip.mlp_forward(obs, None, None, policy=_wb(bad.policy_network))


def test_explicit_backward_matches_autograd():
    """`fused_mlp.actor_critic` (MFMA forward + explicit backward) against the nn.Module path under autograd."""
    from rodent_amd.training import fused_mlp
    M, K = 1024, 1263
    g = torch.Generator(device=DEV); g.manual_seed(5)
    obs = torch.randn(M, K, device=DEV, generator=g)
    mean, std = torch.randn(K, device=DEV, generator=g) * 0.2, torch.rand(K, device=DEV, generator=g) + 0.5
    n = _nets(K, seed=5)
    gp, gv = torch.randn(M, 60, device=DEV, generator=g), torch.randn(M, device=DEV, generator=g)
    params = list(n.policy_network.parameters()) + list(n.value_network.parameters())
    pol, val = fused_mlp.actor_critic(obs, mean, std, n.policy_network, n.value_network)
    ((pol * gp).sum() + (val * gv).sum()).backward()
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    # the same accumulated into pre-allocated .grad buffers (train.py keeps them as views of the flat all-reduce buffer)
    for p in params:
        p.grad = torch.full_like(p, 0.5)
    pol, val = fused_mlp.actor_critic(obs, mean, std, n.policy_network, n.value_network)
    ((pol * gp).sum() + (val * gv).sum()).backward()
    got_direct = [p.grad.clone() - 0.5 for p in params]
    for p in params:
        p.grad = None
    x = (obs - mean) / std
    (((n.policy_network(x)) * gp).sum() + (n.value_network(x).squeeze(-1) * gv).sum()).backward()
    for p, a, b in zip(params, got, got_direct):
        scale = max(p.grad.abs().max().item(), 1e-6)
        assert (a - p.grad).abs().max().item() <= 2e-4 * scale, (p.shape, (a - p.grad).abs().max().item(), scale)
        assert (b - p.grad).abs().max().item() <= 2e-4 * scale + 1e-6, (p.shape, (b - p.grad).abs().max().item(), scale)
