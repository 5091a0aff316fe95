"""GPU: the fused PPO loss + output-gradient kernels (C ABI `rr_ppo_loss`) and the autograd-free minibatch update built on
them, against `losses.compute_ppo_loss` + autograd in float32 (the path they replace) and in float64 (the yardstick)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(entropy_cost=1e-3, discounting=0.97, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.3)


def _fixed_noise_dist(A, noise):
    from rodent_amd.training.networks import NormalTanhDistribution

    class FixedNoise(NormalTanhDistribution):
        def sample_no_postprocessing(self, logits, generator=None):
            loc, scale = self._params(logits)
            return loc + scale * noise.to(loc.dtype).to(loc.device).reshape(loc.shape)
    return FixedNoise(A)


def _batch(T, B, R, A, seed):
    g = torch.Generator().manual_seed(seed)
    data = dict(raw_action=torch.randn(R, T, A, generator=g) * 0.8, log_prob=torch.randn(R, T, generator=g) * 2 - 25,
                reward=torch.rand(R, T, generator=g), truncation=(torch.rand(R, T, generator=g) < 0.05).float())
    data["discount"] = 1 - (torch.rand(R, T, generator=g) < 0.1).float()
    logits = torch.randn((T + 1) * B, 2 * A, generator=g) * 0.7
    values = torch.randn((T + 1) * B, generator=g) * 3
    noise = torch.randn(T * B, A, generator=g)
    idx = torch.randperm(R, generator=g)[:B]
    return data, logits, values, noise, idx


def _reference(data, logits, values, noise, idx, T, B, A, dtype, device, normalize_advantage=True):
    """compute_ppo_loss + autograd in `dtype`; returns (metrics, d loss / d logits, d loss / d values)."""
    from rodent_amd.training.agents.ppo import losses
    c = lambda x: x.to(dtype).to(device)
    lg = c(logits).clone().requires_grad_(True)
    vl = c(values).clone().requires_grad_(True)
    rows = idx if idx is not None else torch.arange(B)
    mbd = {k: c(data[k][rows]).transpose(0, 1) for k in ("raw_action", "log_prob", "reward", "discount", "truncation")}
    v = vl.reshape(T + 1, B)
    loss, m = losses.compute_ppo_loss(lg[:T * B].reshape(T, B, 2 * A), v[:T], v[T], mbd, _fixed_noise_dist(A, noise),
                                      normalize_advantage=normalize_advantage, **CFG)
    loss.backward()
    return torch.stack([m[k] for k in ("total_loss", "policy_loss", "v_loss", "entropy_loss")]).double().cpu(), \
        lg.grad.double().cpu(), vl.grad.double().cpu()


@pytest.mark.parametrize("T,B,R,A,use_idx,norm", [(10, 256, 700, 30, True, True), (10, 2048, 2048, 30, True, True), (5, 77, 77, 30, False, True),
                                                  (7, 130, 200, 45, True, False), (3, 8, 8, 2, False, True)])
def test_fused_loss_and_output_gradients(T, B, R, A, use_idx, norm):
    from rodent_amd import hip
    data, logits, values, noise, idx = _batch(T, B, R, A, seed=T * 1000 + B)
    if not use_idx:
        idx = None
    m64, gl64, gv64 = _reference(data, logits, values, noise, idx, T, B, A, torch.float64, "cpu", norm)
    m32, gl32, gv32 = _reference(data, logits, values, noise, idx, T, B, A, torch.float32, DEV, norm)
    dd = {k: v.to(DEV).contiguous() for k, v in data.items()}
    gl, gv, m = hip.ppo_loss(logits.to(DEV), values.to(DEV), dd, idx.to(DEV) if idx is not None else None, noise.to(DEV), T,
                             normalize_advantage=norm, **CFG)
    torch.cuda.synchronize()
    gl, gv, m = gl.double().cpu(), gv.double().cpu(), m.double().cpu()
    assert torch.isfinite(gl).all() and torch.isfinite(gv).all()
    assert (gl[T * B:] == 0).all() and (gv[T * B:] == 0).all()                 # bootstrap rows
    for name, got, t32, want in (("logits", gl, gl32, gl64), ("values", gv, gv32, gv64), ("metrics", m, m32, m64)):
        scale = want.abs().max()
        err, err32 = (got - want).abs().max() / scale, (t32 - want).abs().max() / scale
        print(f"T={T} B={B} A={A} {name}: fused {err:.2e}  torch-f32 {err32:.2e}  (relative to max |.|)")
        assert err <= 3 * err32 + 2e-6, (name, float(err), float(err32))        # in the float32 class of the path it replaces
    # every sample is either inside the clip range (tie of the two surrogates) or outside: both branches are exercised
    assert (gl64[:T * B].abs().sum(1) > 0).all()


def test_fused_update_equals_the_autograd_path():
    """FusedUpdate fills the flat gradient buffer with what compute_ppo_loss + backward (on the same fused forward) produce."""
    from rodent_amd.training import distributed as D, fused_mlp, networks
    from rodent_amd.training.agents.ppo import fused_update, losses
    torch.manual_seed(0)
    T, B, R, K, A = 6, 96, 300, 211, 30
    nets = networks.make_ppo_networks(K, A, device=DEV)
    pnet, vnet, dist = nets.policy_network, nets.value_network, nets.parametric_action_distribution
    params = list(pnet.parameters()) + list(vnet.parameters())
    flat = D.FlatGrads(params)
    g = torch.Generator(device=DEV).manual_seed(1)
    data, _, _, _, idx = _batch(T, B, R, A, seed=9)
    data = {k: v.to(DEV).contiguous() for k, v in data.items()}
    data["obs"] = torch.randn(R, T + 1, K, device=DEV, generator=g) * 2 + 0.5
    idx = idx.to(DEV)
    mean, std = torch.randn(K, device=DEV, generator=g) * 0.3, torch.rand(K, device=DEV, generator=g) + 0.5
    fu = fused_update.FusedUpdate(pnet, vnet, dist, T, normalize_advantage=True, **CFG)
    gen = torch.Generator(device=DEV).manual_seed(77)
    m_f = fu(data, idx, mean, std, gen)
    got = flat.flat.clone()
    m_f = {k: float(v) for k, v in m_f.items()}
    # the path it replaces, same noise stream
    gen = torch.Generator(device=DEV).manual_seed(77)
    mbd = {k: data[k][idx].transpose(0, 1) for k in ("raw_action", "log_prob", "reward", "discount", "truncation")}
    raw = data["obs"][idx].transpose(0, 1)
    logits_all, values_all = fused_mlp.actor_critic(raw.reshape((T + 1) * B, -1), mean, std, pnet, vnet)
    values = values_all.reshape(T + 1, B)
    loss, m = losses.compute_ppo_loss(logits_all[:T * B].reshape(T, B, -1), values[:T], values[T], mbd, dist, normalize_advantage=True,
                                      generator=gen, **CFG)
    flat.zero_()
    loss.backward()
    want = flat.flat.clone()
    o = 0
    for p in params:                                           # per tensor: the scales differ by orders of magnitude
        a, b = got[o:o + p.numel()], want[o:o + p.numel()]
        o += p.numel()
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 1e-9, (tuple(p.shape), float((a - b).abs().max()), float(b.abs().max()))
    for k in m_f:
        assert abs(m_f[k] - float(m[k])) <= 1e-5 * max(1.0, abs(float(m[k]))), (k, m_f[k], float(m[k]))


@pytest.mark.parametrize("M,H", [(22528, 256), (20480, 32), (100, 64), (7, 256), (513, 128)])
def test_silu_backward_elementwise_kernel(M, H):
    from rodent_amd import hip
    g = torch.Generator().manual_seed(M + H)
    G, z = torch.randn(M, H, generator=g), torch.randn(M, H, generator=g) * 3
    s = torch.sigmoid(z.double())
    want_d = G.double() * s * (1 + z.double() * (1 - s))
    want_h, want_b = z.double() * s, want_d.sum(0)
    Gd, zd, bg = G.to(DEV), z.to(DEV), torch.empty(H, device=DEV)
    d, h = hip.mlp_silu_backward(Gd, zd, bg)
    torch.cuda.synchronize()
    assert d.data_ptr() == Gd.data_ptr() and h.data_ptr() == zd.data_ptr()                  # in place
    assert (d.double().cpu() - want_d).abs().max() <= 2e-6 * want_d.abs().max()
    assert (h.double().cpu() - want_h).abs().max() <= 2e-6 * want_h.abs().max()
    ref32 = (G * torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z)))).sum(0).double()       # a float32 column sum's error as the yardstick
    err, err32 = (bg.double().cpu() - want_b).abs().max(), (ref32 - want_b).abs().max()
    assert err <= 3 * err32 + 1e-6 * want_b.abs().max(), (float(err), float(err32))


@pytest.mark.parametrize("M", [22528, 2048, 45])
def test_value_backward_chain_kernel(M):
    """rr_mlp_value_backward against float64: delta_j, h_j = silu(z_j), db_j of every hidden layer; the float32 composition of
    the same formulas (library products) is the yardstick."""
    from rodent_amd import hip
    nh, H = 5, 256
    g = torch.Generator().manual_seed(M)
    z = torch.randn(nh, M, H, generator=g) * 1.5
    Ws = [None] + [torch.randn(H, H, generator=g) / 16 for _ in range(1, nh)]
    wh = torch.randn(1, H, generator=g) / 16
    gv = torch.randn(M, generator=g)

    def chain(dt, dev):
        c = lambda x: x.to(dt).to(dev)
        zz = c(z)
        s = torch.sigmoid(zz)
        sp, hh = s * (1 + zz * (1 - s)), zz * s
        d = [None] * nh
        d[nh - 1] = c(gv)[:, None] * c(wh) * sp[nh - 1]
        for j in range(nh - 1, 0, -1):
            d[j - 1] = (d[j] @ c(Ws[j])) * sp[j - 1]
        return torch.stack(d).double().cpu(), hh.double().cpu()
    d64, h64 = chain(torch.float64, "cpu")
    d32, _ = chain(torch.float32, DEV)
    pre = z.to(DEV).contiguous()
    bgs = [torch.empty(H, device=DEV) for _ in range(nh)]
    wt = [None] + [Ws[j].to(DEV).t().contiguous() for j in range(1, nh)]
    delta, h = hip.mlp_value_backward(gv.to(DEV), wh.to(DEV), wt, pre, bgs)
    torch.cuda.synchronize()
    assert h.data_ptr() == pre.data_ptr()
    assert (h.double().cpu() - h64).abs().max() <= 2e-6 * h64.abs().max()
    for j in range(nh):
        scale = d64[j].abs().max()
        err, err32 = (delta[j].double().cpu() - d64[j]).abs().max() / scale, (d32[j] - d64[j]).abs().max() / scale
        print(f"M={M} layer {j}: mfma chain {err:.2e}  torch-f32 {err32:.2e}")
        assert err <= 3 * err32 + 2e-6, (j, float(err), float(err32))
        want_b = d64[j].sum(0)
        assert (bgs[j].double().cpu() - want_b).abs().max() <= 3e-5 * d64[j].abs().sum(0).max(), j


@pytest.mark.parametrize("M,O,I,rows,norm", [(22528, 256, 256, False, False), (22528, 256, 1263, True, True), (20480, 32, 32, False, False),
                                             (20480, 60, 32, False, False), (20480, 32, 1263, True, True), (22528, 1, 256, False, False),
                                             (45, 256, 256, False, False), (1000, 32, 77, True, False), (131, 60, 32, False, True)])
def test_weight_grad_kernel(M, O, I, rows, norm):
    from rodent_amd import hip
    g = torch.Generator().manual_seed(M + O + I)
    R = M + 37 if rows else M
    a, b = torch.randn(M, O, generator=g), torch.randn(R, I, generator=g) * 2 + 0.3
    ridx = torch.randperm(R, generator=g)[:M] if rows else None
    mean, std = (torch.randn(I, generator=g) * 0.3, torch.rand(I, generator=g) + 0.5) if norm else (None, None)
    x = b[ridx] if rows else b
    if norm:
        x = (x - mean) / std
    want = a.double().t() @ x.double()
    ref32 = (a.to(DEV).t() @ x.to(DEV)).double().cpu()
    out = torch.empty(O, I, device=DEV)
    hip.mlp_weight_grad(a.to(DEV), b.to(DEV), out, rows=ridx.to(DEV) if rows else None, mean=mean.to(DEV) if norm else None,
                        std=std.to(DEV) if norm else None, delta_colsum=a.to(DEV).sum(0) if norm else None)
    torch.cuda.synchronize()
    scale = want.abs().max()
    err, err32 = (out.double().cpu() - want).abs().max() / scale, (ref32 - want).abs().max() / scale
    print(f"M={M} O={O} I={I}: mfma split-row {err:.2e}  library f32 {err32:.2e}")
    assert err <= 3 * err32 + 2e-6, (float(err), float(err32))


def test_forward_reads_the_minibatch_in_place():
    """rr_mlp_forward with obs_rows equals the forward on the gathered copy, bit for bit."""
    from rodent_amd import hip
    from rodent_amd.training import fused_mlp, networks
    torch.manual_seed(1)
    K, A, R, M = 300, 30, 5000, 1000
    nets = networks.make_ppo_networks(K, A, device=DEV)
    obs = torch.randn(R, K, device=DEV)
    rows = torch.randperm(R, device=DEV)[:M]
    mean, std = torch.randn(K, device=DEV) * 0.1, torch.rand(K, device=DEV) + 0.5
    pp, vp = fused_mlp.net_params(nets.policy_network), fused_mlp.net_params(nets.value_network)
    a = hip.mlp_forward(obs, mean, std, pp, vp, want_pre=True, rows=rows)
    b = hip.mlp_forward(obs[rows].contiguous(), mean, std, pp, vp, want_pre=True)
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)


def test_policy_sample_kernel_matches_the_distribution():
    from rodent_amd import hip
    from rodent_amd.training.networks import NormalTanhDistribution
    g = torch.Generator().manual_seed(4)
    N, A = 2048, 30
    logits, eps = torch.randn(N, 2 * A, generator=g) * 0.8, torch.randn(N, A, generator=g)
    dist = NormalTanhDistribution(A)
    loc, scale = dist._params(logits.double())
    raw64 = loc + scale * eps.double()
    act64, lp64 = torch.tanh(raw64), dist.log_prob(logits.double(), raw64)
    loc32, scale32 = dist._params(logits)
    raw32 = loc32 + scale32 * eps
    lp32 = dist.log_prob(logits, raw32).double()
    act, raw, lp = hip.policy_sample(logits.to(DEV), eps.to(DEV), dist.min_std)
    torch.cuda.synchronize()
    assert (raw.double().cpu() - raw64).abs().max() <= 2e-6 * raw64.abs().max()
    assert (act.double().cpu() - act64).abs().max() <= 2e-6
    err, err32 = (lp.double().cpu() - lp64).abs().max(), (lp32 - lp64).abs().max()
    print(f"log_prob: kernel {err:.2e}  composed float32 {err32:.2e}  (|log_prob| up to {lp64.abs().max():.1f})")
    assert err <= 3 * err32 + 1e-5


@pytest.mark.parametrize("M,extra,P", [(20480, 2048, 60), (333, 0, 60), (50, 7, 2)])
def test_policy_backward_chain_kernel(M, extra, P):
    """rr_policy_backward against float64: delta_j, h_j = silu(z_j), db_j of the 32-wide stack; rows behind the first M of each
    layer (the bootstrap rows of a minibatch) are left alone."""
    from rodent_amd import hip
    nh, H = 4, 32
    g = torch.Generator().manual_seed(M + P)
    z = torch.randn(nh, M + extra, H, generator=g) * 1.5
    Ws = [None] + [torch.randn(H, H, generator=g) / 5 for _ in range(1, nh)]
    wh = torch.randn(P, H, generator=g) / 5
    gl = torch.randn(M, P, generator=g)

    def chain(dt, dev):
        c = lambda x: x.to(dt).to(dev)
        zz = c(z[:, :M])
        s = torch.sigmoid(zz)
        sp, hh = s * (1 + zz * (1 - s)), zz * s
        d = [None] * nh
        d[nh - 1] = (c(gl) @ c(wh)) * sp[nh - 1]
        for j in range(nh - 1, 0, -1):
            d[j - 1] = (d[j] @ c(Ws[j])) * sp[j - 1]
        return torch.stack(d).double().cpu(), hh.double().cpu()
    d64, h64 = chain(torch.float64, "cpu")
    d32, _ = chain(torch.float32, DEV)
    pre = z.to(DEV).contiguous()
    bgs = [torch.empty(H, device=DEV) for _ in range(nh)]
    delta, h = hip.policy_backward(gl.to(DEV), wh.to(DEV), [None] + [Ws[j].to(DEV) for j in range(1, nh)], pre, bgs)
    torch.cuda.synchronize()
    assert (h[:, :M].double().cpu() - h64).abs().max() <= 2e-6 * h64.abs().max()
    assert torch.equal(h[:, M:].cpu(), z[:, M:])                                   # untouched
    for j in range(nh):
        scale = d64[j].abs().max()
        err, err32 = (delta[j].double().cpu() - d64[j]).abs().max() / scale, (d32[j] - d64[j]).abs().max() / scale
        print(f"M={M} layer {j}: policy chain {err:.2e}  torch-f32 {err32:.2e}")
        assert err <= 3 * err32 + 2e-6, (j, float(err), float(err32))
        assert (bgs[j].double().cpu() - d64[j].sum(0)).abs().max() <= 3e-5 * d64[j].abs().sum(0).max(), j


@pytest.mark.parametrize("M,K,use_rows", [(2048, 1263, False), (1024, 1263, True), (100, 333, False), (1, 1263, False)])
def test_policy_act_two_launches(M, K, use_rows):
    """rr_policy_act (first layer split over k + tail kernel with the tanh-normal head) against the float64 policy; the float32
    nn.Linear policy is the yardstick.  Stochastic and deterministic modes, optional row indirection."""
    import copy
    from rodent_amd import hip
    from rodent_amd.training import fused_mlp, networks
    torch.manual_seed(M + K)
    A = 30
    nets = networks.make_ppo_networks(K, A, device=DEV)
    net, dist = nets.policy_network, nets.parametric_action_distribution
    for l in net.layers:
        l.bias.data.uniform_(-0.2, 0.2)
    R = M + 50 if use_rows else M
    obs = torch.randn(R, K, device=DEV) * 2 + 0.3
    rows = torch.randperm(R, device=DEV)[:M] if use_rows else None
    mean, std = torch.randn(K, device=DEV) * 0.3, torch.rand(K, device=DEV) + 0.5
    eps = torch.randn(M, A, device=DEV)
    x = obs[rows] if use_rows else obs
    net64 = copy.deepcopy(net).double()
    with torch.no_grad():
        lg64 = net64((x.double() - mean.double()) / std.double())
        lg32 = net((x - mean) / std).double()
    loc, scale = dist._params(lg64)
    raw64 = loc + scale * eps.double()
    lp64 = dist.log_prob(lg64, raw64)
    act, raw, lp, lg = hip.policy_act(obs, mean, std, fused_mlp.net_params(net), eps, dist.min_std, want_logits=True, rows=rows)
    torch.cuda.synchronize()
    scale_l = lg64.abs().max()
    err, err32 = (lg.double() - lg64).abs().max() / scale_l, (lg32 - lg64).abs().max() / scale_l
    print(f"M={M} K={K}: logits two-launch {float(err):.2e}  nn.Linear f32 {float(err32):.2e}")
    assert err <= 3 * err32 + 2e-6
    tol = 20 * float(err.clamp_min(1e-7)) * float(scale_l)                      # what the logits' error can do to the head's outputs
    assert (raw.double() - raw64).abs().max() <= tol + 1e-5
    assert (act.double() - torch.tanh(raw64)).abs().max() <= tol + 1e-5
    assert (lp.double() - lp64).abs().max() <= 50 * tol + 1e-4
    act_d, raw_d, lp_d, _ = hip.policy_act(obs, mean, std, fused_mlp.net_params(net), None, dist.min_std, rows=rows)
    assert raw_d is None and lp_d is None
    assert (act_d.double() - torch.tanh(loc)).abs().max() <= tol + 1e-5


def test_weight_grad_batch_against_single_calls():
    """rr_mlp_weight_grad_batch (products of one tile shape share a launch, one reduction).  Round 3: the row range of a product is cut
    into slices by what the LAUNCH holds (a batch of five 128 x 128-tile products needs fewer slices each than one product alone), so a
    batched product and the same product alone add their partial tiles in different groupings: equal to float32 summation round-off
    (checked against float64), no longer bit for bit.  What stays bit for bit: a repeated batch (fixed slice order, no atomics), and a
    single product against the batch of one it is."""
    from rodent_amd import hip
    g = torch.Generator(device=DEV).manual_seed(2)
    M, K = 4096, 1263
    obs = torch.randn(M + 9, K, device=DEV, generator=g)
    rows = torch.randperm(M + 9, device=DEV, generator=g)[:M]
    mean, std = torch.randn(K, device=DEV, generator=g) * 0.2, torch.rand(K, device=DEV, generator=g) + 0.5
    mk = lambda o: torch.randn(M, o, device=DEV, generator=g)
    specs = [(mk(60), mk(32), {}), (mk(32), mk(32), {}), (mk(32), mk(32), {}), (mk(32), obs, dict(rows=rows, mean=mean, std=std)),
             (mk(1), mk(256), {}), (mk(256), mk(256), {}), (mk(256), mk(256), {}), (mk(256), obs, dict(rows=rows, mean=mean, std=std))]
    single, items, again = [], [], []
    for d, a, kw in specs:
        if kw:
            kw = dict(kw, delta_colsum=d.sum(0))
        out1, out2, out3 = (torch.empty(d.shape[1], a.shape[1], device=DEV) for _ in range(3))
        hip.mlp_weight_grad(d, a, out1, **kw)
        single.append(out1)
        items.append(dict(delta=d, act=a, out=out2, **kw))
        again.append(dict(delta=d, act=a, out=out3, **kw))
    hip.mlp_weight_grad_batch(items)
    hip.mlp_weight_grad_batch(again)
    one = torch.empty_like(single[5])
    hip.mlp_weight_grad_batch([dict(delta=specs[5][0], act=specs[5][1], out=one)])
    torch.cuda.synchronize()
    assert torch.equal(one, single[5])
    for (d, a, kw), s1, it, ag in zip(specs, single, items, again):
        assert torch.isfinite(s1).all() and torch.equal(it["out"], ag["out"])
        x = a.double() if not kw else (a[kw["rows"]].double() - kw["mean"].double()) / kw["std"].double()
        want = d.double().t() @ x
        scale = float(want.abs().max())
        e_single, e_batch = float((s1.double() - want).abs().max()), float((it["out"].double() - want).abs().max())
        assert e_batch <= 2 * e_single + 2e-6 * scale and e_single <= 2 * e_batch + 2e-6 * scale, (e_single, e_batch, scale)


def test_obs_moments_kernel_against_numpy():
    """`rr_obs_moments` (the normaliser update's sums in one pass over the unroll buffer) against float64 numpy: rows
    t < T of every sequence only, odd widths, more rows than one block; and the resulting update against the tensor-expression form."""
    from rodent_amd import hip
    from rodent_amd.training import running_statistics
    rng = np.random.default_rng(0)
    for shape, T in (((3, 50, 11, 37), 10), ((2, 700, 6, 1263), 5), ((1, 9, 4, 5), 4)):
        x = rng.normal(0.7, 1.9, size=shape).astype(np.float32)
        mean = rng.normal(0.5, 0.3, size=shape[-1]).astype(np.float32)
        got = hip.obs_moments(torch.tensor(x, device=DEV), T, torch.tensor(mean, device=DEV)).cpu().numpy()
        d = x[..., :T, :].astype(np.float64).reshape(-1, shape[-1]) - mean.astype(np.float64)
        np.testing.assert_allclose(got[0], d.sum(0), rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got[1], (d * d).sum(0), rtol=1e-12, atol=1e-9)
    st = running_statistics.init_state(37, DEV)
    ref = running_statistics.init_state(37, DEV)
    for i in range(3):
        buf = torch.tensor(rng.normal(1.0 + i, 2.0, size=(4, 64, 11, 37)).astype(np.float32), device=DEV)
        st = running_statistics.update_from_unroll_buffer(st, buf, 10)
        ref = running_statistics.update(ref, buf[:, :, :10])
        for f in ("count", "mean", "summed_variance", "std"):
            torch.testing.assert_close(getattr(st, f), getattr(ref, f), rtol=3e-5, atol=1e-5)
