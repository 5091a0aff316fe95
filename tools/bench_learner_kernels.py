"""Time the learner's hand-written kernels at the launcher's minibatch shape ([22528 x 1263] observations, value 256 x5,
policy 32 x4): forward with pre-activation dumps, value delta chain, weight gradients.  RR_LIB selects the library build.
Prints one JSON line: ms and TFLOP/s against the 157 TF dense f32-MFMA peak."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import torch
from rodent_amd import hip
from rodent_amd.training import fused_mlp, networks

dev = "cuda:0"
K, M, Mp = 1263, 22528, 20480
torch.manual_seed(0)
n = networks.make_ppo_networks(K, 30, device=dev)
pp, vp = fused_mlp.net_params(n.policy_network), fused_mlp.net_params(n.value_network)
mean, std = torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5
obs = torch.randn(M, K, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {"lib": os.path.basename(hip.LIB_PATH)}
with torch.no_grad():
    t = timeit(lambda: hip.mlp_forward(obs, mean, std, pp, vp, want_pre=True))
    fl = 2 * M * (K * 288 + 4 * 256 * 256 + 256 + 3 * 32 * 32 + 32 * 60)
    out["forward_ms"], out["forward_tflops"] = t, fl / t / 1e9
    _, _, ppre, vpre = hip.mlp_forward(obs, mean, std, pp, vp, want_pre=True)
    g = torch.randn(M, device=dev)
    layers = n.value_network.layers
    wt = [None] + [layers[j].weight.t().contiguous() for j in range(1, 5)]
    bgs = [torch.empty(256, device=dev) for _ in range(5)]
    bufs = {}
    z = vpre.clone()
    t = timeit(lambda: hip.mlp_value_backward(g, layers[5].weight, wt, z, bgs, bufs))
    out["value_chain_ms"], out["value_chain_tflops"] = t, 2 * M * 4 * 256 * 256 / t / 1e9
    d = torch.randn(M, 256, device=dev); h = torch.randn(M, 256, device=dev); w = torch.empty(256, 256, device=dev)
    t = timeit(lambda: hip.mlp_weight_grad(d, h, w))
    out["dw_hidden_ms"], out["dw_hidden_tflops"] = t, 2 * M * 256 * 256 / t / 1e9
    w0 = torch.empty(256, K, device=dev); cs = d.sum(0)
    t = timeit(lambda: hip.mlp_weight_grad(d, obs, w0, mean=mean, std=std, delta_colsum=cs))
    out["dw_first_ms"], out["dw_first_tflops"] = t, 2 * M * 256 * K / t / 1e9
    dp = torch.randn(Mp, 32, device=dev); wp = torch.empty(32, K, device=dev); csp = dp.sum(0)
    t = timeit(lambda: hip.mlp_weight_grad(dp, obs, wp, mean=mean, std=std, delta_colsum=csp))
    out["dw_policy_first_ms"], out["dw_policy_first_tflops"] = t, 2 * Mp * 32 * K / t / 1e9
    t = timeit(lambda: hip.mlp_forward(obs[:2048], mean, std, pp))
    out["rollout_policy_forward_ms"] = t
print(json.dumps(out))
with torch.no_grad():
    eps = torch.randn(2048, 30, device=dev)
    t = timeit(lambda: hip.policy_act(obs[:2048], mean, std, pp, eps, 0.001))
    def old():
        lg = hip.mlp_forward(obs[:2048], mean, std, pp)[0]
        return hip.policy_sample(lg, eps, 0.001)
    t0 = timeit(old)
print(json.dumps({"rollout_actor_two_launch_ms": t, "rollout_actor_forward_plus_sample_ms": t0}))
