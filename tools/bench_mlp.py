"""Time the fused f32-MFMA actor/critic forward (rr_mlp_forward) against the nn.Linear (hipBLASLt) path at the rollout
([2048 x 1263], policy only) and learner ([22528 x 1263], both nets) shapes.  Prints JSON lines with achieved TFLOP/s against
the 157 TF dense f32-MFMA peak (MI355X_MICROARCH.md)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import torch
from rodent_amd import hip
from rodent_amd.training import networks, running_statistics

dev = "cuda:0"
K = 1263
n = networks.make_ppo_networks(K, 30, device=dev)
wb = lambda net: ([l.weight.detach() for l in net.layers], [l.bias.detach() for l in net.layers])
mean, std = torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5
PEAK = 157.3


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


for M, both in ((2048, False), (22528, True), (22528, False)):
    obs = torch.randn(M, K, device=dev)
    flop_p = 2 * (K * 32 + 3 * 32 * 32 + 32 * 60) * M
    flop_v = 2 * (K * 256 + 4 * 256 * 256 + 256) * M
    flop = flop_p + (flop_v if both else 0)
    pol, val = wb(n.policy_network), wb(n.value_network)
    with torch.no_grad():
        t_f = timeit(lambda: hip.mlp_forward(obs, mean, std, policy=pol, value=val if both else None))
        t_fp = timeit(lambda: hip.mlp_forward(obs, mean, std, policy=pol, value=val if both else None, want_pre=True))

        def torch_path():
            x = (obs - mean) / std
            p = n.policy_network(x)
            return (p, n.value_network(x)) if both else p
        t_t = timeit(torch_path)
    print(json.dumps({"M": M, "nets": "policy+value" if both else "policy", "gflop": flop / 1e9, "fused_ms": t_f, "fused_with_pre_dump_ms": t_fp,
                      "torch_ms": t_t, "fused_tflops": flop / t_f / 1e9, "torch_tflops": flop / t_t / 1e9,
                      "fused_frac_of_f32_mfma_peak": flop / t_f / 1e9 / PEAK, "speedup": t_t / t_f}), flush=True)
