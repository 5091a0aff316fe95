"""forward phase stamps (RR_MLP_PROF) under different observation access patterns: contiguous rows, rows scattered over a large buffer
(the learner's minibatch), all rows from a 64-row L2-resident set; and without the pre-activation dumps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import torch
from rodent_amd import hip
from rodent_amd.training import fused_mlp, networks
dev = "cuda:0"
K, M = 1263, 22528
torch.manual_seed(0)
n = networks.make_ppo_networks(K, 30, device=dev)
pp, vp = fused_mlp.net_params(n.policy_network), fused_mlp.net_params(n.value_network)
mean, std = torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5
big = torch.randn(400000, K, device=dev)            # 2 GB
cases = {
    "contiguous": (big[:M], None, True),
    "scattered over 2 GB": (big, torch.randperm(400000, device=dev)[:M], True),
    "64-row set (L2 resident)": (big, torch.arange(M, device=dev) % 64, True),
    "contiguous, no dumps": (big[:M], None, False),
}
with torch.no_grad():
    for name, (obs, rows, pre) in cases.items():
        for _ in range(3):
            hip.mlp_forward(obs, mean, std, pp, vp, want_pre=pre, rows=rows)
        torch.cuda.synchronize()
        print("CASE", name, flush=True)
        path = os.environ.get("RR_MLP_PROF")
        if path and os.path.exists(path):
            print(open(path).read().splitlines()[-1][40:], flush=True)
with torch.no_grad():
    for name, (p_, v_) in {"value net only": (None, vp), "policy net only": (pp, None), "value only, no normaliser": (None, vp)}.items():
        mu, sd = (None, None) if "no normaliser" in name else (mean, std)
        for _ in range(3):
            hip.mlp_forward(big[:M], mu, sd, p_, v_, want_pre=True)
        torch.cuda.synchronize()
        print("CASE", name, flush=True)
        path = os.environ.get("RR_MLP_PROF")
        if path and os.path.exists(path):
            print(open(path).read().splitlines()[-1][40:], flush=True)
