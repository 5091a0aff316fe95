"""the learner-shape forward only ([22528 x 1263], both nets, with dumps), a few calls: the target of tools/pmc_forward.sh"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import torch
from rodent_amd import hip
from rodent_amd.training import fused_mlp, networks
dev = "cuda:0"
K, M = int(os.environ.get("RR_K", 1263)), 22528
torch.manual_seed(0)
n = networks.make_ppo_networks(K, 30, device=dev)
pp, vp = fused_mlp.net_params(n.policy_network), fused_mlp.net_params(n.value_network)
mean, std = torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5
obs = torch.randn(M, K, device=dev)
with torch.no_grad():
    for _ in range(6):
        hip.mlp_forward(obs, mean, std, pp, vp, want_pre=True)
torch.cuda.synchronize()
