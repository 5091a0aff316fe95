"""BASELINE config 5: rodent_pair.xml (two replicated rodents, nv 146, 114 contacts), physics-only random-action
stepping, N = 4096 envs on 1 GPU.  Prints env-steps/s of pipeline_step (n_frames 10)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import numpy as np, torch
from rodent_amd import assets, hip, mjcf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
path = assets.asset_path("rodent_pair")
m = mjcf.load_blob(path)
b = hip.Batch(hip.Model(path, 8, 8), N, dev)
d = b.dims
rng = np.random.default_rng(0)
q = np.tile(m["qpos0"], (N, 1)).astype(np.float32) + rng.uniform(-0.01, 0.01, (N, d.nq)).astype(np.float32)
st = dict(qpos=torch.tensor(q, device=dev), qvel=torch.zeros(N, d.nv, device=dev), act=torch.zeros(N, d.na, device=dev),
          qacc_warmstart=torch.zeros(N, d.nv, device=dev))
b.pipeline_init(st)
for _ in range(10):
    b.pipeline_step(st, torch.rand(N, d.nu, device=dev) * 2 - 1, 10)
b.set_timing(True)
K = 20
for _ in range(K):
    b.pipeline_step(st, torch.rand(N, d.nu, device=dev) * 2 - 1, 10)
torch.cuda.synchronize()
ms, n = b.kernel_time()
print(json.dumps({"config": "rodent_pair.xml physics-only, random actions, CG 8/8, n_frames 10", "num_envs": N,
                  "ms_per_step": ms / n, "env_steps_per_s": N / (ms / n) * 1e3, "lds_bytes_per_env": d.lds_bytes,
                  "finite": bool(torch.isfinite(st["qpos"]).all()), "mean_z": float(st["qpos"][:, 2].mean())}))
