"""How much of a launch's tail would a multi-step (unsynchronised) rollout average away?  Per-env cycle totals of the instrumented
step kernel over consecutive env steps under random actions: a synchronised step lasts max_env(cycles[t]); T unsynchronised steps
last max_env(sum_t cycles[t]).  (Totals include the SIMD partner's interference; a diagnostic, never timed.)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'brax-rodent-run_amd'))
import numpy as np, torch
from rodent_amd import envs
from rodent_amd.envs import wrappers
from tests import util
dev = torch.device('cuda:0'); N = 2048; STEPS = 60
env = envs.get_environment('rodent', track_pos=util.synthetic_track(), num_envs=N, xml_path='rodent_optimized.xml', iterations=8, ls_iterations=8, device=dev)
wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
from rodent_amd import jax_random
state = wenv.reset(jax_random.split(jax_random.PRNGKey(0), N))
for _ in range(40):
    state = wenv.step(state, torch.rand(N, 30, device=dev) * 2 - 1)
buf = torch.zeros(N, 24, dtype=torch.int64, device=dev)
env._batch.set_profile(buf)
tot = []
for t in range(STEPS):
    buf.zero_()
    state = wenv.step(state, torch.rand(N, 30, device=dev) * 2 - 1)
    torch.cuda.synchronize()
    tot.append(buf.sum(1).cpu().numpy().astype(np.float64))
c = np.stack(tot)                      # [STEPS, N]
print("per-step max over envs: mean %.0f; per-step mean over envs: %.0f; ratio %.3f" % (c.max(1).mean(), c.mean(), c.max(1).mean() / c.mean()))
for T in (1, 2, 5, 10, 20, 30):
    k = STEPS // T
    fused = np.array([c[i * T:(i + 1) * T].sum(0).max() for i in range(k)]) / T
    print(f"T={T:3d}: max over envs of the T-step mean = {fused.mean():.0f} cycles per step  ({100 * (fused.mean() / c.max(1).mean() - 1):+.1f} % vs synchronised steps)")
x = c - c.mean(1, keepdims=True)
print("lag-1 autocorrelation of an env's cost:", float((x[1:] * x[:-1]).mean() / (x * x).mean()))
