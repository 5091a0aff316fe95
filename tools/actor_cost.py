"""In-kernel actor: what it costs.  The multi-step kernel with the actor inside against the plain multi-step kernel replaying the SAME
actions from the same state (identical physics): kernel time by HIP events."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'brax-rodent-run_amd'))
import torch
from tests import util
from rodent_amd import envs, jax_random
from rodent_amd.envs import wrappers
from rodent_amd.training import acting, networks, running_statistics
dev = torch.device("cuda:0")
N, T = 2048, 10
env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device=dev)
wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
st = wenv.reset(jax_random.split(jax_random.PRNGKey(0), N))
nets = networks.make_ppo_networks(env.observation_size, env.action_size, device=dev)
net, dist = nets.policy_network, nets.parametric_action_distribution
norm = running_statistics.init_state(env.observation_size, dev)
buf = acting.UnrollBuffer(1, N, T, env.observation_size, env.action_size, dev)
actor = acting.actor_params(net, norm, dist.min_std)
traj = dict(obs=buf.obs[0], raw_action=buf.raw_action[0], log_prob=buf.log_prob[0], reward=buf.reward[0], discount=buf.discount[0], truncation=buf.truncation[0])
for _ in range(4):       # settle into contact
    st, _ = wenv.unroll_policy(st, actor, torch.randn(T, N, env.action_size, device=dev), traj)
b = env._batch
res = []
for rep in range(6):
    noise = torch.randn(T, N, env.action_size, device=dev)
    b.set_timing(True)
    st2, actions = wenv.unroll_policy(st, actor, noise, traj)
    torch.cuda.synchronize(); t_actor = b.kernel_time()[0]
    b.set_timing(True)
    st3 = wenv.unroll(st, actions)
    torch.cuda.synchronize(); t_plain = b.kernel_time()[0]
    assert torch.equal(st2.pipeline_state.qpos, st3.pipeline_state.qpos)
    res.append((t_actor, t_plain)); st = st2
ta, tp = sum(r[0] for r in res[1:]) / 5, sum(r[1] for r in res[1:]) / 5
print(f"{T}-step launch of {N} envs: with the actor {ta:.3f} ms, same actions replayed {tp:.3f} ms: actor = {(ta - tp) / T * 1e3:.1f} us per env step ({100 * (ta / tp - 1):.1f} %)")
