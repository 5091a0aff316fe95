"""GPU: env-level behaviour of the HIP path against the oracle (reset / step / obs / reward / cur_frame)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _mk_env(N, model="rodent_optimized"):
    from rodent_amd import envs
    return envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path=f"{model}.xml",
                                iterations=8, ls_iterations=8, device="cuda:0")


def test_n_frames_fused_equals_repeated_launches():
    """10 substeps inside one launch == 10 launches of one substep, bit for bit (no state is lost between frames)."""
    N = 32
    env = _mk_env(N)
    state = env.reset(7)
    ctrl = torch.rand(N, env.action_size, device="cuda:0") * 2 - 1
    ps = state.pipeline_state
    a = dict(qpos=ps.qpos.clone(), qvel=ps.qvel.clone(), act=ps.act.clone(), qacc_warmstart=ps.qacc_warmstart.clone())
    b = {k: v.clone() for k, v in a.items()}
    env._batch.pipeline_step(a, ctrl, 10)
    for _ in range(10):
        env._batch.pipeline_step(b, ctrl, 1)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_step_is_deterministic_across_launches_and_batch_positions():
    """The same state stepped twice, and the same env placed at different batch positions, give bit-identical results:
    one wavefront owns an env, nothing is shared between envs and no reduction order depends on scheduling."""
    N = 2048
    env = _mk_env(N)
    state = env.reset(3)
    g = torch.Generator(device="cuda:0"); g.manual_seed(11)
    for _ in range(3):
        state = env.step(state, torch.rand(N, env.action_size, device="cuda:0", generator=g) * 2 - 1)
    ps = state.pipeline_state
    ctrl = torch.rand(N, env.action_size, device="cuda:0", generator=g) * 2 - 1
    base = dict(qpos=ps.qpos, qvel=ps.qvel, act=ps.act, qacc_warmstart=ps.qacc_warmstart)
    runs = []
    for rep in range(3):
        st = {k: v.clone() for k, v in base.items()}
        env._batch.pipeline_step(st, ctrl, 10)
        runs.append(st)
    perm = torch.randperm(N, device="cuda:0", generator=g)
    st = {k: v[perm].clone() for k, v in base.items()}
    env._batch.pipeline_step(st, ctrl[perm].contiguous(), 10)
    torch.cuda.synchronize()
    for k in base:
        assert torch.isfinite(runs[0][k]).all(), k
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
        assert torch.equal(runs[0][k][perm], st[k]), k


def test_fused_training_wrapper_equals_the_composition():
    """FusedEpisodeAutoResetWrapper (one launch) == AutoResetWrapper(EpisodeWrapper(...)) (tensor ops), bit for bit, through
    episode ends (episode_length 7), terminations and the restored first states."""
    import dataclasses
    from rodent_amd.envs import wrappers as W
    N = 64
    env = _mk_env(N)
    a_env = W.AutoResetWrapper(W.EpisodeWrapper(W.VmapWrapper(env), 7, 1))
    b_env = W.FusedEpisodeAutoResetWrapper(W.VmapWrapper(env), 7)
    assert isinstance(W.wrap(env, episode_length=7, action_repeat=1), W.FusedEpisodeAutoResetWrapper)
    sa, sb = a_env.reset(4), b_env.reset(4)
    g = torch.Generator(device="cuda:0"); g.manual_seed(9)
    seen_done = 0
    for t in range(40):
        act = torch.rand(N, env.action_size, device="cuda:0", generator=g) * 4 - 2      # wild actions: some envs fall over
        sa, sb = a_env.step(sa, act), b_env.step(sb, act)
        seen_done += int(sa.done.sum())
        for f in dataclasses.fields(sa.pipeline_state):
            x, y = getattr(sa.pipeline_state, f.name), getattr(sb.pipeline_state, f.name)
            if torch.is_tensor(x):
                assert torch.equal(x, y), (t, f.name)
        for name in ("obs", "reward", "done"):
            assert torch.equal(getattr(sa, name), getattr(sb, name)), (t, name)
        for k in ("steps", "truncation", "cur_frame"):
            assert torch.equal(sa.info[k], sb.info[k]), (t, k)
        for k in sa.metrics:
            assert torch.equal(sa.metrics[k], sb.metrics[k]), (t, k)
    assert seen_done >= 5 * N          # every env went through several episode ends


def test_env_step_matches_oracle_rodent_new(oracle_built):
    """The env default model (rodent_new.xml: obs 1279, body 1 = the massless `walker`): 30 teacher-forced env steps,
    integers exact, every obs segment / reward / state within the quantile criterion (tests/parity.py).  rodent_optimized
    gets 1000 steps in tests/test_gpu_ladder.py."""
    from tests import parity
    from tests.hip_impl import HipEnvImpl
    N, T = 16, 30
    track = util.synthetic_track()
    track[:, 2] = 0.04                   # rodent_new's free body sits at the origin; pick z explicitly
    seq, A0, tab = parity.rollout_inputs("rodent_new", N, T, (8, 8), seed=41, n_frames=10, z_range=(-1.0, 1.0))
    rng = np.random.default_rng(42)
    seq = [(st, ctrl, rng.integers(0, 260, N).astype(np.int32)) for st, ctrl in seq]
    A = parity.OracleEnvImpl("rodent_new", N, "f64", (8, 8), track)
    out = parity.envstep_ladder(HipEnvImpl(N, (8, 8), track, "rodent_new"), seq, A, parity.OracleEnvImpl("rodent_new", N, "f32", (8, 8), track), tab)
    print(out)
    parity.check_quantiles(out["quantiles"], parity.ENV_FLOORS)


def test_config5_rodent_pair_at_4096_envs():
    """BASELINE config 5 at its size: rodent_pair.xml (nv 146, 114 contacts), 4096 envs, physics only: finite, deterministic
    across launches, and env e of the big batch equals the same state stepped in a batch of 16 (parity of that instance against
    the oracle: tests/test_gpu_ladder.py::test_teacher_forced_substeps_rodent_pair)."""
    from rodent_amd import assets, hip, mjcf
    N = 4096
    path = assets.asset_path("rodent_pair")
    m = mjcf.load_blob(path)
    b = hip.Batch(hip.Model(path, 8, 8), N, torch.device("cuda:0"))
    d = b.dims
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    q = torch.tensor(np.tile(m["qpos0"], (N, 1)), dtype=torch.float32, device="cuda:0") + (torch.rand(N, d.nq, device="cuda:0", generator=g) * 2 - 1) * 0.01
    st = dict(qpos=q, qvel=torch.zeros(N, d.nv, device="cuda:0"), act=torch.zeros(N, d.na, device="cuda:0"),
              qacc_warmstart=torch.zeros(N, d.nv, device="cuda:0"))
    b.pipeline_init(st)
    for _ in range(5):
        b.pipeline_step(st, torch.rand(N, d.nu, device="cuda:0", generator=g) * 2 - 1, 10)
    ctrl = torch.rand(N, d.nu, device="cuda:0", generator=g) * 2 - 1
    s1, s2 = {k: v.clone() for k, v in st.items()}, {k: v.clone() for k, v in st.items()}
    b.pipeline_step(s1, ctrl, 10); b.pipeline_step(s2, ctrl, 10)
    small = hip.Batch(b.model, 16, torch.device("cuda:0"))
    s3 = {k: v[100:116].clone() for k, v in st.items()}
    small.pipeline_step(s3, ctrl[100:116].contiguous(), 10)
    torch.cuda.synchronize()
    for k in st:
        assert torch.isfinite(s1[k]).all(), k
        assert torch.equal(s1[k], s2[k]) and torch.equal(s1[k][100:116], s3[k]), k


def test_simd_pairing_schedule_is_result_neutral():
    """`rr_batch_set_schedule` / `PipelineEnv._rebalance`: the workgroup -> environment map only decides which environments share
    a SIMD.  A balanced env (re-paired every step here) and an unbalanced one produce bit-identical states, observations and
    rewards over 12 steps; an explicit random map likewise; the per-env cycle counts come back non-zero."""
    from rodent_amd import envs
    N = 2048
    mk = lambda **kw: envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path="rodent_optimized.xml",
                                           iterations=8, ls_iterations=8, device="cuda:0", **kw)
    a, b = mk(balance=False), mk(balance=True, rebalance_every=1)
    assert b._balance and not a._balance
    sa, sb = a.reset(5), b.reset(5)
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    for t in range(12):
        act = torch.rand(N, a.action_size, device="cuda:0", generator=g) * 2 - 1
        sa, sb = a.step(sa, act), b.step(sb, act)
        assert torch.equal(sa.obs, sb.obs) and torch.equal(sa.reward, sb.reward) and torch.equal(sa.done, sb.done), t
        assert torch.equal(sa.pipeline_state.qpos, sb.pipeline_state.qpos) and torch.equal(sa.info["cur_frame"], sb.info["cur_frame"])
    assert int((b._cost > 0).sum()) > N // 2            # (an env without constraint rows runs no line search: work 0)
    assert sorted(b._env_map.tolist()) == list(range(N)) and not torch.equal(b._env_map, torch.arange(N, dtype=torch.int32, device="cuda:0"))
    # heavy with light: the first half of the map holds the costlier half of the work estimates it was built from
    c = b._cost.cpu().numpy()
    b._rebalance()
    assert c[b._env_map[:N // 2].cpu().numpy()].mean() > c[b._env_map[N // 2:].cpu().numpy()].mean()
    # an explicit random permutation through the ABI
    perm = torch.randperm(N, device="cuda:0", generator=g).to(torch.int32)
    a._batch.set_schedule(perm, None)
    act = torch.rand(N, a.action_size, device="cuda:0", generator=g) * 2 - 1
    sa2, sb2 = a.step(sa, act), b.step(sb, act)
    assert torch.equal(sa2.obs, sb2.obs) and torch.equal(sa2.pipeline_state.qvel, sb2.pipeline_state.qvel)


def test_graph_replay_and_sub_batches_equal_host_issued_steps():
    """(1) Replaying a HIP graph of R wrapped env steps (`envs.graphed.GraphedSteps`) leaves the same state, bit for bit, as issuing
    the steps from the host.  (2) Stepping the batch as two sub-batches on two streams gives every env the same trajectory as the
    whole batch in one launch (envs are independent; the split is a scheduling choice)."""
    from rodent_amd import envs, jax_random
    from rodent_amd.envs import graphed, wrappers
    dev = torch.device("cuda:0")
    N, R = 64, 3
    keys = jax_random.split(jax_random.PRNGKey(5), N)
    acts = (torch.rand(2 * R, N, 30, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 2 - 1)

    def make(n, keys_, stream):
        with torch.cuda.stream(stream):
            env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=n, xml_path="rodent_optimized.xml",
                                       iterations=8, ls_iterations=8, device=dev)
            wenv = wrappers.wrap(env, episode_length=4, action_repeat=1)           # short episodes: the auto-reset path is exercised
            return wenv, wenv.reset(keys_)

    def run_host(wenv, state, a, stream):
        with torch.cuda.stream(stream):
            for t in range(a.shape[0]):
                state = wenv.step(state, a[t])
        return state

    s0 = torch.cuda.Stream(dev)
    wenv, st = make(N, keys, s0)
    want = run_host(wenv, st, acts, s0)
    torch.cuda.synchronize()
    # (1) graph replay; the action of a step is read through a cursor kept on the device (a replay cannot take host arguments)
    wenv_g, st_g = make(N, keys, s0)
    cursor = torch.zeros((), dtype=torch.long, device=dev)

    def step_fn(state):
        a = acts.index_select(0, cursor.reshape(1))[0]
        cursor.add_(1)
        return wenv_g.step(state, a)
    with torch.cuda.stream(s0):
        st1 = step_fn(st_g)                                   # host-issued first step: the template must be a step OUTPUT
    torch.cuda.synchronize()
    g = graphed.GraphedSteps(step_fn, st1, R, s0)             # capture only records: the cursor still reads 1
    got = g.replay()                                          # steps 2 .. R+1
    torch.cuda.synchronize()
    assert int(cursor) == 1 + R
    want_r = run_host(wenv_r := make(N, keys, s0)[0], wenv_r.reset(keys), acts[:1 + R], s0)
    torch.cuda.synchronize()
    la, lb = graphed.tree_leaves(got), graphed.tree_leaves(want_r)
    assert len(la) == len(lb) > 10
    for x, y in zip(la, lb):
        assert x.shape == y.shape and torch.equal(x, y)
    # (2) two sub-batches on two streams
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    h = N // 2
    w1, t1 = make(h, keys[:h], s1)
    w2, t2 = make(h, keys[h:], s2)
    o1, o2 = run_host(w1, t1, acts[:, :h], s1), run_host(w2, t2, acts[:, h:], s2)
    torch.cuda.synchronize()
    for name in ("obs", "reward", "done"):
        assert torch.equal(torch.cat([getattr(o1, name), getattr(o2, name)]), getattr(want, name)), name
    assert torch.equal(torch.cat([o1.pipeline_state.qpos, o2.pipeline_state.qpos]), want.pipeline_state.qpos)
    assert torch.equal(torch.cat([o1.info["steps"], o2.info["steps"]]), want.info["steps"])


def _nccl_graph_worker(port, out):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from rodent_amd import envs, jax_random
        from rodent_amd.envs import graphed, wrappers
        dev = torch.device("cuda:0")
        t = torch.ones(4, device=dev)
        torch.distributed.all_reduce(t)                       # the communicator and its watchdog thread are live
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=32, xml_path="rodent_optimized.xml",
                                       iterations=8, ls_iterations=8, device=dev)
            wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
            gen = torch.Generator(device=dev); gen.manual_seed(1)
            state = wenv.reset(jax_random.split(jax_random.PRNGKey(0), 32))

            def step_fn(s):
                return wenv.step(s, torch.empty(32, env.action_size, device=dev).uniform_(-1.0, 1.0, generator=gen))
            state = step_fn(state)
        torch.cuda.synchronize()
        g = graphed.GraphedSteps(step_fn, state, 5, st, [gen])
        for _ in range(4):
            s = g.replay()
        torch.distributed.barrier()
        torch.cuda.synchronize()
        out["ok"] = bool(torch.isfinite(s.obs).all()) and float(s.info["steps"].max()) > 0
    finally:
        torch.distributed.destroy_process_group()


def test_graph_capture_next_to_an_rccl_communicator():
    """bench.py's multi-GPU run captures the step while an RCCL process group (and its watchdog thread) is alive."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        p = ctx.Process(target=_nccl_graph_worker, args=(29731, out))
        p.start(); p.join(180)
        assert p.exitcode == 0 and out.get("ok") is True


def test_multi_step_rollout_equals_single_steps():
    """rr_env_unroll (T wrapped env steps in one launch, state on chip, Episode + AutoReset applied in place) leaves every leaf of
    the state exactly as T calls of the wrapped step do -- through episode ends (episode_length 7), unhealthy terminations and
    clip-frame saturation."""
    from rodent_amd import envs, jax_random
    from rodent_amd.envs import graphed, wrappers
    dev = torch.device("cuda:0")
    N, T = 96, 23
    g = torch.Generator(device=dev).manual_seed(11)
    acts = torch.rand(T, N, 30, device=dev, generator=g) * 2 - 1
    keys = jax_random.split(jax_random.PRNGKey(8), N)

    def make():
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path="rodent_optimized.xml", iterations=8,
                                   ls_iterations=8, device=dev, healthy_z_range=(0.045, 0.5))
        wenv = wrappers.wrap(env, episode_length=7, action_repeat=1)
        return wenv, wenv.step(wenv.reset(keys), acts[0])
    wenv, st = make()
    want = st
    for t in range(1, T):
        want = wenv.step(want, acts[t])
    wenv2, st2 = make()
    got = wenv2.unroll(st2, acts[1:])
    torch.cuda.synchronize()
    assert float(want.info["steps"].max()) <= 7 and float((want.done > 0).float().sum()) >= 0
    la, lb = graphed.tree_leaves(got), graphed.tree_leaves(want)
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        assert x.shape == y.shape and torch.equal(x, y)
    # chained launches (8 + 14 steps) give the same state
    got2 = wenv2.unroll(wenv2.unroll(st2, acts[1:9]), acts[9:])
    torch.cuda.synchronize()
    for x, y in zip(graphed.tree_leaves(got2), lb):
        assert torch.equal(x, y)
    # configurations without a multi-step instance refuse loudly
    env_n = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=8, xml_path="rodent_optimized.xml", iterations=4,
                                 ls_iterations=8, solver="newton", device=dev)
    wn = wrappers.wrap(env_n, episode_length=7, action_repeat=1)
    sn = wn.step(wn.reset(keys[:8]), acts[0, :8])
    with pytest.raises(RuntimeError, match="multi-step"):
        wn.unroll(sn, acts[1:3, :8].contiguous())
