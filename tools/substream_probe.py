"""Does stepping the 2048 environments as S independent sub-batches on S HIP streams shorten the step?  A launch lasts as long
as its slowest environment; with sub-batches in flight on separate streams one sub-batch's tail overlaps the others' bulk.
Config-2 workload (random actions, Episode + AutoReset wrappers).  usage: substream_probe.py [S ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import torch
import bench
from rodent_amd import envs, jax_random
from rodent_amd.envs import wrappers, graphed

dev = torch.device("cuda:0")
N, STEPS, WARM, R = 2048, 300, 50, 10
GRAPH = os.environ.get("RR_PROBE_GRAPH", "1") == "1"
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n = N // S
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    subs = []
    for s, st in enumerate(streams):
        with torch.cuda.stream(st):
            env = envs.get_environment("rodent", track_pos=bench.synthetic_track(), num_envs=n, xml_path="rodent_optimized.xml",
                                       terminate_when_unhealthy=True, solver="cg", iterations=8, ls_iterations=8, device=dev)
            wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
            gen = torch.Generator(device=dev); gen.manual_seed(100 + s)
            state = wenv.reset(jax_random.split(jax_random.fold_in(jax_random.PRNGKey(0), s), n))
            subs.append([wenv, env, gen, state])
    torch.cuda.synchronize()

    def run(k):
        for _ in range(k):
            for sub, st in zip(subs, streams):
                with torch.cuda.stream(st):
                    wenv, env, gen, state = sub
                    a = torch.empty(n, env.action_size, device=dev).uniform_(-1.0, 1.0, generator=gen)
                    sub[3] = wenv.step(state, a)
    run(WARM)
    torch.cuda.synchronize()
    UT = int(os.environ.get("RR_PROBE_UNROLL", "0"))
    if UT:          # multi-step launches: UT wrapped steps per launch, actions drawn UT steps at a time
        def run(k):
            for _ in range(k // UT):
                for sub, st in zip(subs, streams):
                    with torch.cuda.stream(st):
                        wenv, env, gen, state = sub
                        a = torch.empty(UT, n, env.action_size, device=dev).uniform_(-1.0, 1.0, generator=gen)
                        sub[3] = wenv.unroll(state, a)
        run(2 * UT)
        torch.cuda.synchronize()
    elif GRAPH:
        def make(sub):
            wenv, env, gen, state = sub

            def step_fn(st_):
                a = torch.empty(n, env.action_size, device=dev).uniform_(-1.0, 1.0, generator=gen)
                return wenv.step(st_, a)
            return step_fn
        gs = [graphed.GraphedSteps(make(sub), sub[3], R, st, [sub[2]]) for sub, st in zip(subs, streams)]
        torch.cuda.synchronize()

        def run(k):
            for _ in range(k // R):
                for g in gs:
                    g.replay()
        run(5 * R)
        torch.cuda.synchronize()
        off = int(os.environ.get("RR_PROBE_OFFSET_CYCLES", "0"))
        if off and S > 1:                       # start the sub-batches out of phase (stream s waits s * off spin cycles first)
            for si, st in enumerate(streams):
                with torch.cuda.stream(st):
                    if si:
                        t_a = torch.cuda.Event(enable_timing=True); t_b = torch.cuda.Event(enable_timing=True)
                        t_a.record(); torch.cuda._sleep(si * off); t_b.record()
            torch.cuda.synchronize()
            print("sleep of", off, "cycles took", t_a.elapsed_time(t_b), "ms", flush=True)
            for si, st in enumerate(streams):   # again, this time followed immediately by the timed replays
                with torch.cuda.stream(st):
                    if si:
                        torch.cuda._sleep(si * off)
    t0 = time.perf_counter()
    run(STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if GRAPH and not UT:
        assert all(torch.isfinite(g.state.obs).all() for g in gs)
    else:
        assert all(torch.isfinite(sub[3].obs).all() for sub in subs)
    print(json.dumps({"graph": GRAPH and not UT, "unroll": UT, "substreams": S, "envs_each": n, "env_steps_per_s": N * STEPS / dt, "ms_per_2048_env_step": dt / STEPS * 1e3}), flush=True)
    del subs
