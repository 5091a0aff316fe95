"""bench.py -- env-steps/sec; default = BASELINE.json config 2 (random-action rollouts; the configuration the metric is quoted
on); `--config 3` = the full PPO loop with the launcher's hyper-parameters, `--config 5` = rodent_pair.xml physics only.
Config 2:
`rodent_optimized.xml`, 2048 envs per GPU, CG 8/8, n_frames 10 (1 env-step = 10 physics substeps).

A "step" is one `Rodent.step` over the whole 2048-env batch through the training wrappers
(Episode(150) + AutoReset), i.e. one fused HIP launch (physics x10 + reward/done/obs epilogue) plus
the wrapper selects; actions are fresh U(-1,1) draws made on the GPU each step.  All inputs are
resident in HBM before the timed region.  N>1: one process per GPU (torch.distributed, backend nccl
= RCCL), env shards are independent, no data-path collective ("weak" scaling); timing is barrier +
synchronize on both sides and the MAX over ranks.

Protocol: with explicit --steps K --warmup W exactly K steps are timed once (the driver's contract).  With neither flag the
SURVEY.md 8(d) protocol runs: 50 warm-up steps, 1000 timed steps, 5 repeats, the MEDIAN repeat is reported (covers
episode boundaries at 150 steps).

Prints ONE JSON line (rank 0) with `roofline` (HBM, algorithmic bytes/env-step from SURVEY.md 8(d) x
envs per launch / mean kernel time from hipEvents on the launch stream) and `cpu_baseline` (the C
oracle, float32, OpenMP over envs, bounded sample, on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NUM_ENVS = 2048
MODEL = "rodent_optimized"


def synthetic_track(T=250):
    t = np.arange(T, dtype=np.float64)
    return np.stack([0.004 * t, np.zeros(T), np.full(T, 0.0681)], axis=1)


def host_cores():
    """CPU share of this process: cgroup quota when set (the 1-GPU box grants 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("RR_CPU_THREADS", "16")))


def pmc_traffic(env_steps_per_launch):
    """HBM-side bytes per LAUNCH of the step kernel: NOT measured by this run (rocprofv3 PMC passes cannot run inside it) but read
    from the latest committed summary (FETCH_SIZE and WRITE_SIZE collected in their own runs of the bench command, tools/
    profile_round3.sh -> tools/pmc_traffic.py).  Returned only when that summary was taken in the SAME launch mode (env steps per
    launch, hence the same kernel instance) as this run; the source file and its mode always go into the line (`traffic_source`)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        mode = int(d.get("env_steps_per_launch", 1))
        src = f"profiles/{os.path.basename(files[-1])} (rocprofv3 --pmc passes of `{d.get('command', 'bench.py --unroll 1 --no-graph')}`: {mode} env step(s) per launch)"
        if mode != env_steps_per_launch:
            return None, src + f"; this run launches {env_steps_per_launch} env steps at a time: not comparable, traffic omitted"
        return float(d["traffic_bytes_per_launch"]), src
    except Exception:
        return None, None


def sq_counters(env_steps_per_launch, envs):
    """VALU-busy share and instructions per env-step of the step kernel from the latest committed SQ counter pass (profiles/r*_sq.json;
    NOT measured by this run -- `counters_source` names the file and the launch mode it was taken in); the kernel is issue / latency
    bound, so these -- not the HBM fraction the north star asks for -- are the numbers that track kernel quality.  valu_issue_floor_ms:
    the time one bench step would take if the SIMDs' vector ALUs never idled = VALU instructions per env-step x envs per SIMD x 4 cycles
    (wave64 on a 16-lane-per-cycle... SIMD: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU measures 4.1) / 2.4 GHz."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        out = {k: d[k] for k in ("valu_busy_frac", "instructions_per_env_step", "valu_instructions_per_env_step", "lds_bank_conflict_frac") if k in d}
        mode = int(d.get("env_steps_per_launch", 1))
        out["counters_source"] = (f"profiles/{os.path.basename(files[-1])} ({mode} env step(s) per launch"
                                  + ("" if mode == env_steps_per_launch else f"; this run: {env_steps_per_launch}") + ")")
        if "valu_instructions_per_env_step" in d:
            out["valu_issue_floor_ms"] = d["valu_instructions_per_env_step"] * (envs / 1024.0) * 4.0 / 2.4e9 * 1e3     # 256 CUs x 4 SIMDs
        return out
    except Exception:
        return {}


def cpu_baseline(n_envs=512, steps=100, solver="cg", iterations=(8, 8)):
    """The oracle (kind 'port': our C restatement; the reference JAX path cannot run here) timed on the host cores."""
    from oracle import ref
    from rodent_amd import assets, mjcf
    ref.build()
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    path = assets.asset_path(MODEL)
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f32")
    M.set_iterations(*iterations)
    M.set_solver(solver)
    rng = np.random.default_rng(0)
    datas = []
    for e in range(n_envs):
        d = ref.RefData(M)
        q = m["qpos0"].astype(np.float64) + rng.uniform(-0.01, 0.01, M.nq)
        d.init(q, rng.uniform(-0.01, 0.01, M.nv))
        datas.append(d)
    for _ in range(2):
        ref.step_batch(M, datas, rng.uniform(-1, 1, (n_envs, M.nu)), 10)
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.step_batch(M, datas, rng.uniform(-1, 1, (n_envs, M.nu)), 10)
    dt = time.perf_counter() - t0
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {steps} env-steps (10 substeps each), C oracle float32, OpenMP over envs, {dt:.1f} s"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 5), help="BASELINE.json config (2 = the headline metric's; 1 = Rodent.step, 4 envs, rodent_cpu.xml)")
    ap.add_argument("--num-envs", type=int, default=None, help="envs per GPU (default 2048; 4096 for config 5; 4 for config 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--substreams", type=int, default=1, help="config 2: the envs of a GPU are stepped as this many sub-batches on separate "
                    "HIP streams (one sub-batch's slowest envs overlap the others' bulk); 1 = one launch per step")
    ap.add_argument("--unroll", type=int, default=250, help="config 2: env steps per launch (rr_env_unroll: the wrapped step scanned inside the kernel, "
                    "actions drawn for that many steps at a time); 1 = one launch per step (HIP-graph replay unless --no-graph)")
    ap.add_argument("--balance", action="store_true", help="config 2: pair the costliest envs of the previous launch with the cheapest on a SIMD "
                    "(rr_batch_set_schedule); result-neutral")
    ap.add_argument("--no-graph", action="store_true", help="config 2: issue every step from the host instead of replaying a HIP graph of it")
    ap.add_argument("--solver", default="cg", choices=("cg", "newton"), help="config 2 only; the headline configuration is cg 8/8")
    ap.add_argument("--iterations", type=int, default=8)
    ap.add_argument("--ls-iterations", type=int, default=8)
    args = ap.parse_args()
    args.protocol = "driver" if (args.steps is not None or args.warmup is not None) else "survey-8d"
    if args.config == 3:
        dflt = (2, 1, 1)
    else:
        dflt = (1000, 50, 5) if args.protocol == "survey-8d" else (100, 20, 1)
    args.repeats = dflt[2]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup
    if args.num_envs is None:
        args.num_envs = 4096 if args.config == 5 else (4 if args.config == 1 else NUM_ENVS)
    return args


class _stdout_to_stderr:
    """Rank 0's stdout carries ONE JSON line: native libraries that greet on stdout while the process group forms (gloo prints its
    peer count) are sent to stderr for that stretch (file-descriptor level, so C++ streams are covered)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, as torch.distributed.run sets them), from a parent that never touches a GPU (no exec of a process
    that initialised HIP; `torch.cuda.device_count()` does not initialise it on this image).  Rank 0's stdout is the JSON line."""
    import socket
    import subprocess
    share = os.environ.get("RR_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not share and os.environ.get("RR_BENCH_CPU_REHEARSAL") != "1":
        raise SystemExit(f"bench.py --gpus {n}: {have} GPU(s) visible.  One process per GPU is the only mode measured; "
                         "RR_BENCH_SHARE_GPU=1 rehearses the N-rank path on one GPU (gloo) and is not a scaling number.")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p_ in procs:
            rc = rc or p_.wait()
            if rc:                                   # one rank failed: the others would wait in a collective forever
                break
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            try:
                p_.wait(timeout=30)
            except Exception:
                p_.kill()
    return rc


def cpu_rehearsal(args, rank, world):
    """RR_BENCH_CPU_REHEARSAL=1 (tests/test_bench_launcher.py; NOT a measurement): the launcher, the rendezvous, ppo.train's multi-rank
    path and the rollout / learner / all-reduce split on the CPU over gloo with a toy env -- the N-rank command shape without GPUs."""
    import torch.distributed as dist
    from rodent_amd.training.agents.ppo import train as ppo
    from tests.fake_env import PointEnv
    if world > 1:
        with _stdout_to_stderr():
            dist.init_process_group("gloo")
    times = []
    ppo.train(environment=PointEnv(16 * world), num_timesteps=10 ** 9, episode_length=20, num_envs=16 * world, batch_size=16 * world,
              num_minibatches=2, unroll_length=5, num_updates_per_batch=2, num_evals=2, num_eval_envs=0, normalize_observations=True, seed=0,
              max_training_steps=args.warmup + args.steps, timing_fn=times.append)
    timed = times[args.warmup:]
    el = torch.tensor([sum(t["rollout_s"] + t["learner_s"] for t in timed)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "env-steps/sec (whole node), rodent 2048 envs/GPU", "value": timed[0]["env_steps"] * len(timed) / float(el), "unit": "env-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "data": "cpu-rehearsal (toy env, gloo): not a measurement",
                          "config": {"workload": "launcher rehearsal", "env_steps_per_training_step": timed[0]["env_steps"],
                                     "rollout_s_per_training_step": sum(t["rollout_s"] for t in timed) / len(timed),
                                     "learner_s_per_training_step": sum(t["learner_s"] for t in timed) / len(timed),
                                     "allreduce_s_per_training_step": sum(t["allreduce_s"] for t in timed) / len(timed)}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, or without a launcher)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RR_BENCH_CPU_REHEARSAL") == "1":
        return cpu_rehearsal(args, rank, world)
    # rehearsal on a 1-GPU box: RR_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two ranks
    # on one device); the driver's multi-GPU run uses one GPU per rank over RCCL
    share = os.environ.get("RR_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        with _stdout_to_stderr():
            if share:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dist = None
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from rodent_amd import envs, jax_random
    from rodent_amd.envs import wrappers

    N = args.num_envs
    model = "rodent_pair" if args.config == 5 else ("rodent_cpu" if args.config == 1 else MODEL)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    extra = {}
    if args.config == 3:
        # ---- the full PPO loop: a "step" is one TRAINING step = 64 unrolls x 10 env steps x N envs + 512 minibatch updates
        from rodent_amd.training.agents.ppo import train as ppo
        env = envs.get_environment("rodent", track_pos=synthetic_track(), num_envs=N, xml_path=f"{model}.xml",
                                   terminate_when_unhealthy=True, solver="cg", iterations=8, ls_iterations=8, device=dev)
        env._batch.set_timing(True)
        times = []
        fence()
        ppo.train(environment=env, num_timesteps=500_000_000, num_evals=100, reward_scaling=1, episode_length=150,
                  normalize_observations=True, action_repeat=1, unroll_length=10, num_minibatches=64, num_updates_per_batch=8,
                  discounting=0.97, learning_rate=5e-5, entropy_cost=1e-3, num_envs=N * world, batch_size=N * world, seed=0,
                  num_eval_envs=0, max_training_steps=args.warmup + args.steps, timing_fn=times.append)
        fence()
        timed = times[args.warmup:]
        elapsed = max_over_ranks(sum(t["rollout_s"] + t["learner_s"] for t in timed))
        env_steps_per_step = timed[0]["env_steps"]            # global (all ranks)
        total_env_steps = env_steps_per_step * len(timed)
        extra = {"rollout_s_per_training_step": sum(t["rollout_s"] for t in timed) / len(timed),
                 "learner_s_per_training_step": sum(t["learner_s"] for t in timed) / len(timed),
                 # inside learner_s: the 512 gradient all-reduces + the normaliser's, as the learner's stream sees them (rank 0; 0 on one rank)
                 "allreduce_s_per_training_step": sum(t["allreduce_s"] for t in timed) / len(timed),
                 "env_steps_per_training_step": env_steps_per_step}
        # the rollout phase's launches carry many env steps each (the whole phase is one launch when the actor runs in-kernel): the
        # roofline line is per env step of all envs = total step-kernel time / rollout steps per env (the reset launch counts ~0.1 %)
        extra["rollout_steps_per_env"] = env_steps_per_step // (N * world) * len(times)
        workload = ("full PPO loop, launcher hyper-parameters (unroll 10, 64 minibatches x 8 epochs, lr 5e-5), rodent_optimized.xml, "
                    "CG 8/8, n_frames 10; step = one training step")
        batch = env._batch
        repeats_ms = [elapsed / len(timed) * 1e3]
        extra["step_kernel_time_source"] = "hip events around the rollout's launches"
        if batch.kernel_time()[1] == 0:
            # the rollouts ran as sub-batches replayed from HIP graphs (acting.SubBatchRollout): events cannot be read back from a replayed
            # graph, so the step kernel is timed on a short host-issued pass of the same env batch (random actions)
            st_ = env.reset(0)
            for _ in range(30):
                st_ = env.step(st_, torch.empty(N, env.action_size, device=dev).uniform_(-1.0, 1.0, generator=gen))
            torch.cuda.synchronize(dev)
            extra["step_kernel_time_source"] = "host-issued pass of 30 steps after training (rollouts are HIP-graph replays of sub-batches)"
    elif args.config == 1:
        # ---- BASELINE config 1: Rodent.step() with num_envs = 4 on rodent_cpu.xml (self-collisions between sphere / capsule pairs, tendon
        # transmissions, no floor / free joint) through the DYN kernel instance; plumbing-sized: 4 waves on the whole GPU, launch-latency bound
        env = envs.get_environment("rodent", track_pos=synthetic_track(), num_envs=N, xml_path=f"{model}.xml", terminate_when_unhealthy=True,
                                   solver="cg", iterations=6, ls_iterations=6, device=dev)
        wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
        state = wenv.reset(jax_random.split(jax_random.fold_in(jax_random.PRNGKey(0), rank), N))
        nu = env.action_size
        batch = env._batch
        workload = f"{model}.xml Rodent.step through Episode(150)+AutoReset wrappers, {N} envs, CG 6/6, n_frames 10, random actions (DYN instance)"
        for _ in range(args.warmup):
            state = wenv.step(state, torch.empty(N, nu, device=dev).uniform_(-1.0, 1.0, generator=gen))
        batch.set_timing(True)
        repeats_ms = []
        for rep in range(args.repeats):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                state = wenv.step(state, torch.empty(N, nu, device=dev).uniform_(-1.0, 1.0, generator=gen))
            fence()
            repeats_ms.append(max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3)
        assert torch.isfinite(state.obs).all(), "non-finite state in the rollout"
        total_env_steps = N * world * args.steps
        elapsed = float(np.median(repeats_ms)) * args.steps * 1e-3
    else:
        if args.config == 2:
            # The N envs of this GPU are stepped as S sub-batches of N / S envs, each with its own batch, stream, wrapper state and
            # action generator.  A launch lasts as long as its slowest env (one resident round: 2 waves per SIMD); with S launches in
            # flight one sub-batch's tail overlaps the others' bulk.  Every env still makes exactly one step per bench step.
            from rodent_amd.envs import graphed
            S_ = args.substreams
            if N % S_:
                raise SystemExit(f"--substreams {S_} does not divide {N} envs")
            n_sub = N // S_
            keys = jax_random.split(jax_random.fold_in(jax_random.PRNGKey(0), rank), N)
            subs = []
            for si in range(S_):
                st = torch.cuda.Stream(dev)
                with torch.cuda.stream(st):
                    env = envs.get_environment("rodent", track_pos=synthetic_track(), num_envs=n_sub, xml_path=f"{model}.xml",
                                               terminate_when_unhealthy=True, solver=args.solver, iterations=args.iterations,
                                               ls_iterations=args.ls_iterations, device=dev, balance=args.balance, rebalance_every=1)
                    wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
                    g_ = torch.Generator(device=dev)
                    g_.manual_seed(1234 + 64 * rank + si)
                    subs.append(dict(stream=st, env=env, wenv=wenv, gen=g_, state=wenv.reset(keys[si * n_sub:(si + 1) * n_sub])))
            nu = subs[0]["env"].action_size

            def sub_step(sub, state):
                action = torch.empty(n_sub, nu, device=dev).uniform_(-1.0, 1.0, generator=sub["gen"])      # fresh U(-1,1) draws, one launch
                return sub["wenv"].step(state, action)

            def eager_steps(k):
                for _ in range(k):
                    for sub in subs:
                        with torch.cuda.stream(sub["stream"]):
                            sub["state"] = sub_step(sub, sub["state"])
            single_steps = eager_steps
            workload = (f"{model}.xml random-action rollout, Rodent.step through Episode(150)+AutoReset wrappers, "
                        f"{args.solver.upper()} {args.iterations}/{args.ls_iterations}, n_frames 10; {S_} sub-batches of {n_sub} envs on {S_} streams, "
                        + ("host-issued steps" if args.no_graph else "HIP-graph replay of the step"))
            eager_steps(args.warmup)
            torch.cuda.synchronize(dev)
            R_ = 1
            UT_ = max(r for r in range(1, max(args.unroll, 1) + 1) if args.steps % r == 0)      # env steps per launch
            if UT_ > 1 and not all(sub["env"]._batch.unroll_supported() for sub in subs):          # e.g. the Newton instance: one launch per step
                UT_ = 1
            if UT_ > 1:
                # multi-step launches: the rollout's scan over the wrapped step runs INSIDE the kernel (envs never wait for each other between
                # steps, state on chip, wrappers in place; bit-identical to the per-step calls: tests/test_gpu_env.py)
                workload = (f"{model}.xml random-action rollout, Rodent.step through Episode(150)+AutoReset wrappers, "
                            f"{args.solver.upper()} {args.iterations}/{args.ls_iterations}, n_frames 10; "
                            + (f"{S_} sub-batches of {n_sub} envs on {S_} streams, " if S_ > 1 else "")
                            + f"{UT_} env steps per launch (rr_env_unroll: the rollout's scan inside the kernel)")
                args.no_graph = True

                def unroll_steps(k):
                    for _ in range(k // UT_):
                        for sub in subs:
                            with torch.cuda.stream(sub["stream"]):
                                a_ = torch.empty(UT_, n_sub, nu, device=dev).uniform_(-1.0, 1.0, generator=sub["gen"])
                                sub["state"] = sub["wenv"].unroll(sub["state"], a_)
                unroll_steps(2 * UT_)
                torch.cuda.synchronize(dev)
                for sub in subs:
                    sub["env"]._batch.set_timing(True)        # live: HIP events around every launch of the timed region
                eager_steps = unroll_steps
            if not args.no_graph:
                R_ = max(r for r in range(1, 11) if args.steps % r == 0)            # steps per replay
                try:
                    for sub in subs:
                        sub["graph"] = graphed.GraphedSteps(lambda st_, sub=sub: sub_step(sub, st_), sub["state"], R_, sub["stream"], [sub["gen"]])
                except Exception as e:                                               # keep measuring: issue the steps from the host
                    print(f"bench: HIP-graph capture of the step failed ({e!r}); host-issued steps", file=sys.stderr)
                    args.no_graph, R_ = True, 1
                    workload = workload.replace("HIP-graph replay of the step", "host-issued steps (graph capture failed)")
                torch.cuda.synchronize(dev)
            repeats_ms = []
            for rep in range(args.repeats):
                fence()
                t0 = time.perf_counter()
                if args.no_graph:
                    eager_steps(args.steps)
                else:
                    for _ in range(args.steps // R_):
                        for sub in subs:
                            sub["graph"].replay()
                fence()
                repeats_ms.append(max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3)
            if not args.no_graph:
                for sub in subs:
                    sub["state"] = sub["graph"].state
            assert all(torch.isfinite(sub["state"].obs).all() for sub in subs), "non-finite state in the rollout"
            # kernel durations by HIP events on each sub-batch's stream: a host-issued pass over the same states (events cannot be
            # read back from inside a replayed graph); rocprofv3's kernel trace of this command shows the replayed launches themselves
            if UT_ == 1:
                for sub in subs:
                    sub["env"]._batch.set_timing(True)
                eager_steps(args.steps if args.no_graph else min(args.steps, 100))
            torch.cuda.synchronize(dev)
            kt = [sub["env"]._batch.kernel_time() for sub in subs]
            kern_ms, launches = sum(k[0] for k in kt), sum(k[1] for k in kt)
            batch = subs[0]["env"]._batch
            total_env_steps = N * world * args.steps
            elapsed = float(np.median(repeats_ms)) * args.steps * 1e-3
            extra = {"substreams": S_, "steps_per_graph_replay": None if args.no_graph else R_, "env_steps_per_launch": UT_}
            if UT_ > 1:      # for the record, outside the timed region: the same rollout as one synchronised launch per step (Rodent.step per call)
                for sub in subs:
                    sub["env"]._batch.set_timing(False)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                single_steps(50)
                torch.cuda.synchronize(dev)
                extra["ms_per_step_one_launch_per_step"] = (time.perf_counter() - t1) / 50 * 1e3
        else:
            # ---- config 5: rodent_pair.xml (two replicated rodents, nv 146, 114 contacts), physics only (pipeline_step)
            from rodent_amd import assets, hip, mjcf
            path = assets.asset_path(model)
            m = mjcf.load_blob(path)
            batch = hip.Batch(hip.Model(path, 8, 8), N, dev)
            d = batch.dims
            nu = d.nu
            q = torch.tensor(np.tile(m["qpos0"], (N, 1)), dtype=torch.float32, device=dev)
            q += (torch.rand(N, d.nq, device=dev, generator=gen) * 2 - 1) * 0.01
            state = dict(qpos=q, qvel=torch.zeros(N, d.nv, device=dev), act=torch.zeros(N, d.na, device=dev),
                         qacc_warmstart=torch.zeros(N, d.nv, device=dev))
            batch.pipeline_init(state)
            first = {k: v.clone() for k, v in state.items()}

            def one_step(state):
                ctrl = torch.empty(N, nu, device=dev).uniform_(-1.0, 1.0, generator=gen)
                out = {k: torch.empty_like(v) for k, v in state.items()}
                batch.pipeline_step_to(state, out, ctrl, 10)
                bad = (out["qpos"][:, 2] < 0.03) | (out["qpos"][:, 2] > 0.5) | ~torch.isfinite(out["qpos"]).all(1)   # keep the population in contact
                for k in out:
                    out[k] = torch.where(bad[:, None], first[k], out[k])
                return out
            workload = f"{model}.xml physics only (pipeline_step, n_frames 10), random actions, CG 8/8, restart of fallen envs"
            for _ in range(args.warmup):
                state = one_step(state)
            batch.set_timing(True)
            repeats_ms = []
            for rep in range(args.repeats):
                fence()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    state = one_step(state)
                fence()
                repeats_ms.append(max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3)
            assert torch.isfinite(state["qpos"]).all(), "non-finite state in the rollout"
            total_env_steps = N * world * args.steps
            elapsed = float(np.median(repeats_ms)) * args.steps * 1e-3
    if args.config != 2:
        kern_ms, launches = batch.kernel_time()

    if rank == 0:
        value = total_env_steps / elapsed
        d = batch.dims
        S = d.nq + d.nv + d.na + d.nv
        # full env step 4(2S + nu + obs + 2) = 7180 B (rodent_optimized); physics only 4(2S + nu) (config 5: 3880 B)
        bytes_per_env_step = 4 * (2 * S + d.nu + (d.obs_dim + 2 if args.config != 5 else 0))
        avg_kernel_s = (kern_ms / max(launches, 1)) * 1e-3
        # config 2: `substreams` launches of N / substreams envs are in flight together, each lasting avg_kernel_s
        concurrent = extra.get("substreams", 1)
        per_launch = extra.get("env_steps_per_launch", 1)          # env steps of every env inside one launch
        if args.config == 3 and launches and extra.get("step_kernel_time_source", "").startswith("hip events"):
            per_launch = max(1.0, extra["rollout_steps_per_env"] / launches)
        achieved = bytes_per_env_step * (N // concurrent) * per_launch * concurrent / avg_kernel_s / 1e9
        traffic, traffic_src = pmc_traffic(int(per_launch)) if args.config == 2 else (None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                # `achieved` prices every env step at SURVEY.md 8(d)'s algorithmic 7180 B (state in + out, action, observation); a multi-step
                # launch keeps the state on chip, so the bytes it must move per env step are the action and the observation only
                "achieved_basis": "SURVEY.md 8(d) algorithmic bytes x env steps of the launch / launch duration (HIP events, live)",
                "bytes_moved_per_env_step_multi_step_launch": 4 * (d.nu + (d.obs_dim + 2 if args.config != 5 else 0)),
                # avg_kernel_ms: kernel time of ONE bench step (all envs advance one step) = launch duration / env steps per launch
                # (sub-batch launches run side by side); avg_launch_ms: the duration of a launch itself, as rocprofv3 lists it
                "kernel": "rr_step_kernel", "avg_kernel_ms": avg_kernel_s * 1e3 / per_launch, "avg_launch_ms": avg_kernel_s * 1e3, "launches": launches,
                "envs_per_launch": N // concurrent, "concurrent_launches": concurrent, "env_steps_per_launch": per_launch,
                "algorithmic_bytes_per_env_step": bytes_per_env_step,
                "limiter": "VALU issue + dependent LDS/L2 latency, not HBM (SURVEY.md 8(d)); see valu_busy_frac"}
        if args.config == 2:
            roof.update(sq_counters(int(per_launch), N))
        out = {
            "metric": "env-steps/sec (whole node), rodent 2048 envs/GPU", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.config, "envs_per_gpu": N, "global_envs": N * world,
                       "parallelism": f"env-shards x{world}" + (", no data-path collective" if args.config != 3 else
                                                                ", gradient + normaliser all-reduce (RCCL)"),
                       "protocol": args.protocol, "repeats": args.repeats, "ms_per_step_repeats": repeats_ms, **extra},
            "roofline": roof,
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            out["cpu_baseline"] = cpu_baseline(solver=args.solver, iterations=(args.iterations, args.ls_iterations))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
