"""bench.py -- env-steps/sec of random-action rollouts, BASELINE.json config 2:
`rodent_optimized.xml`, 2048 envs per GPU, CG 8/8, n_frames 10 (1 env-step = 10 physics substeps).

A "step" is one `Rodent.step` over the whole 2048-env batch through the training wrappers
(Episode(150) + AutoReset), i.e. one fused HIP launch (physics x10 + reward/done/obs epilogue) plus
the wrapper selects; actions are fresh U(-1,1) draws made on the GPU each step.  All inputs are
resident in HBM before the timed region.  N>1: one process per GPU (torch.distributed, backend nccl
= RCCL), env shards are independent, no data-path collective ("weak" scaling); timing is barrier +
synchronize on both sides and the MAX over ranks.

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


def pmc_traffic():
    """HBM-side bytes per launch of the step kernel from the latest committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in their own runs, tools/pmc_traffic.py); None when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["traffic_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(n_envs=512, steps=100):
    """The oracle (kind 'port': our C restatement; the reference JAX path cannot run here) timed on the host cores."""
    from oracle import ref
    from rodent_amd import assets, mjcf
    ref.build()
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    path = assets.asset_path(MODEL)
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f32")
    M.set_iterations(8, 8)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--num-envs", type=int, default=NUM_ENVS, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a 1-GPU box: RR_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two ranks
    # on one device); the driver's multi-GPU run uses one GPU per rank over RCCL
    share = os.environ.get("RR_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dist = None
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from rodent_amd import envs
    from rodent_amd.envs import wrappers

    N = args.num_envs
    env = envs.get_environment("rodent", track_pos=synthetic_track(), num_envs=N, xml_path=f"{MODEL}.xml",
                               terminate_when_unhealthy=True, solver="cg", iterations=8, ls_iterations=8, device=dev)
    wenv = wrappers.wrap(env, episode_length=150, action_repeat=1)
    from rodent_amd import jax_random
    keys = jax_random.split(jax_random.fold_in(jax_random.PRNGKey(0), rank), N)
    state = wenv.reset(keys)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def one_step(state):
        action = torch.rand(N, env.action_size, device=dev, generator=gen) * 2 - 1
        return wenv.step(state, action)

    for _ in range(args.warmup):
        state = one_step(state)
    env._batch.set_timing(True)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = one_step(state)
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = env._batch.kernel_time()
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(state.obs).all(), "non-finite observation in the rollout"

    if rank == 0:
        total_env_steps = N * world * args.steps
        value = total_env_steps / elapsed
        d = env.sys.model.dims
        S = d.nq + d.nv + d.na + d.nv
        bytes_per_env_step = 4 * (2 * S + d.nu + d.obs_dim + 2)          # 7180 B for rodent_optimized
        avg_kernel_s = (kern_ms / max(launches, 1)) * 1e-3
        achieved = bytes_per_env_step * N / avg_kernel_s / 1e9
        out = {
            "metric": "env-steps/sec (whole node), rodent 2048 envs/GPU", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{MODEL}.xml random-action rollout, Rodent.step through Episode(150)+AutoReset wrappers, "
                                   f"CG 8/8, n_frames 10", "envs_per_gpu": N, "global_envs": N * world,
                       "parallelism": f"env-shards x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
                         "kernel": "rr_step_kernel", "avg_kernel_ms": avg_kernel_s * 1e3, "launches": launches,
                         "algorithmic_bytes_per_env_step": bytes_per_env_step},
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
