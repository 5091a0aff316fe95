"""Launcher with the configuration of the reference's `brax_rodent_run_ppo.py` [REF :39-55,97-114],
against this package instead of brax (only the imports change):

    from rodent_amd import envs
    from rodent_amd.training.agents.ppo import train as ppo
    from rodent_amd.io import model

Single GPU:  python examples/rodent_run_ppo.py
N GPUs:      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/rodent_run_ppo.py
wandb / video rendering of the reference are observability only and are replaced by a JSON-lines log.
"""
import argparse
import functools
import json
import os
import sys
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))

import numpy as np
import torch

from rodent_amd import envs, preprocessing, rollout
from rodent_amd.io import model
from rodent_amd.training.agents.ppo import train as ppo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-timesteps", type=int, default=500_000_000)
    ap.add_argument("--eval-every", type=int, default=5_000_000)
    ap.add_argument("--envs-per-gpu", type=int, default=1024)          # [REF :43] 1024 * n_gpus
    ap.add_argument("--xml", default="./models/rodent_new.xml")        # [REF Rodent_Env_Brax.py:16]
    ap.add_argument("--clip", default=None, help="reference clip: .npz / .h5 written by preprocessing.save_reference_clip (clip name "
                    "--clip-name) or .npy with the root positions [T,3]; a synthetic line if absent")
    ap.add_argument("--clip-name", default="84")
    ap.add_argument("--max-training-steps", type=int, default=None)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    config = {
        "env_name": "rodent", "algo_name": "ppo", "task_name": "run", "num_envs": args.envs_per_gpu * n_gpus,
        "num_timesteps": args.num_timesteps, "eval_every": args.eval_every, "episode_length": 150,
        "batch_size": args.envs_per_gpu * n_gpus, "learning_rate": 5e-5, "terminate_when_unhealthy": True,
        "solver": "cg", "iterations": 8, "ls_iterations": 8, "vision": False,
    }
    if args.clip and os.path.exists(args.clip) and not args.clip.endswith(".npy"):
        track_pos = np.asarray(preprocessing.load_reference_clip(args.clip, args.clip_name).position[0])     # reference_clip.position [REF :84]
    elif args.clip and os.path.exists(args.clip):
        track_pos = np.load(args.clip)
    else:   # the reference clip (clips/84.p) is not distributed: straight line at 0.2 m/s, torso rest height
        t = np.arange(250)
        track_pos = np.stack([0.004 * t, np.zeros(250), np.full(250, 0.0681)], axis=1)

    envs.register_environment("rodent", envs.Rodent)
    env = envs.get_environment(
        config["env_name"], track_pos=track_pos, terminate_when_unhealthy=config["terminate_when_unhealthy"],
        solver=config["solver"], iterations=config["iterations"], ls_iterations=config["ls_iterations"],
        vision=config["vision"], num_envs=args.envs_per_gpu, xml_path=args.xml, device=f"cuda:{local_rank}")

    train_fn = functools.partial(
        ppo.train, num_timesteps=config["num_timesteps"], num_evals=int(config["num_timesteps"] / config["eval_every"]),
        reward_scaling=1, episode_length=config["episode_length"], normalize_observations=True, action_repeat=1,
        unroll_length=10, num_minibatches=64, num_updates_per_batch=8, discounting=0.97,
        learning_rate=config["learning_rate"], entropy_cost=1e-3, num_envs=config["num_envs"],
        batch_size=config["batch_size"], seed=0, max_training_steps=args.max_training_steps)

    run_id = uuid.uuid4()
    model_path = f"./model_checkpoints/{run_id}"

    def progress(num_steps, metrics):
        metrics["num_steps"] = num_steps
        print(json.dumps({k: (float(v) if np.isscalar(v) else v) for k, v in metrics.items()}), flush=True)

    eval_env = env.with_num_envs(1)                  # the launcher's un-vmapped jit_reset / jit_step pair [REF :93-94]
    ref_clip = preprocessing.ReferenceClip(position=track_pos, quaternion=np.tile([1.0, 0, 0, 0], (len(track_pos), 1)),
                                           joints=np.zeros((len(track_pos), env.sys.nq - 7)))

    def policy_params_fn(num_steps, make_policy, params, model_path=model_path):
        """Checkpoint + the 500-step evaluation rollout paired with the reference clip [REF brax_rodent_run_ppo.py:135-191];
        the qpos pairs are saved instead of rendered (mujoco.Renderer / wandb are out of scope)."""
        os.makedirs(model_path, exist_ok=True)
        model.save_params(f"{model_path}/{num_steps}", params)
        qposes = rollout.eval_rollout(eval_env, make_policy, params, steps=500, seed=0)
        rollout.save_rollout(f"{model_path}/{num_steps}_rollout.npz", rollout.qpos_pairs(ref_clip, qposes), eval_env.dt, qposes)

    make_inference_fn, params, _ = train_fn(environment=env, progress_fn=progress, policy_params_fn=policy_params_fn)
    if int(os.environ.get("RANK", "0")) == 0:
        os.makedirs(model_path, exist_ok=True)
        model.save_params(f"{model_path}/brax_ppo_rodent_run_finished", params)
        print(f"Run finished. Model saved to {model_path}/brax_ppo_rodent_run_finished")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
