"""Evaluation rollout of the launcher's `policy_params_fn` [REF brax_rodent_run_ppo.py:135-197]: a single (un-batched) env is
reset, driven for 500 steps by the deterministic policy, and the rollout's qpos is paired frame by frame with the
reference clip's qpos -- the `np.append(qpos1, qpos2)` vectors the reference feeds to `rodent_pair.xml` for rendering
(SURVEY.md a6, 8(f)-3).  The renderer itself (mujoco.Renderer, imageio, wandb) is observability and out of scope: the pairs
are returned / saved as .npz; `pair_poses` runs the pair model's forward pass on them (what `mj_forward` provides the renderer).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import jax_random


@torch.no_grad()
def eval_rollout(env, make_policy, params, steps: int = 500, seed: int = 0, record: Optional[list] = None) -> np.ndarray:
    """`env`: a `Rodent` with num_envs = 1 (the reference's jit_reset / jit_step pair).  Key handling as the launcher:
    `key = PRNGKey(seed); _, key = split(key); reset_rng, act_rng = split(key)`; the policy is deterministic.
    Returns the rollout's qpos [steps + 1, nq] (float32).  `record` (tests): a list that receives (state, action, next_state) per step."""
    if env.num_envs != 1:
        raise ValueError("the evaluation rollout steps a single env (use env.with_num_envs(1))")
    key = jax_random.PRNGKey(seed)
    _, key = jax_random.split(key)
    reset_rng, act_rng = jax_random.split(key)
    policy = make_policy(params, deterministic=True)
    state = env.reset(reset_rng[None])
    qposes = [state.pipeline_state.qpos[0].clone()]
    for _ in range(steps):
        _, act_rng = jax_random.split(act_rng)
        ctrl, _ = policy(state.obs, None)
        prev, state = state, env.step(state, ctrl)
        if record is not None:
            record.append((prev, ctrl, state))
        qposes.append(state.pipeline_state.qpos[0].clone())
    return torch.stack(qposes).cpu().numpy()


def qpos_pairs(reference_clip, qposes_rollout: np.ndarray, clip_length: int = 250) -> np.ndarray:
    """[min(T_ref, T_rollout), 2 nq]: reference qpos (position | quaternion | joints, first `clip_length` frames) next to the
    rollout's, as `zip(qposes_ref, qposes_rollout)` + `np.append` do in the launcher."""
    ref = np.hstack([np.asarray(reference_clip.position)[:clip_length], np.asarray(reference_clip.quaternion)[:clip_length],
                     np.asarray(reference_clip.joints)[:clip_length]])
    n = min(len(ref), len(qposes_rollout))
    return np.concatenate([ref[:n], qposes_rollout[:n]], axis=1).astype(np.float32)


def pair_poses(pairs: np.ndarray, device="cuda:0", pair_model: str = "rodent_pair.xml"):
    """Forward pass of the two-rodent model on the paired qpos (all frames in one launch): body positions [T, nbody, 3] and
    rotation matrices [T, nbody, 9] -- the scene the reference renders frame by frame with `mj_forward`."""
    from . import assets, hip
    model = hip.Model(assets.resolve_model(pair_model))
    T = pairs.shape[0]
    b = hip.Batch(model, T, torch.device(device))
    d = b.dims
    if pairs.shape[1] != d.nq:
        raise ValueError(f"pairs have {pairs.shape[1]} columns, the pair model {d.nq} qpos")
    st = dict(qpos=torch.tensor(pairs, dtype=torch.float32, device=device), qvel=torch.zeros(T, d.nv, device=device),
              act=torch.zeros(T, d.na, device=device), qacc_warmstart=torch.zeros(T, d.nv, device=device))
    out = dict(xpos=torch.empty(T, d.nbody * 3, device=device), xmat=torch.empty(T, d.nbody * 9, device=device))
    b.pipeline_init(st, out)
    return out["xpos"].reshape(T, d.nbody, 3), out["xmat"].reshape(T, d.nbody, 9)


def save_rollout(path: str, pairs: np.ndarray, dt: float, qposes_rollout: Optional[np.ndarray] = None):
    np.savez(path, qpos_pairs=pairs, dt=np.float32(dt), **({"qposes_rollout": qposes_rollout} if qposes_rollout is not None else {}))
