"""Reference-clip ingestion: the `ReferenceClip` container and the feature pass of the reference's
`preprocessing/mjx_preprocess.py` [REF preprocessing/mjx_preprocess.py:23-41 ReferenceClip, :93-161 process_clip /
extract_features, :195-222 compute_velocity_from_kinematics, :225-283 save / load] on the HIP backend.

* `extract_features` is forward kinematics only: the T frames of a clip are stepped as ONE batch of T "environments"
  through `rr_pipeline_init` (C ABI; the pose fields come from the kernel's dump), where the reference scans
  `set_position -> smooth.kinematics` over the frames.
* `compute_velocity_from_kinematics` restates the finite-difference rule (translation and joints: forward differences;
  angular velocity: axis-angle of conj(q_t) * q_{t+1}, angle wrapped to (-pi, pi], divided by dt) in numpy.
* Files: the reference writes HDF5 groups `clip_name/attribute` [REF :225-247]; h5py is not available in this image, so the
  same `clip_name/attribute` keys go into an .npz (used when the name does not end in .h5, or h5py is missing).  The
  reference's `.p` files are pickles of a flax dataclass of jax arrays [REF brax_rodent_run_ppo.py:61-77]: they are NOT loaded
  (unpickling executes code from the file); convert them to .npz / .h5 on a machine that has jax.
Out of scope (stated): the dm_control `rescale_subtree(root, 0.9, 0.9)` step of `process_clip_to_train` -- body positions
are those of the model the env was built with.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Union

import numpy as np

_TOL = 1e-10
FIELDS = ("position", "quaternion", "joints", "body_positions", "velocity", "joints_velocity", "angular_velocity", "body_quaternions")


@dataclasses.dataclass
class ReferenceClip:
    """This dataclass is used to store the trajectory in the env (same fields as the reference's)."""
    position: Optional[np.ndarray] = None          # qpos[:, :3]
    quaternion: Optional[np.ndarray] = None        # qpos[:, 3:7]
    joints: Optional[np.ndarray] = None            # qpos[:, 7:]
    body_positions: Optional[np.ndarray] = None    # xpos  [T, nbody, 3]
    velocity: Optional[np.ndarray] = None          # inferred
    joints_velocity: Optional[np.ndarray] = None
    angular_velocity: Optional[np.ndarray] = None
    body_quaternions: Optional[np.ndarray] = None  # xquat [T, nbody, 4]

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


def quat_mul(a, b):
    aw, ax, ay, az = np.moveaxis(np.asarray(a, np.float64), -1, 0)
    bw, bx, by, bz = np.moveaxis(np.asarray(b, np.float64), -1, 0)
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def quat_diff(source, target):
    s = np.asarray(source, np.float64) * np.array([1.0, -1.0, -1.0, -1.0])
    return quat_mul(s, target)


def quat_to_axisangle(quat):
    """Axis-angle vector (angle encoded by the length), angle wrapped to (-pi, pi]; batched over leading axes."""
    q = np.asarray(quat, np.float64)
    w = np.clip(q[..., 0], -1.0, 1.0)
    angle = 2 * np.arccos(w)
    qn = np.sin(angle / 2)
    wrapped = (angle + np.pi) % (2 * np.pi) - np.pi
    small = angle < _TOL
    axis = q[..., 1:4] / np.where(small, 1.0, qn)[..., None]
    return np.where(small[..., None], 0.0, axis * wrapped[..., None])


def compute_velocity_from_kinematics(qpos_trajectory, dt: float) -> np.ndarray:
    """[T, nq] -> [T-1, nv]; the first 7 columns are a free joint [REF preprocessing/mjx_preprocess.py:195-222]."""
    q = np.asarray(qpos_trajectory, np.float64)
    trans = (q[1:, :3] - q[:-1, :3]) / dt
    d = quat_diff(q[:-1, 3:7], q[1:, 3:7])
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    gyro = quat_to_axisangle(d) / dt
    joints = (q[1:, 7:] - q[:-1, 7:]) / dt
    return np.concatenate([trans, gyro, joints], axis=1)


def extract_features(env, mocap_qpos) -> ReferenceClip:
    """FK-only feature pass on the GPU: frame t of the clip is environment t of one `rr_pipeline_init` launch.
    `env`: a `Rodent` (any num_envs; a sibling batch of T envs is created).  Returns position / quaternion / joints /
    body_positions / body_quaternions as float32 numpy arrays [T, ...]."""
    import torch
    from . import hip
    q = np.ascontiguousarray(np.asarray(mocap_qpos, np.float32))
    T = q.shape[0]
    s = env.sys
    if q.shape[1] != s.nq:
        raise ValueError(f"clip has {q.shape[1]} qpos columns, the model {s.nq}")
    batch = hip.Batch(s.model, T, env.device)
    st = dict(qpos=torch.from_numpy(q).to(env.device), qvel=torch.zeros(T, s.nv, device=env.device),
              act=torch.zeros(T, s.na, device=env.device), qacc_warmstart=torch.zeros(T, s.nv, device=env.device))
    dbg = torch.zeros(T, batch.dims.dbg_floats, device=env.device)
    batch.pipeline_init(st, out=dict(debug=dbg))
    lay = batch.debug_layout()
    g = dbg.cpu().numpy()
    f = lambda name, w: g[:, lay[name][0]:lay[name][0] + lay[name][1]].reshape(T, -1, w)
    xquat = f("xquat", 4).copy()
    xquat[:, 0] = [1, 0, 0, 0]                        # world body
    return ReferenceClip(position=q[:, :3].copy(), quaternion=q[:, 3:7].copy(), joints=q[:, 7:].copy(),
                         body_positions=f("xpos", 3).copy(), body_quaternions=xquat)


def process_clip(mocap_qpos, env, max_qvel: float = 20.0, dt: float = 0.02) -> ReferenceClip:
    """Joint angles -> the features the reference trajectory is composed of [REF preprocessing/mjx_preprocess.py:93-134]."""
    clip = extract_features(env, mocap_qpos)
    q = np.asarray(mocap_qpos, np.float64)
    q = np.concatenate([q, q[-1:]], axis=0)           # padding for the velocity corner case
    qvel = compute_velocity_from_kinematics(q, dt)
    qvel[:, 6:] = np.clip(qvel[:, 6:], -max_qvel, max_qvel)
    return clip.replace(velocity=qvel[:, :3].astype(np.float32), angular_velocity=qvel[:, 3:6].astype(np.float32),
                        joints_velocity=qvel[:, 6:].astype(np.float32))


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def save_reference_clip(filename: str, clip_names: Union[List[str], str], reference_clip: ReferenceClip):
    """`clip_name/attribute` datasets, single clip (str) or a stacked multi-clip (list) [REF :225-247]."""
    assert isinstance(clip_names, (str, list))
    items = {}
    for attr in FIELDS:
        value = getattr(reference_clip, attr)
        if value is None:
            continue
        if isinstance(clip_names, str):
            items[f"{clip_names}/{attr}"] = np.asarray(value)
        else:
            for i, name in enumerate(clip_names):
                items[f"{name}/{attr}"] = np.asarray(value)[i]
    h5 = _h5py()
    if filename.endswith(".h5") and h5 is not None:
        with h5.File(filename, "w") as hf:
            for k, v in items.items():
                hf.create_dataset(k, data=v)
    else:
        np.savez(filename, **items)


def load_reference_clip(filename: str, clip_names: Union[List[str], str]) -> ReferenceClip:
    """Stacks the named clips in the given order [REF :250-283]; .npz (this build) or .h5 (when h5py is importable)."""
    if filename.endswith((".p", ".pkl", ".pickle")):
        raise ValueError("pickled reference clips are not loaded (unpickling executes code from the file); convert to .npz / .h5")
    if isinstance(clip_names, str):
        clip_names = [clip_names]
    if filename.endswith(".h5"):
        h5 = _h5py()
        if h5 is None:
            raise RuntimeError("h5py is not available: save the clip as .npz (same clip_name/attribute keys)")
        with h5.File(filename, "r") as hf:
            data = {f"{c}/{a}": hf[f"{c}/{a}"][:] for c in clip_names for a in FIELDS if f"{c}/{a}" in hf}
    else:
        with np.load(filename if filename.endswith(".npz") else filename + ".npz", allow_pickle=False) as z:
            data = {k: z[k] for k in z.files}
    agg = {}
    for a in FIELDS:
        vals = [data[f"{c}/{a}"] for c in clip_names if f"{c}/{a}" in data]
        if vals:
            agg[a] = np.stack(vals)
    return ReferenceClip(**agg)
