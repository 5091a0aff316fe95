"""CPU: ReferenceClip container, velocity inference and clip files (rodent_amd/preprocessing.py) against closed-form cases
[REF preprocessing/mjx_preprocess.py:195-283]."""
import numpy as np
import pytest

from rodent_amd import preprocessing as pp


def _traj(T=12, dt=0.02, w=np.array([0.3, -1.1, 0.7]), v=np.array([0.2, 0.0, -0.1])):
    """Constant world-frame linear velocity, constant BODY-frame angular velocity, linearly moving joints."""
    q = np.zeros((T, 74))
    quat = np.array([1.0, 0, 0, 0])
    ang = np.linalg.norm(w) * dt
    dq = np.concatenate([[np.cos(ang / 2)], w / np.linalg.norm(w) * np.sin(ang / 2)])
    for t in range(T):
        q[t, :3] = v * dt * t
        q[t, 3:7] = quat
        q[t, 7:] = 0.01 * t * np.arange(67) / 67
        quat = pp.quat_mul(quat, dq)
    return q


def test_velocity_from_kinematics_closed_form():
    dt, w, v = 0.02, np.array([0.3, -1.1, 0.7]), np.array([0.2, 0.0, -0.1])
    q = _traj(dt=dt, w=w, v=v)
    qvel = pp.compute_velocity_from_kinematics(q, dt)
    assert qvel.shape == (11, 73)
    np.testing.assert_allclose(qvel[:, :3], np.tile(v, (11, 1)), atol=1e-12)
    np.testing.assert_allclose(qvel[:, 3:6], np.tile(w, (11, 1)), atol=1e-9)           # conj(q_t) * q_{t+1}: body-frame rate
    np.testing.assert_allclose(qvel[:, 6:], np.tile(0.01 * np.arange(67) / 67 / dt, (11, 1)), atol=1e-12)
    # identical frames -> zero rotation (the angle < tol branch), no NaN
    same = np.repeat(q[:1], 3, axis=0)
    assert np.array_equal(pp.compute_velocity_from_kinematics(same, dt), np.zeros((2, 73)))
    # angle wrap: a rotation of pi + 0.2 about z comes back as -(pi - 0.2)
    a = np.pi + 0.2
    aa = pp.quat_to_axisangle(np.array([np.cos(a / 2), 0, 0, np.sin(a / 2)]))
    np.testing.assert_allclose(aa, [0, 0, -(np.pi - 0.2)], atol=1e-12)


def test_clip_files_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    clip = pp.ReferenceClip(**{f: rng.normal(size=(5, 3)).astype(np.float32) for f in pp.FIELDS})
    p = str(tmp_path / "one.npz")
    pp.save_reference_clip(p, "clip_84", clip)
    back = pp.load_reference_clip(p, "clip_84")
    for f in pp.FIELDS:
        np.testing.assert_array_equal(getattr(back, f), getattr(clip, f)[None])          # a leading clip axis, as the reference stacks
    multi = pp.ReferenceClip(**{f: rng.normal(size=(2, 5, 3)).astype(np.float32) for f in pp.FIELDS})
    p2 = str(tmp_path / "two.npz")
    pp.save_reference_clip(p2, ["a", "b"], multi)
    back = pp.load_reference_clip(p2, ["b", "a"])                                        # order follows the request
    np.testing.assert_array_equal(back.position, multi.position[::-1])
    with pytest.raises(ValueError, match="pickled"):
        pp.load_reference_clip("clips/84.p", "x")
    assert pp.ReferenceClip().replace(position=np.zeros(3)).position.shape == (3,)
