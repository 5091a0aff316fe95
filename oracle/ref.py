"""ctypes wrapper around the CPU oracle (oracle/rodent_ref.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib(precision: str):
    if precision not in _LIBS:
        path = os.path.join(_HERE, "_build", f"librodent_ref_{precision}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.ref_model_load.restype = C.c_void_p
        lib.ref_model_load.argtypes = [C.c_char_p]
        lib.ref_model_dim.restype = C.c_int
        lib.ref_model_dim.argtypes = [C.c_void_p, C.c_char_p]
        lib.ref_model_set_iterations.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.ref_model_set_solver.argtypes = [C.c_void_p, C.c_int]
        lib.ref_data_new.restype = C.c_void_p
        lib.ref_data_new.argtypes = [C.c_void_p]
        lib.ref_data_free.argtypes = [C.c_void_p]
        dp = C.POINTER(C.c_double)
        lib.ref_init.argtypes = [C.c_void_p, C.c_void_p, dp, dp]
        lib.ref_forward.argtypes = [C.c_void_p, C.c_void_p]
        lib.ref_step.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_int]
        lib.ref_get.restype = C.c_long
        lib.ref_get.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, dp, C.c_long]
        lib.ref_set.restype = C.c_long
        lib.ref_set.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, dp, C.c_long]
        lib.ref_get_obs.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_int, C.c_int, dp]
        lib.ref_env_step.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_int, dp, C.c_int, C.POINTER(C.c_int),
                                     C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, dp, dp]
        lib.ref_step_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), dp, C.c_int, C.c_int]
        lib.ref_env_step_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), dp, C.c_int, C.c_int, dp, C.c_int, C.POINTER(C.c_int),
                                           C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, dp, dp]
        lib.ref_sig_reset.argtypes = [C.c_void_p]
        lib.ref_sig_get.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _LIBS[precision] = lib
    return _LIBS[precision]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class RefModel:
    def __init__(self, blob_path: str, precision: str = "f64"):
        self.lib = _lib(precision)
        self.h = self.lib.ref_model_load(blob_path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load model blob {blob_path}")
        for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nM", "ncon", "nlimit", "nefc", "obs_dim"):
            setattr(self, k, self.lib.ref_model_dim(self.h, k.encode()))

    def set_solver(self, solver: str):
        """'cg' or 'newton' [REF Rodent_Env_Brax.py:42-45]"""
        self.lib.ref_model_set_solver(self.h, {"cg": 1, "newton": 2}[solver.lower()])

    def set_iterations(self, iterations, ls_iterations):
        self.lib.ref_model_set_iterations(self.h, iterations, ls_iterations)


class RefData:
    """One environment of the oracle."""

    def __init__(self, model: RefModel):
        self.m = model
        self.lib = model.lib
        self.h = self.lib.ref_data_new(model.h)

    def __del__(self):
        try:
            self.lib.ref_data_free(self.h)
        except Exception:
            pass

    def init(self, qpos, qvel):
        qpos = np.ascontiguousarray(qpos, np.float64)
        qvel = np.ascontiguousarray(qvel, np.float64)
        assert qpos.size == self.m.nq and qvel.size == self.m.nv
        self.lib.ref_init(self.m.h, self.h, _dp(qpos), _dp(qvel))

    def forward(self):
        self.lib.ref_forward(self.m.h, self.h)

    def step(self, ctrl, n_frames=1):
        ctrl = np.ascontiguousarray(ctrl, np.float64)
        assert ctrl.size == self.m.nu
        self.lib.ref_step(self.m.h, self.h, _dp(ctrl), n_frames)

    def get(self, name):
        n = self.lib.ref_get(self.m.h, self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(n, np.float64)
        self.lib.ref_get(self.m.h, self.h, name.encode(), _dp(out), n)
        return out

    def set(self, name, val):
        val = np.ascontiguousarray(val, np.float64).ravel()
        n = self.lib.ref_set(self.m.h, self.h, name.encode(), _dp(val), val.size)
        if n < 0:
            raise KeyError(name)

    def sig_reset(self):
        self.lib.ref_sig_reset(self.h)

    def sig(self):
        """(hash of the contact/limit active sets of every forward pass since sig_reset, hash of the solver's final row
        states, summed CG iterations, number of forward passes): the discrete decisions a trajectory took."""
        out = (C.c_uint64 * 4)()
        self.lib.ref_sig_get(self.h, out)
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def obs(self, track_pos, cur_frame):
        tp = np.ascontiguousarray(track_pos, np.float64)
        out = np.zeros(self.m.obs_dim, np.float64)
        self.lib.ref_get_obs(self.m.h, self.h, _dp(tp), tp.shape[0], int(cur_frame), _dp(out))
        return out

    def env_step(self, action, track_pos, cur_frame, n_frames=10, healthy_reward=1.0, ctrl_cost_weight=0.1,
                 healthy_z_range=(0.03, 0.5), terminate_when_unhealthy=True):
        action = np.ascontiguousarray(action, np.float64)
        tp = np.ascontiguousarray(track_pos, np.float64)
        obs = np.zeros(self.m.obs_dim, np.float64)
        rew, done = C.c_double(), C.c_double()
        metrics = np.zeros(3, np.float64)
        cf = C.c_int(int(cur_frame))
        self.lib.ref_env_step(self.m.h, self.h, _dp(action), n_frames, _dp(tp), tp.shape[0], C.byref(cf),
                              healthy_reward, ctrl_cost_weight, healthy_z_range[0], healthy_z_range[1],
                              int(terminate_when_unhealthy), _dp(obs), C.byref(rew), C.byref(done), _dp(metrics))
        return obs, rew.value, done.value, cf.value, metrics


def step_batch(model: RefModel, datas, ctrl, n_frames=10):
    """OpenMP-parallel pipeline_step over a list of RefData (cpu_baseline timing)."""
    ctrl = np.ascontiguousarray(ctrl, np.float64)
    arr = (C.c_void_p * len(datas))(*[d.h for d in datas])
    model.lib.ref_step_batch(model.h, arr, _dp(ctrl), len(datas), n_frames)


class RefBatch:
    """N oracle environments stepped together (OpenMP over envs): the CPU counterpart of `Rodent` for the long parity runs."""
    STATE = ("qpos", "qvel", "act", "qacc_warmstart")

    def __init__(self, model: RefModel, n: int):
        self.m, self.n = model, n
        self.d = [RefData(model) for _ in range(n)]
        self._arr = (C.c_void_p * n)(*[d.h for d in self.d])

    def init(self, qpos, qvel):
        for e, d in enumerate(self.d):
            d.init(qpos[e], qvel[e])

    def get(self, name):
        return np.stack([d.get(name) for d in self.d])

    def set_state(self, st, envs=None):
        for e in (range(self.n) if envs is None else envs):
            for k in self.STATE:
                self.d[e].set(k, st[k][e])

    def state(self):
        return {k: self.get(k) for k in self.STATE}

    def obs(self, track_pos, cur_frame):
        return np.stack([d.obs(track_pos, int(cur_frame[e])) for e, d in enumerate(self.d)])

    def sig_reset(self):
        for d in self.d:
            d.sig_reset()

    def sigs(self):
        return [d.sig() for d in self.d]

    def env_step(self, action, track_pos, cur_frame, n_frames=10, healthy_reward=1.0, ctrl_cost_weight=0.1,
                 healthy_z_range=(0.03, 0.5), terminate_when_unhealthy=True):
        """-> obs [N, obs_dim], reward [N], done [N], cur_frame [N] (int32, new), metrics [N, 3]"""
        m, n = self.m, self.n
        action = np.ascontiguousarray(action, np.float64).reshape(n, m.nu)
        tp = np.ascontiguousarray(track_pos, np.float64)
        cf = np.ascontiguousarray(cur_frame, np.int32).copy()
        obs = np.zeros((n, m.obs_dim)); rew = np.zeros(n); done = np.zeros(n); met = np.zeros((n, 3))
        m.lib.ref_env_step_batch(m.h, self._arr, _dp(action), n, n_frames, _dp(tp), tp.shape[0],
                                 cf.ctypes.data_as(C.POINTER(C.c_int)), healthy_reward, ctrl_cost_weight, healthy_z_range[0],
                                 healthy_z_range[1], int(terminate_when_unhealthy), _dp(obs), _dp(rew), _dp(done), _dp(met))
        return obs, rew, done, cf, met
