"""ctypes binding of librodent_hip.so (C ABI: include/rodent_rr.h).

PyTorch is used only for device memory and streams: every tensor handed to the library is a
contiguous float32 / int32 CUDA(HIP) tensor whose `data_ptr()` crosses the ABI.  There is no CPU
fallback: if the shared library is missing or no GPU is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RR_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "librodent_hip.so")


class RRDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nM", "ncon", "nlimit",
                                         "nefc", "obs_dim", "iterations", "ls_iterations", "lds_bytes", "dbg_floats")] \
        + [("timestep", C.c_float), ("solver", C.c_int32), ("fixed_instance", C.c_int32)]


class RRState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qpos", "qvel", "act", "qacc_warmstart")]


class RROutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cinert", "cvel", "qfrc_actuator", "xpos", "xmat", "subtree_com", "debug",
                                                "contact_dist", "contact_pos", "contact_frame")]


class RRMlpNet(C.Structure):
    _fields_ = [("weights", C.POINTER(C.c_void_p)), ("biases", C.POINTER(C.c_void_p)), ("sizes", C.POINTER(C.c_int32)), ("nlayers", C.c_int32)]


class RREnvIO(C.Structure):
    _fields_ = [("track_pos", C.c_void_p), ("track_len", C.c_int32), ("cur_frame", C.c_void_p), ("obs", C.c_void_p),
                ("reward", C.c_void_p), ("done", C.c_void_p), ("metrics", C.c_void_p), ("healthy_reward", C.c_float),
                ("ctrl_cost_weight", C.c_float), ("healthy_z_min", C.c_float), ("healthy_z_max", C.c_float),
                ("terminate_when_unhealthy", C.c_int32)]


class RRUnrollIO(C.Structure):
    _fields_ = [("first", RRState), ("first_obs", C.c_void_p), ("prev_done", C.c_void_p), ("steps_in", C.c_void_p), ("steps_out", C.c_void_p),
                ("truncation_out", C.c_void_p), ("episode_length", C.c_float)]


class RRActorIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs_in", "mean", "std", "w0", "b0")] + [("hidden_wt", C.c_void_p * 4), ("hidden_b", C.c_void_p * 4)] + \
        [(n, C.c_void_p) for n in ("head_wt", "head_b", "noise", "actions_out", "traj_obs", "traj_raw_action", "traj_log_prob", "traj_reward",
                                   "traj_discount", "traj_truncation")] + [("min_std", C.c_float), ("nhidden", C.c_int32), ("segment_length", C.c_int32)]


class RRDwItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("delta", "act", "act_rows", "mean", "std", "delta_colsum")] + \
        [("M", C.c_int32), ("O", C.c_int32), ("I", C.c_int32), ("grad", C.c_void_p)]


class RRPpoCfg(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("entropy_cost", "discounting", "reward_scaling", "gae_lambda", "clipping_epsilon", "min_std")] + \
        [("normalize_advantage", C.c_int32)]


EXPORTS = ["rr_model_load", "rr_model_dims", "rr_model_set_solver", "rr_model_set_solver_type", "rr_model_destroy", "rr_model_table", "rr_kernarg_layout", "rr_batch_create",
           "rr_batch_destroy", "rr_pipeline_init", "rr_pipeline_step", "rr_env_step", "rr_env_reset", "rr_pipeline_step_to", "rr_env_step_to", "rr_batch_contact_overflow", "rr_batch_unroll_supported", "rr_env_unroll", "rr_env_unroll_policy",
           "rr_compute_gae", "rr_mlp_forward", "rr_ppo_loss_workspace_bytes", "rr_ppo_loss", "rr_policy_act_workspace_bytes", "rr_policy_act", "rr_policy_sample", "rr_policy_backward_workspace_bytes", "rr_policy_backward", "rr_mlp_silu_backward_workspace_bytes", "rr_mlp_silu_backward", "rr_mlp_value_backward_workspace_bytes", "rr_mlp_value_backward", "rr_mlp_weight_grad_workspace_bytes", "rr_mlp_weight_grad", "rr_mlp_weight_grad_batch_workspace_bytes", "rr_mlp_weight_grad_batch", "rr_obs_moments_workspace_bytes", "rr_obs_moments", "rr_wrap_episode_autoreset", "rr_debug_layout", "rr_batch_set_schedule", "rr_batch_set_profile", "rr_batch_set_timing", "rr_batch_kernel_time", "rr_last_error"]

_lib = None


def lib():
    """Load librodent_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.rr_last_error.restype = C.c_char_p
        L.rr_model_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.rr_model_dims.argtypes = [C.c_void_p, C.POINTER(RRDims)]
        L.rr_model_set_solver.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.rr_model_set_solver_type.argtypes = [C.c_void_p, C.c_int32]
        L.rr_model_destroy.argtypes = [C.c_void_p]
        L.rr_model_destroy.restype = None
        L.rr_model_table.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
        L.rr_kernarg_layout.argtypes = [C.POINTER(C.c_int32)] * 3
        L.rr_batch_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rr_batch_destroy.argtypes = [C.c_void_p]
        L.rr_batch_destroy.restype = None
        L.rr_pipeline_init.argtypes = [C.c_void_p, C.POINTER(RRState), C.POINTER(RROutputs)]
        L.rr_pipeline_step.argtypes = [C.c_void_p, C.POINTER(RRState), C.c_void_p, C.c_int32, C.POINTER(RROutputs)]
        L.rr_env_step.argtypes = [C.c_void_p, C.POINTER(RRState), C.c_void_p, C.c_int32, C.POINTER(RREnvIO),
                                  C.POINTER(RROutputs)]
        L.rr_pipeline_step_to.argtypes = [C.c_void_p, C.POINTER(RRState), C.POINTER(RRState), C.c_void_p, C.c_int32, C.POINTER(RROutputs)]
        L.rr_env_step_to.argtypes = [C.c_void_p, C.POINTER(RRState), C.POINTER(RRState), C.c_void_p, C.c_int32, C.POINTER(RREnvIO),
                                     C.c_void_p, C.POINTER(RROutputs)]
        L.rr_batch_unroll_supported.argtypes = [C.c_void_p, C.c_int32]
        L.rr_batch_contact_overflow.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.rr_env_unroll.argtypes = [C.c_void_p, C.POINTER(RRState), C.POINTER(RRState), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(RREnvIO),
                                    C.c_void_p, C.POINTER(RRUnrollIO)]
        L.rr_env_unroll_policy.argtypes = [C.c_void_p, C.POINTER(RRState), C.POINTER(RRState), C.c_int32, C.c_int32, C.POINTER(RREnvIO), C.c_void_p,
                                           C.POINTER(RRUnrollIO), C.POINTER(RRActorIO)]
        L.rr_env_reset.argtypes = [C.c_void_p, C.POINTER(RRState), C.POINTER(RREnvIO), C.POINTER(RROutputs)]
        L.rr_debug_layout.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_int32)),
                                      C.POINTER(C.POINTER(C.c_int32))]
        L.rr_compute_gae.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rr_mlp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(RRMlpNet), C.POINTER(RRMlpNet),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rr_ppo_loss_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
        L.rr_ppo_loss_workspace_bytes.restype = C.c_size_t
        L.rr_ppo_loss.argtypes = [C.c_void_p] * 9 + [C.c_int32] * 3 + [C.POINTER(RRPpoCfg)] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
        L.rr_policy_act_workspace_bytes.argtypes = [C.c_int32]
        L.rr_policy_act_workspace_bytes.restype = C.c_size_t
        L.rr_policy_act.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(RRMlpNet), C.c_void_p, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_policy_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rr_policy_backward_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
        L.rr_policy_backward_workspace_bytes.restype = C.c_size_t
        L.rr_policy_backward.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_mlp_silu_backward_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
        L.rr_mlp_silu_backward_workspace_bytes.restype = C.c_size_t
        L.rr_mlp_silu_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_mlp_value_backward_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
        L.rr_mlp_value_backward_workspace_bytes.restype = C.c_size_t
        L.rr_mlp_value_backward.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_mlp_weight_grad_workspace_bytes.argtypes = [C.c_int32] * 3
        L.rr_mlp_weight_grad_workspace_bytes.restype = C.c_size_t
        L.rr_mlp_weight_grad.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_mlp_weight_grad_batch_workspace_bytes.argtypes = [C.POINTER(RRDwItem), C.c_int32]
        L.rr_mlp_weight_grad_batch_workspace_bytes.restype = C.c_size_t
        L.rr_mlp_weight_grad_batch.argtypes = [C.POINTER(RRDwItem), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_wrap_episode_autoreset.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)] + \
            [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p]
        L.rr_obs_moments_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        L.rr_obs_moments_workspace_bytes.restype = C.c_size_t
        L.rr_obs_moments.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rr_batch_set_profile.argtypes = [C.c_void_p, C.c_void_p]
        L.rr_batch_set_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rr_batch_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.rr_batch_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise RuntimeError(f"librodent_hip: {lib().rr_last_error().decode()} (status {rc})")
    return rc


def _ptr(t: Optional[torch.Tensor], dtype=torch.float32, numel: Optional[int] = None):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"expected a contiguous {dtype} device tensor, got {t.dtype} on {t.device}")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"tensor has {t.numel()} elements, expected {numel}")
    return t.data_ptr()


class Model:
    """A compiled model loaded from an RRM1 blob (host side)."""

    def __init__(self, blob_path: str, iterations: Optional[int] = None, ls_iterations: Optional[int] = None, solver: Optional[str] = None):
        self.h = C.c_void_p()
        _check(lib().rr_model_load(blob_path.encode(), C.byref(self.h)))
        if solver is not None:
            _check(lib().rr_model_set_solver_type(self.h, {"cg": 1, "newton": 2}[solver.lower()]))
        if iterations is not None:
            d = self.dims
            _check(lib().rr_model_set_solver(self.h, iterations, ls_iterations if ls_iterations is not None else d.ls_iterations))

    def table(self, name: str):
        """A static table of the LOADED model by its MuJoCo field name (C ABI `rr_model_table`) as a numpy copy."""
        import numpy as np
        ptr, cnt, dt = C.c_void_p(), C.c_size_t(), C.c_int32()
        _check(lib().rr_model_table(self.h, name.encode(), C.byref(ptr), C.byref(cnt), C.byref(dt)))
        ctype = C.c_float if dt.value == 0 else C.c_int32
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt.value,)).copy()

    @property
    def dims(self) -> RRDims:
        d = RRDims()
        _check(lib().rr_model_dims(self.h, C.byref(d)))
        return d

    def __del__(self):
        try:
            if self.h:
                lib().rr_model_destroy(self.h)
        except Exception:
            pass


class Batch:
    """`num_envs` environments of one model on one GPU.  All launches go to the stream that was torch's current stream on
    `device` WHEN THE BATCH WAS CREATED (the C ABI binds the stream at rr_batch_create); create the batch inside the
    `torch.cuda.stream(...)` context it should run on."""

    def __init__(self, model: Model, num_envs: int, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("rodent_amd.hip.Batch needs a GPU (no CPU fallback)")
        self.model = model
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.N = int(num_envs)
        self.dims = model.dims
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
        _check(lib().rr_batch_create(model.h, self.N, self.device.index or 0, C.c_void_p(stream), C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().rr_batch_destroy(self.h)
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def zeros_state(self) -> Dict[str, torch.Tensor]:
        d, N, dev = self.dims, self.N, self.device
        return dict(qpos=torch.zeros(N, d.nq, device=dev), qvel=torch.zeros(N, d.nv, device=dev),
                    act=torch.zeros(N, d.na, device=dev), qacc_warmstart=torch.zeros(N, d.nv, device=dev))

    def _state(self, st) -> RRState:
        d, N = self.dims, self.N
        return RRState(_ptr(st["qpos"], numel=N * d.nq), _ptr(st["qvel"], numel=N * d.nv),
                       _ptr(st["act"], numel=N * d.na), _ptr(st["qacc_warmstart"], numel=N * d.nv))

    def _outputs(self, out) -> Optional[RROutputs]:
        if not out:
            return None
        d, N = self.dims, self.N
        sizes = dict(cinert=10 * d.nbody, cvel=6 * d.nbody, qfrc_actuator=d.nv, xpos=3 * d.nbody, xmat=9 * d.nbody,
                     subtree_com=3, debug=d.dbg_floats, contact_dist=d.ncon, contact_pos=3 * d.ncon, contact_frame=9 * d.ncon)
        unknown = set(out) - set(sizes)
        if unknown:
            raise ValueError(f"unknown rr_outputs field(s): {sorted(unknown)}")
        o = RROutputs()
        for k, w in sizes.items():
            setattr(o, k, _ptr(out.get(k), numel=N * w) if out.get(k) is not None else None)
        return o

    def _env(self, env) -> RREnvIO:
        d, N = self.dims, self.N
        tp = env["track_pos"]
        e = RREnvIO()
        e.track_pos = _ptr(tp)
        e.track_len = tp.shape[0]
        e.cur_frame = _ptr(env["cur_frame"], torch.int32, N)
        e.obs = _ptr(env["obs"], numel=N * d.obs_dim)
        e.reward = _ptr(env.get("reward"), numel=N) if env.get("reward") is not None else None
        e.done = _ptr(env.get("done"), numel=N) if env.get("done") is not None else None
        e.metrics = _ptr(env.get("metrics"), numel=3 * N) if env.get("metrics") is not None else None
        e.healthy_reward = env.get("healthy_reward", 1.0)
        e.ctrl_cost_weight = env.get("ctrl_cost_weight", 0.1)
        e.healthy_z_min, e.healthy_z_max = env.get("healthy_z_range", (0.03, 0.5))
        e.terminate_when_unhealthy = int(env.get("terminate_when_unhealthy", True))
        return e

    # ------------------------------------------------------------------ C-ABI calls
    def pipeline_init(self, st, out=None):
        o = self._outputs(out)
        _check(lib().rr_pipeline_init(self.h, C.byref(self._state(st)), C.byref(o) if o else None))

    def pipeline_step(self, st, ctrl, n_frames: int, out=None):
        o = self._outputs(out)
        _check(lib().rr_pipeline_step(self.h, C.byref(self._state(st)), _ptr(ctrl, numel=self.N * self.dims.nu),
                                      int(n_frames), C.byref(o) if o else None))

    def env_step(self, st, action, n_frames: int, env, out=None):
        o = self._outputs(out)
        _check(lib().rr_env_step(self.h, C.byref(self._state(st)), _ptr(action, numel=self.N * self.dims.nu),
                                 int(n_frames), C.byref(self._env(env)), C.byref(o) if o else None))

    def pipeline_step_to(self, st_in, st_out, ctrl, n_frames: int, out=None):
        """Out-of-place pipeline_step: reads st_in, writes st_out (no copies of the previous state needed)."""
        o = self._outputs(out)
        _check(lib().rr_pipeline_step_to(self.h, C.byref(self._state(st_in)), C.byref(self._state(st_out)),
                                         _ptr(ctrl, numel=self.N * self.dims.nu), int(n_frames), C.byref(o) if o else None))

    def env_step_to(self, st_in, st_out, action, n_frames: int, env, cur_frame_in, out=None):
        o = self._outputs(out)
        _check(lib().rr_env_step_to(self.h, C.byref(self._state(st_in)), C.byref(self._state(st_out)),
                                    _ptr(action, numel=self.N * self.dims.nu), int(n_frames), C.byref(self._env(env)),
                                    _ptr(cur_frame_in, torch.int32, self.N), C.byref(o) if o else None))

    def contact_overflow(self) -> int:
        """(launch, env) events with more pairs in penetration than contact slots (candidate-pair models; synchronises the stream)."""
        n = C.c_int64()
        _check(lib().rr_batch_contact_overflow(self.h, C.byref(n)))
        return int(n.value)

    def unroll_supported(self, with_actor: bool = False) -> bool:
        return _check(lib().rr_batch_unroll_supported(self.h, int(with_actor))) == 1

    def env_unroll(self, st_in, st_out, actions, n_frames: int, env, cur_frame_in, first, first_obs, prev_done, steps_in, steps_out,
                   truncation_out, episode_length: float):
        """`actions.shape[0]` env steps with the Episode + AutoReset wrappers in one launch (C ABI `rr_env_unroll`)."""
        T = actions.shape[0]
        for t in (first_obs, prev_done, steps_in, steps_out, truncation_out):
            _ptr(t)
        w = RRUnrollIO(self._state(first), first_obs.data_ptr(), prev_done.data_ptr(), steps_in.data_ptr(), steps_out.data_ptr(),
                       truncation_out.data_ptr(), float(episode_length))
        _check(lib().rr_env_unroll(self.h, C.byref(self._state(st_in)), C.byref(self._state(st_out)),
                                   _ptr(actions, numel=T * self.N * self.dims.nu), T, int(n_frames), C.byref(self._env(env)),
                                   _ptr(cur_frame_in, torch.int32, self.N), C.byref(w)))

    def env_unroll_policy(self, st_in, st_out, T: int, n_frames: int, env, cur_frame_in, first, first_obs, prev_done, steps_in, steps_out,
                          truncation_out, episode_length: float, actor: dict, noise, actions_out, traj: dict, obs_in, segment: int = 0):
        """T x [policy -> sample -> wrapped env step] with the transitions recorded, one launch (C ABI `rr_env_unroll_policy`).
        actor: mean, std (or None), w0, b0, hidden_wt / hidden_b (lists, transposed weights), head_wt, head_b (padded), min_std;
        traj: obs [N, T+1, K], raw_action [N, T, A], log_prob / reward / discount / truncation [N, T] (contiguous views); with
        `segment` = L < T: U = T / L such blocks ([U, N, L+1, K], ...), a whole rollout phase of U unrolls."""
        for t in (first_obs, prev_done, steps_in, steps_out, truncation_out, noise, actions_out, obs_in, actor["w0"], actor["b0"],
                  actor["head_wt"], actor["head_b"], *actor["hidden_wt"], *actor["hidden_b"], *traj.values()):
            _ptr(t)
        nh = len(actor["hidden_wt"]) + 1
        A_, K = self.dims.nu, self.dims.obs_dim
        L_ = segment or T
        if T % L_:
            raise ValueError("rr_env_unroll_policy: the number of steps must be a multiple of the segment length")
        if (noise.numel() != T * self.N * A_ or actions_out.numel() != T * self.N * A_ or traj["obs"].numel() != (T // L_) * self.N * (L_ + 1) * K
                or traj["raw_action"].numel() != self.N * T * A_ or any(traj[k].numel() != self.N * T for k in ("log_prob", "reward", "discount", "truncation"))
                or actor["w0"].shape != (32, K) or actor["head_wt"].shape != (32, 64) or actor["head_b"].numel() != 64):
            raise ValueError("rr_env_unroll_policy: inconsistent shapes")
        w = RRUnrollIO(self._state(first), first_obs.data_ptr(), prev_done.data_ptr(), steps_in.data_ptr(), steps_out.data_ptr(),
                       truncation_out.data_ptr(), float(episode_length))
        p = lambda t: t.data_ptr() if t is not None else None
        a = RRActorIO(obs_in.data_ptr(), p(actor.get("mean")), p(actor.get("std")), actor["w0"].data_ptr(), actor["b0"].data_ptr(),
                      (C.c_void_p * 4)(*([t.data_ptr() for t in actor["hidden_wt"]] + [None] * (4 - nh + 1))),
                      (C.c_void_p * 4)(*([t.data_ptr() for t in actor["hidden_b"]] + [None] * (4 - nh + 1))),
                      actor["head_wt"].data_ptr(), actor["head_b"].data_ptr(), noise.data_ptr(), actions_out.data_ptr(), traj["obs"].data_ptr(),
                      traj["raw_action"].data_ptr(), traj["log_prob"].data_ptr(), traj["reward"].data_ptr(), traj["discount"].data_ptr(),
                      traj["truncation"].data_ptr(), float(actor["min_std"]), nh, int(L_))
        _check(lib().rr_env_unroll_policy(self.h, C.byref(self._state(st_in)), C.byref(self._state(st_out)), int(T), int(n_frames),
                                          C.byref(self._env(env)), _ptr(cur_frame_in, torch.int32, self.N), C.byref(w), C.byref(a)))

    def env_reset(self, st, env, out=None):
        o = self._outputs(out)
        _check(lib().rr_env_reset(self.h, C.byref(self._state(st)), C.byref(self._env(env)), C.byref(o) if o else None))

    def debug_layout(self):
        names, offs, sizes = C.POINTER(C.c_char_p)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        n = _check(lib().rr_debug_layout(self.h, C.byref(names), C.byref(offs), C.byref(sizes)))
        return {names[i].decode(): (offs[i], sizes[i]) for i in range(n)}

    def set_profile(self, buf: Optional[torch.Tensor]):
        """Diagnostic: int64 device tensor [N,16] receiving per-phase cycle sums (None = off)."""
        _check(lib().rr_batch_set_profile(self.h, buf.data_ptr() if buf is not None else None))

    def set_schedule(self, env_map: Optional[torch.Tensor], cost: Optional[torch.Tensor]):
        """workgroup -> environment map (int32 [N] device, a permutation) and per-env cycle output (int32 [N] device); None = off.
        The tensors must stay alive while the batch launches (C ABI `rr_batch_set_schedule`)."""
        self._sched = (env_map, cost)
        _check(lib().rr_batch_set_schedule(self.h, _ptr(env_map, torch.int32, self.N) if env_map is not None else None,
                                           _ptr(cost, torch.int32, self.N) if cost is not None else None))

    def set_timing(self, enable: bool):
        _check(lib().rr_batch_set_timing(self.h, int(enable)))

    def kernel_time(self):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().rr_batch_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_: float, discount: float):
    """GAE on the GPU in one launch (C ABI `rr_compute_gae`): inputs time-major [T, B] float32 device tensors."""
    T, B = values.shape
    args = [t.contiguous() for t in (truncation, termination, rewards, values, bootstrap_value)]
    for t in args:
        _ptr(t)
    vs = torch.empty_like(args[3])
    adv = torch.empty_like(args[3])
    stream = torch.cuda.current_stream(values.device).cuda_stream
    _check(lib().rr_compute_gae(*[t.data_ptr() for t in args], T, B, float(lambda_), float(discount), vs.data_ptr(),
                                adv.data_ptr(), C.c_void_p(stream)))
    return vs, adv


def wrap_episode_autoreset(first, cur, prev_done, prev_steps, done, steps, truncation, episode_length: float, action_repeat: float):
    """Fused EpisodeWrapper + AutoResetWrapper bookkeeping (C ABI `rr_wrap_episode_autoreset`): `first` / `cur` are equally
    long lists of contiguous float32 device tensors with a leading env axis; `cur`, `done`, `steps`, `truncation` are written."""
    n = len(cur)
    N = done.numel()
    for t in list(first) + list(cur) + [prev_done, prev_steps, done, steps, truncation]:
        _ptr(t)
    F = (C.c_void_p * n)(*[t.data_ptr() for t in first])
    Cu = (C.c_void_p * n)(*[t.data_ptr() for t in cur])
    W = (C.c_int32 * n)(*[t.numel() // N for t in cur])
    for f, c in zip(first, cur):
        if f.shape != c.shape:
            raise ValueError("first / current state shapes differ")
    stream = torch.cuda.current_stream(done.device).cuda_stream
    _check(lib().rr_wrap_episode_autoreset(N, n, F, Cu, W, prev_done.data_ptr(), prev_steps.data_ptr(), done.data_ptr(), steps.data_ptr(),
                                           truncation.data_ptr(), float(episode_length), float(action_repeat), C.c_void_p(stream)))


def _mlp_net(weights, biases, in_dim=None):
    """(RRMlpNet, keep-alive) for a list of nn.Linear-layout weights [out, in] and biases [out] (float32 device tensors); in_dim: the input
    width when the first matrix is stored with padded rows."""
    n = len(weights)
    for w, b in zip(weights, biases):
        _ptr(w); _ptr(b)
        if w.dim() != 2 or b.numel() != w.shape[0]:
            raise ValueError("weights must be [out, in] with biases [out]")
    W = (C.c_void_p * n)(*[w.data_ptr() for w in weights])
    B = (C.c_void_p * n)(*[b.data_ptr() for b in biases])
    S = (C.c_int32 * (n + 1))(in_dim or weights[0].shape[1], *[w.shape[0] for w in weights])
    return RRMlpNet(C.cast(W, C.POINTER(C.c_void_p)), C.cast(B, C.POINTER(C.c_void_p)), C.cast(S, C.POINTER(C.c_int32)), n), (W, B, S)


def mlp_forward(obs, mean=None, std=None, policy=None, value=None, want_pre=False, rows=None):
    """Fused normalise + policy MLP + value MLP forward on the f32 matrix cores (C ABI `rr_mlp_forward`).

    obs [M, K] float32 device; policy / value: (weights, biases) lists in nn.Linear layout or None.  Returns
    (policy_out [M, P] | None, value_out [M] | None, policy_pre [L-1, M, 32] | None, value_pre [L-1, M, 256] | None).
    rows (int64 [M], optional): sample m is row rows[m] of obs (the minibatch addressed in place)."""
    M, K = obs.shape
    _ptr(obs)
    if rows is not None:
        M = rows.numel()
        _ptr(rows, torch.int64)
    dev = obs.device
    pn = vn = None
    keep = []
    pol_out = val_out = pol_pre = val_pre = None
    if policy is not None:
        pn, k = _mlp_net(*policy, in_dim=K); keep.append((k, policy))
        pol_out = torch.empty(M, policy[0][-1].shape[0], device=dev)
        if want_pre:
            pol_pre = torch.empty(len(policy[0]) - 1, M, 32, device=dev)
    if value is not None:
        vn, k = _mlp_net(*value, in_dim=K); keep.append((k, value))
        val_out = torch.empty(M, device=dev)
        if want_pre:
            val_pre = torch.empty(len(value[0]) - 1, M, 256, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: t.data_ptr() if t is not None else None
    _check(lib().rr_mlp_forward(obs.data_ptr(), rows.data_ptr() if rows is not None else None, M, K, _ptr(mean, numel=K) if mean is not None else None,
                                _ptr(std, numel=K) if std is not None else None, C.byref(pn) if pn is not None else None,
                                C.byref(vn) if vn is not None else None, p(pol_out), p(val_out), p(pol_pre), p(val_pre), C.c_void_p(stream)))
    return pol_out, val_out, pol_pre, val_pre


def ppo_loss(policy_logits, values, data, idx, noise, T: int, *, entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon,
             normalize_advantage=True, min_std=0.001, out=None):
    """Loss half of a PPO minibatch update and its gradient w.r.t. the network outputs (C ABI `rr_ppo_loss`, 3 launches).

    policy_logits [(T+1)*B, 2A] / values [(T+1)*B]: the networks' outputs on the gathered minibatch, time-major; `data`: the
    collected batch, batch-major (`raw_action` [R, T, A]; `log_prob`, `reward`, `discount`, `truncation` [R, T]); idx [B] int64
    minibatch rows (None = the first B rows); noise [T*B, A].  Returns (grad_logits, grad_values, metrics[4]); `out` = a dict
    of persistent buffers {grad_logits, grad_values, metrics, workspace} reused across calls (HIP-graph capture needs that)."""
    M, P2 = policy_logits.shape
    B = M // (T + 1)
    A = P2 // 2
    dev = policy_logits.device
    for t in (policy_logits, values, noise, data["raw_action"], data["log_prob"], data["reward"], data["discount"], data["truncation"]):
        _ptr(t)
    if values.numel() != M or noise.numel() != T * B * A or data["raw_action"].shape[1:] != (T, A) or data["log_prob"].shape[1] != T:
        raise ValueError("rr_ppo_loss: inconsistent shapes")
    if idx is not None:
        _ptr(idx, torch.int64, B)
    elif data["log_prob"].shape[0] < B:
        raise ValueError("rr_ppo_loss: fewer batch rows than the minibatch")
    out = out if out is not None else {}
    wb = lib().rr_ppo_loss_workspace_bytes(T, B)
    if "workspace" not in out or out["workspace"].numel() * 8 < wb or out["grad_logits"].shape != policy_logits.shape:
        out["workspace"] = torch.empty((wb + 7) // 8, dtype=torch.float64, device=dev)
        out["grad_logits"] = torch.empty_like(policy_logits)
        out["grad_values"] = torch.empty(M, device=dev)
        out["metrics"] = torch.empty(4, device=dev)
    cfg = RRPpoCfg(entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon, min_std, 1 if normalize_advantage else 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _check(lib().rr_ppo_loss(policy_logits.data_ptr(), values.data_ptr(), data["raw_action"].data_ptr(), data["log_prob"].data_ptr(),
                             data["reward"].data_ptr(), data["discount"].data_ptr(), data["truncation"].data_ptr(),
                             idx.data_ptr() if idx is not None else None, noise.data_ptr(), T, B, A, C.byref(cfg),
                             out["grad_logits"].data_ptr(), out["grad_values"].data_ptr(), out["metrics"].data_ptr(),
                             out["workspace"].data_ptr(), out["workspace"].numel() * 8, C.c_void_p(stream)))
    return out["grad_logits"], out["grad_values"], out["metrics"]


_silu_ws = {}


def mlp_silu_backward(g, z, bias_grad):
    """delta = g * silu'(z) written over `g`, h = silu(z) written over `z`, bias_grad[n] = column sums of delta (C ABI
    `rr_mlp_silu_backward`).  g, z: [M, H] contiguous float32 device tensors.  Returns (delta, h) = (g, z)."""
    M, H = g.shape
    _ptr(g); _ptr(z, numel=M * H); _ptr(bias_grad, numel=H)
    wb = lib().rr_mlp_silu_backward_workspace_bytes(M, H)
    key = (g.device, M, H, torch.cuda.current_stream(g.device).cuda_stream)      # per stream: the workspace is scratch of the launch
    if key not in _silu_ws:
        _silu_ws[key] = torch.empty((wb + 3) // 4, device=g.device)
    ws = _silu_ws[key]
    _check(lib().rr_mlp_silu_backward(g.data_ptr(), z.data_ptr(), M, H, g.data_ptr(), z.data_ptr(), bias_grad.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                      C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)))
    return g, z


def mlp_value_backward(grad_value, head_weight, hidden_weights_t, pre_act, bias_grads, bufs=None):
    """Delta chain of the value network's hidden stack on the matrix cores (C ABI `rr_mlp_value_backward`).

    grad_value [M]; head_weight [1, 256] or [256]; hidden_weights_t: list, entry j >= 1 = W_j.t().contiguous() (entry 0 ignored);
    pre_act [nh, M, 256] (rr_mlp_forward's value_pre; overwritten by silu(z)); bias_grads: list of nh [256] tensors (written).
    Returns (delta [nh, M, 256], h = pre_act)."""
    nh, M, H = pre_act.shape
    if H != 256 or len(bias_grads) != nh or len(hidden_weights_t) != nh:
        raise ValueError("rr_mlp_value_backward: inconsistent shapes")
    _ptr(grad_value, numel=M); _ptr(head_weight, numel=256); _ptr(pre_act)
    for j in range(nh):
        _ptr(bias_grads[j], numel=256)
        if j > 0:
            _ptr(hidden_weights_t[j], numel=256 * 256)
    bufs = bufs if bufs is not None else {}
    wb = lib().rr_mlp_value_backward_workspace_bytes(M, nh)
    if bufs.get("vb_key") != (M, nh):
        bufs["vb_ws"] = torch.empty((wb + 3) // 4, device=pre_act.device)
        bufs["vb_delta"] = torch.empty_like(pre_act)
        bufs["vb_key"] = (M, nh)
    wt = (C.c_void_p * nh)(*[hidden_weights_t[j].data_ptr() if j > 0 else None for j in range(nh)])
    bg = (C.c_void_p * nh)(*[b.data_ptr() for b in bias_grads])
    _check(lib().rr_mlp_value_backward(grad_value.data_ptr(), head_weight.data_ptr(), wt, nh, M, pre_act.data_ptr(), bufs["vb_delta"].data_ptr(), bg,
                                       bufs["vb_ws"].data_ptr(), bufs["vb_ws"].numel() * 4, C.c_void_p(torch.cuda.current_stream(pre_act.device).cuda_stream)))
    return bufs["vb_delta"], pre_act


_dw_ws = {}


def mlp_weight_grad(delta, act, out, rows=None, mean=None, std=None, delta_colsum=None):
    """out[o, i] = sum_m delta[m, o] * x[m, i] (C ABI `rr_mlp_weight_grad`): x = act[m] or, with `rows`, act[rows[m]]; with
    mean / std (and delta_colsum [O] = the column sums of delta), x = (act - mean) / std.  delta [M, O], act [R, I],
    out [O, I]: contiguous float32."""
    M, O = delta.shape
    I = act.shape[1]
    _ptr(delta); _ptr(act); _ptr(out, numel=O * I)
    if rows is not None:
        _ptr(rows, torch.int64, M)
    elif act.shape[0] < M:
        raise ValueError("rr_mlp_weight_grad: fewer activation rows than delta rows")
    key = (delta.device, M, O, I, torch.cuda.current_stream(delta.device).cuda_stream)      # per stream (scratch of the launch)
    if key not in _dw_ws:
        _dw_ws[key] = torch.empty((lib().rr_mlp_weight_grad_workspace_bytes(M, O, I) + 3) // 4, device=delta.device)
    ws = _dw_ws[key]
    _check(lib().rr_mlp_weight_grad(delta.data_ptr(), act.data_ptr(), rows.data_ptr() if rows is not None else None,
                                    _ptr(mean, numel=I) if mean is not None else None, _ptr(std, numel=I) if std is not None else None,
                                    _ptr(delta_colsum, numel=O) if mean is not None else None, M, O, I, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, C.c_void_p(torch.cuda.current_stream(delta.device).cuda_stream)))
    return out


def obs_moments(obs, T: int, mean, bufs=None):
    """S1 = sum (x - mean), S2 = sum (x - mean)^2 per observation column (float64 [2, K]) over the rows t < T of every sequence of
    `obs` [..., Tp1, K] (contiguous float32 device tensor; C ABI `rr_obs_moments`): the normaliser update's sums in one pass."""
    K, Tp1 = obs.shape[-1], obs.shape[-2]
    nseq = obs.numel() // (Tp1 * K)
    _ptr(obs); _ptr(mean, numel=K)
    nb = lib().rr_obs_moments_workspace_bytes(nseq, T, K)
    bufs = {} if bufs is None else bufs
    ws = bufs.get("mom_ws")
    if ws is None or ws.numel() * 8 < nb or ws.device != obs.device:
        ws = bufs["mom_ws"] = torch.empty((nb + 7) // 8, dtype=torch.float64, device=obs.device)
    sums = torch.empty(2, K, dtype=torch.float64, device=obs.device)
    _check(lib().rr_obs_moments(obs.data_ptr(), nseq, Tp1, T, K, mean.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)))
    return sums


def policy_sample(logits, noise, min_std: float):
    """(action, raw_action, log_prob) of the tanh-normal policy head in one launch (C ABI `rr_policy_sample`).
    logits [N, 2A], noise [N, A]: contiguous float32 device tensors."""
    N, P2 = logits.shape
    A = P2 // 2
    _ptr(logits); _ptr(noise, numel=N * A)
    action, raw, lp = torch.empty(N, A, device=logits.device), torch.empty(N, A, device=logits.device), torch.empty(N, device=logits.device)
    _check(lib().rr_policy_sample(logits.data_ptr(), noise.data_ptr(), N, A, min_std, action.data_ptr(), raw.data_ptr(), lp.data_ptr(),
                                  C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)))
    return action, raw, lp


def policy_backward(grad_logits, head_weight, hidden_weights, pre_act, bias_grads, bufs=None):
    """Delta chain of the policy network's 32-wide hidden stack in one launch (C ABI `rr_policy_backward`).

    grad_logits [M, P]; head_weight [P, 32]; hidden_weights: list, entry j >= 1 = W_j [32, 32] (entry 0 ignored); pre_act
    [nh, >= M, 32] (rr_mlp_forward's policy_pre; the first M rows of each layer are overwritten by silu(z)); bias_grads: list of
    nh [32] tensors.  Returns (delta [nh, M, 32], h = pre_act)."""
    nh = pre_act.shape[0]
    M, P = grad_logits.shape
    if pre_act.shape[2] != 32 or pre_act.shape[1] < M or len(bias_grads) != nh or len(hidden_weights) != nh or head_weight.shape != (P, 32):
        raise ValueError("rr_policy_backward: inconsistent shapes")
    _ptr(grad_logits); _ptr(head_weight); _ptr(pre_act)
    for j in range(nh):
        _ptr(bias_grads[j], numel=32)
        if j > 0:
            _ptr(hidden_weights[j], numel=1024)
    bufs = bufs if bufs is not None else {}
    if bufs.get("pb_key") != (M, nh):
        bufs["pb_ws"] = torch.empty((lib().rr_policy_backward_workspace_bytes(M, nh) + 3) // 4, device=pre_act.device)
        bufs["pb_delta"] = torch.empty(nh, M, 32, device=pre_act.device)
        bufs["pb_key"] = (M, nh)
    wt = (C.c_void_p * nh)(*[hidden_weights[j].data_ptr() if j > 0 else None for j in range(nh)])
    bg = (C.c_void_p * nh)(*[b.data_ptr() for b in bias_grads])
    _check(lib().rr_policy_backward(grad_logits.data_ptr(), head_weight.data_ptr(), wt, nh, M, P, pre_act.data_ptr(), pre_act.shape[1], bufs["pb_delta"].data_ptr(), bg,
                                    bufs["pb_ws"].data_ptr(), bufs["pb_ws"].numel() * 4, C.c_void_p(torch.cuda.current_stream(pre_act.device).cuda_stream)))
    return bufs["pb_delta"], pre_act


_pa_ws = {}


def policy_act(obs, mean, std, policy, noise, min_std: float, want_logits: bool = False, rows=None):
    """The rollout's actor step in two launches (C ABI `rr_policy_act`): obs [M, K] (or obs[rows]) -> normalise -> policy MLP ->
    tanh-normal head.  noise [M, A] or None (deterministic: action = tanh(loc)).  Returns (action, raw_action | None,
    log_prob | None, logits | None)."""
    M, K = obs.shape
    _ptr(obs)
    if rows is not None:
        M = rows.numel()
        _ptr(rows, torch.int64)
    pn, keep = _mlp_net(*policy)
    A = policy[0][-1].shape[0] // 2
    dev = obs.device
    action = torch.empty(M, A, device=dev)
    raw = lp = logits = None
    if noise is not None:
        _ptr(noise, numel=M * A)
        raw, lp = torch.empty(M, A, device=dev), torch.empty(M, device=dev)
    if want_logits:
        logits = torch.empty(M, 2 * A, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, M, stream)                 # per stream: sub-batches on several streams run this concurrently
    if key not in _pa_ws:
        _pa_ws[key] = torch.empty((lib().rr_policy_act_workspace_bytes(M) + 3) // 4, device=dev)
    ws = _pa_ws[key]
    p = lambda t: t.data_ptr() if t is not None else None
    _check(lib().rr_policy_act(obs.data_ptr(), p(rows), M, K, _ptr(mean, numel=K) if mean is not None else None,
                               _ptr(std, numel=K) if std is not None else None, C.byref(pn), p(noise), min_std, action.data_ptr(), p(raw), p(lp),
                               p(logits), ws.data_ptr(), ws.numel() * 4, C.c_void_p(stream)))
    return action, raw, lp, logits


_dwb_ws = {}


def mlp_weight_grad_batch(items):
    """Several weight gradients in one call (C ABI `rr_mlp_weight_grad_batch`, at most 12): `items` = dicts with the arguments of
    `mlp_weight_grad` (delta, act, out, and optionally rows, mean, std, delta_colsum).  Products of one tile shape share a launch,
    one reduction launch sums all partial tiles."""
    n = len(items)
    arr = (RRDwItem * n)()
    dev = items[0]["delta"].device
    shape_key = []
    for i, it in enumerate(items):
        delta, act, out = it["delta"], it["act"], it["out"]
        M, O = delta.shape
        I = act.shape[1]
        _ptr(delta); _ptr(act); _ptr(out, numel=O * I)
        rows, mean, std, cs = it.get("rows"), it.get("mean"), it.get("std"), it.get("delta_colsum")
        if rows is not None:
            _ptr(rows, torch.int64, M)
        elif act.shape[0] < M:
            raise ValueError("rr_mlp_weight_grad_batch: fewer activation rows than delta rows")
        p = lambda t: t.data_ptr() if t is not None else None
        arr[i] = RRDwItem(delta.data_ptr(), act.data_ptr(), p(rows), _ptr(mean, numel=I) if mean is not None else None,
                          _ptr(std, numel=I) if std is not None else None, _ptr(cs, numel=O) if mean is not None else None, M, O, I, out.data_ptr())
        shape_key.append((M, O, I))
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, tuple(shape_key), stream)
    if key not in _dwb_ws:
        _dwb_ws[key] = torch.empty((lib().rr_mlp_weight_grad_batch_workspace_bytes(arr, n) + 3) // 4, device=dev)
    ws = _dwb_ws[key]
    _check(lib().rr_mlp_weight_grad_batch(arr, n, ws.data_ptr(), ws.numel() * 4, C.c_void_p(stream)))
