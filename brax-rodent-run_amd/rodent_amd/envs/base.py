"""`State` / `PipelineState` / `PipelineEnv`: the torch-tensor mirror of `brax.envs.base`
(the reference imports `PipelineEnv, State` from there [REF Rodent_Env_Brax.py:4]).

Batched by construction: every leaf has a leading env axis [N, ...] instead of being vmapped.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional

import torch

from .. import assets, hip


@dataclasses.dataclass
class PipelineState:
    """What the reference env reads from `mjx.Data` [REF Rodent_Env_Brax.py:110,116,151-155,161]."""
    qpos: torch.Tensor            # [N, nq]
    qvel: torch.Tensor            # [N, nv]
    act: torch.Tensor             # [N, na]
    qacc_warmstart: torch.Tensor  # [N, nv]
    # Derived quantities of the last forward pass.  `Rodent.step` leaves them None unless the env was built with
    # `pipeline_outputs=True`: the observation already carries cinert / cvel / qfrc_actuator [REF Rodent_Env_Brax.py:151-155]
    # and writing them a second time was 7.7 KB per env-step of HBM traffic (+ the AutoReset select over them).
    cinert: Optional[torch.Tensor] = None          # [N, nbody, 10]
    cvel: Optional[torch.Tensor] = None            # [N, nbody, 6]
    qfrc_actuator: Optional[torch.Tensor] = None   # [N, nv]
    xpos: Optional[torch.Tensor] = None            # [N, nbody, 3]
    xmat: Optional[torch.Tensor] = None            # [N, nbody, 9]
    # brax `State.contact` (mjx.Data.contact) geometry [NB mjcf.ipynb:917-921]; filled with `contact_outputs=True`;
    # the static integer fields are `sys.contact_geom1 / contact_geom2 / contact_link_idx`
    contact_dist: Optional[torch.Tensor] = None    # [N, ncon]
    contact_pos: Optional[torch.Tensor] = None     # [N, ncon, 3]
    contact_frame: Optional[torch.Tensor] = None   # [N, ncon, 3, 3]

    @property
    def q(self):
        return self.qpos

    @property
    def qd(self):
        return self.qvel

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def tree(self):
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if getattr(self, f.name) is not None}


@dataclasses.dataclass
class Contact:
    """brax `State.contact` / `mjx.Data.contact` with the fields the reference's notebook shows [NB mjcf.ipynb:917-921]:
    per-step geometry from the kernel (`contact_outputs=True`), static parameters and integer ids from the loaded model
    (C ABI `rr_model_table`).  Leading env axis on the per-step fields."""
    dist: torch.Tensor            # [N, ncon]
    pos: torch.Tensor             # [N, ncon, 3]
    frame: torch.Tensor           # [N, ncon, 3, 3]
    includemargin: torch.Tensor   # [ncon] (0: the rodent models use no margin)
    friction: torch.Tensor        # [ncon, 5]
    solref: torch.Tensor          # [ncon, 2]
    solreffriction: torch.Tensor  # [ncon, 2] (0)
    solimp: torch.Tensor          # [ncon, 5]
    geom1: torch.Tensor           # [ncon] int32
    geom2: torch.Tensor           # [ncon] int32
    link_idx: tuple               # (geom_bodyid[geom1] - 1, geom_bodyid[geom2] - 1), int32 [ncon] each
    elasticity: torch.Tensor      # [ncon] (0)


@dataclasses.dataclass
class State:
    """Environment state for training and inference (brax.envs.base.State)."""
    pipeline_state: Optional[PipelineState]
    obs: torch.Tensor
    reward: torch.Tensor
    done: torch.Tensor
    metrics: Dict[str, torch.Tensor] = dataclasses.field(default_factory=dict)
    info: Dict[str, Any] = dataclasses.field(default_factory=dict)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class System:
    """The few `sys.*` attributes the reference env touches [REF Rodent_Env_Brax.py:82-89]."""

    def __init__(self, blob_path: str, iterations: int, ls_iterations: int, solver: str = "cg"):
        from .. import mjcf
        self.blob_path = blob_path
        self.tables = mjcf.load_blob(blob_path)
        self.model = hip.Model(blob_path, iterations=iterations, ls_iterations=ls_iterations, solver=solver)
        self.solver = solver.lower()
        d = self.model.dims
        self.nq, self.nv, self.nu, self.na, self.nbody = d.nq, d.nv, d.nu, d.na, d.nbody
        self.obs_dim = d.obs_dim
        self.dt = float(d.timestep)
        self.qpos0 = self.tables["qpos0"]
        self.ncon = d.ncon
        # integer ids as the LOADED model holds them (C ABI rr_model_table): Contact.geom1 / geom2 and brax's link_idx
        # (= geom_bodyid[g] - 1) [NB mjcf.ipynb:917-921]
        self.contact_geom1 = self.model.table("con_geom1")
        self.contact_geom2 = self.model.table("con_geom2")
        self.geom_bodyid = self.model.table("geom_bodyid")
        self.contact_link_idx = (self.geom_bodyid[self.contact_geom1] - 1, self.geom_bodyid[self.contact_geom2] - 1)


class PipelineEnv:
    """API for driving the HIP physics backend (`brax.envs.base.PipelineEnv` with backend='mjx'
    replaced by the fused HIP kernel): `pipeline_init` / `pipeline_step` [REF Rodent_Env_Brax.py:87,101]."""

    def __init__(self, sys: System, num_envs: int, n_frames: int = 1, backend: str = "hip", device=None, debug=False,
                 pipeline_outputs: bool = False, contact_outputs: bool = False, balance: Optional[bool] = None, rebalance_every: int = 4):
        if backend not in ("hip", "mjx"):
            raise ValueError(f"backend {backend!r} not available: this build provides the HIP backend only")
        self.sys = sys
        self._n_frames = n_frames
        self._debug = debug
        self._pipeline_outputs = pipeline_outputs
        self._contact_outputs = contact_outputs
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._batch = hip.Batch(sys.model, self.num_envs, self.device)
        # SIMD pairing (scheduling only; results are unaffected; OFF by default).  When the batch is exactly one resident round of
        # the GPU (2 waves per SIMD: N = 8 x CUs), workgroups w and w + N/2 share a SIMD and the launch lasts as long as its
        # slowest environment.  With `balance=True` the envs are re-mapped every `rebalance_every` steps so that the costliest
        # ones of the last launch (work estimate reported by the kernel) sit with the cheapest ones.  With the TRUE cost of the
        # step being launched the effect is real (MI355X, same 2048 states: heavy|light pairs 1.412 ms, random 1.452 ms,
        # heavy|heavy 1.519 ms; tools/pairing_probe.py), but the previous step's cost does not predict it well enough under
        # random actions: the bench measured 1.5027 ms balanced against 1.5038 ms unbalanced, so it stays an option.
        self._balance = bool(balance)
        self._rebalance_every, self._launches = max(1, int(rebalance_every)), 0
        if self._balance:
            self._env_map = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
            self._cost = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            self._batch.set_schedule(self._env_map, self._cost)

    def _rebalance(self):
        """Call before a step launch: every `rebalance_every` launches, pair heavy with light environments."""
        if not self._balance:
            return
        self._launches += 1
        if self._launches % self._rebalance_every == 0:
            order = torch.argsort(self._cost, descending=True)
            half = self.num_envs // 2
            self._env_map[:half] = order[:half]
            self._env_map[half:] = order.flip(0)[:half]

    # -- brax surface
    @property
    def dt(self) -> float:
        return self.sys.dt * self._n_frames

    @property
    def observation_size(self) -> int:
        return self.sys.obs_dim

    @property
    def action_size(self) -> int:
        return self.sys.nu

    @property
    def backend(self) -> str:
        return "hip"

    def _alloc_outputs(self, full: bool = True):
        """Buffers for the optional rr_outputs; `full=False` honours the env's pipeline_outputs / contact_outputs switches
        (the rollout path: nothing unless asked for)."""
        N, s, dev = self.num_envs, self.sys, self.device
        out = {}
        if full or self._pipeline_outputs:
            out.update(cinert=torch.empty(N, s.nbody, 10, device=dev), cvel=torch.empty(N, s.nbody, 6, device=dev),
                       qfrc_actuator=torch.empty(N, s.nv, device=dev), xpos=torch.empty(N, s.nbody, 3, device=dev),
                       xmat=torch.empty(N, s.nbody, 9, device=dev))
        if self._contact_outputs:
            out.update(contact_dist=torch.empty(N, s.ncon, device=dev), contact_pos=torch.empty(N, s.ncon, 3, device=dev),
                       contact_frame=torch.empty(N, s.ncon, 3, 3, device=dev))
        return out

    def contact(self, pipeline_state: PipelineState) -> Contact:
        """The brax-style `Contact` of a pipeline state produced with `contact_outputs=True`."""
        if pipeline_state.contact_dist is None:
            raise ValueError("build the env with contact_outputs=True to get the contact geometry")
        m, dev, nc = self.sys.model, self.device, self.sys.ncon
        t = lambda name, w: torch.from_numpy(m.table(name).reshape(nc, w) if w > 1 else m.table(name)).to(dev)
        i32 = lambda a: torch.from_numpy(a.astype("int32")).to(dev)
        z = torch.zeros(nc, device=dev)
        return Contact(dist=pipeline_state.contact_dist, pos=pipeline_state.contact_pos, frame=pipeline_state.contact_frame,
                       includemargin=z, friction=t("con_friction", 5), solref=t("con_solref", 2), solreffriction=torch.zeros(nc, 2, device=dev),
                       solimp=t("con_solimp", 5), geom1=i32(self.sys.contact_geom1), geom2=i32(self.sys.contact_geom2),
                       link_idx=(i32(self.sys.contact_link_idx[0]), i32(self.sys.contact_link_idx[1])), elasticity=z.clone())

    def pipeline_init(self, q: torch.Tensor, qd: torch.Tensor) -> PipelineState:
        N, s, dev = self.num_envs, self.sys, self.device
        st = dict(qpos=q.to(dev, torch.float32).contiguous().clone(), qvel=qd.to(dev, torch.float32).contiguous().clone(),
                  act=torch.zeros(N, s.na, device=dev), qacc_warmstart=torch.zeros(N, s.nv, device=dev))
        out = self._alloc_outputs()
        self._batch.pipeline_init(st, out)
        return PipelineState(**st, **out)

    def pipeline_step(self, pipeline_state: PipelineState, action: torch.Tensor) -> PipelineState:
        st_in = dict(qpos=pipeline_state.qpos, qvel=pipeline_state.qvel, act=pipeline_state.act,
                     qacc_warmstart=pipeline_state.qacc_warmstart)
        st = {k: torch.empty_like(v) for k, v in st_in.items()}      # out of place: the argument stays valid
        out = self._alloc_outputs()
        self._rebalance()
        self._batch.pipeline_step_to(st_in, st, action.to(self.device, torch.float32).contiguous(), self._n_frames, out)
        return PipelineState(**st, **out)
