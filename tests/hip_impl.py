"""The HIP path (through the C ABI) in the calling convention of the parity harness (tests/parity.py)."""
import numpy as np
import torch

from tests import parity

DEV = "cuda:0"


class HipImpl:
    def __init__(self, model_name, n, iterations, debug, solver=None):
        from rodent_amd import assets, hip, mjcf
        self.model = hip.Model(assets.asset_path(model_name), *iterations, solver=solver)
        self.batch = hip.Batch(self.model, n, torch.device(DEV))
        self.n, self.debug = n, debug
        tab = mjcf.load_blob(assets.asset_path(model_name))
        self.lim_dof = tab["jnt_dofadr"][tab["limit_jnt"]]
        self.lay = self.batch.debug_layout()
        self.dbg = torch.zeros(n, self.batch.dims.dbg_floats, device=DEV) if debug else None

    def _dev(self, st):
        return {k: torch.tensor(st[k], dtype=torch.float32, device=DEV).contiguous() for k in parity.STATE}

    def substep(self, st, ctrl):
        ds = self._dev(st)
        self.batch.pipeline_step(ds, torch.tensor(ctrl, dtype=torch.float32, device=DEV), 1, out=dict(debug=self.dbg) if self.debug else None)
        out = {k: v.cpu().numpy().astype(np.float64) for k, v in ds.items()}
        if self.debug:
            g = self.dbg.cpu().numpy().astype(np.float64)
            f = lambda name: g[:, self.lay[name][0]:self.lay[name][0] + self.lay[name][1]]
            lim = f("limit_pos_D_aref").reshape(self.n, -1, 3)[:, self.lim_dof]
            out.update(con_dist=f("con_dist"), lim_pos=lim[:, :, 0], lim_D=lim[:, :, 1], lim_aref=lim[:, :, 2],
                       niter=f("niter_cost")[:, 0].astype(int))
        return out


class NoDiscrete:
    """Wraps an impl without a debug dump: the discrete checks compare the oracle with itself (state criteria only)."""

    def __init__(self, impl, A):
        self.impl, self.A = impl, A

    def substep(self, st, ctrl):
        out = self.impl.substep(st, ctrl)
        want = self.A.substep(st, ctrl)
        for k in ("con_dist", "lim_pos", "lim_D", "lim_aref", "niter"):
            out[k] = want[k]
        return out


class HipEnvImpl:
    def __init__(self, n, iterations, track, model_name="rodent_optimized", solver="cg"):
        from rodent_amd import envs
        self.env = envs.get_environment("rodent", track_pos=track, num_envs=n, xml_path=f"{model_name}.xml", solver=solver,
                                        iterations=iterations[0], ls_iterations=iterations[1], device=DEV)
        self.state0 = self.env.reset(0)

    def env_step(self, st, ctrl, cur_frame):
        ps = self.state0.pipeline_state.replace(**{k: torch.tensor(st[k], dtype=torch.float32, device=DEV).contiguous() for k in parity.STATE})
        s = self.state0.replace(pipeline_state=ps, info=dict(cur_frame=torch.tensor(cur_frame, dtype=torch.int32, device=DEV)))
        ns = self.env.step(s, torch.tensor(ctrl, dtype=torch.float32, device=DEV))
        out = {k: getattr(ns.pipeline_state, k).cpu().numpy().astype(np.float64) for k in parity.STATE}
        out.update(obs=ns.obs.cpu().numpy().astype(np.float64), reward=ns.reward.cpu().numpy().astype(np.float64),
                   done=ns.done.cpu().numpy().astype(np.float64), cur_frame=ns.info["cur_frame"].cpu().numpy())
        return out
