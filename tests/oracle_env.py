"""CPU counterpart of `Rodent` + the training wrappers, on the oracle (test infrastructure).

`OracleRodent` restates `Rodent.reset / step` [REF Rodent_Env_Brax.py:71-136] and the brax
`EpisodeWrapper + AutoResetWrapper` semantics [UP brax.envs.wrappers.training; SURVEY.md 3.4] with
numpy around `oracle/rodent_ref.c`, so a HIP trajectory (rollout through `wrappers.wrap`) can be
compared step by step: same reset keys -> same initial states; done -> the stored first state comes
back, `info` (hence `cur_frame`) does not.
"""
import numpy as np

from oracle import ref
from rodent_amd import assets, jax_random, mjcf


class OracleRodent:
    def __init__(self, model_name, n, precision="f64", iterations=(8, 8), track=None, episode_length=None,
                 healthy_z_range=(0.03, 0.5), reset_noise_scale=1e-2, n_frames=10):
        self.path = assets.asset_path(model_name)
        self.tables = mjcf.load_blob(self.path)
        self.M = ref.RefModel(self.path, precision)
        self.M.set_iterations(*iterations)
        self.n = n
        self.b = ref.RefBatch(self.M, n)
        self.track = np.asarray(track, np.float64)
        self.episode_length = episode_length
        self.z = healthy_z_range
        self.noise = reset_noise_scale
        self.n_frames = n_frames

    # -- Rodent.reset [REF Rodent_Env_Brax.py:71-96]
    def reset_state(self, rng):
        n = self.n
        if isinstance(rng, (int, np.integer)):
            rng = jax_random.split(jax_random.PRNGKey(int(rng)), n)
        keys = np.asarray(rng, np.uint32).reshape(n, 2)
        ks = jax_random.split(keys, 4)
        start = jax_random.randint(ks[:, 0], 0, 100)
        qpos = np.tile(self.tables["qpos0"].astype(np.float32), (n, 1))
        qpos[:, :3] = self.track.astype(np.float32)[np.clip(start, 0, len(self.track) - 1)]
        qpos = qpos + jax_random.uniform(ks[:, 1], self.M.nq, -self.noise, self.noise)
        qvel = jax_random.uniform(ks[:, 2], self.M.nv, -self.noise, self.noise)
        return qpos.astype(np.float64), qvel.astype(np.float64), start.astype(np.int32)

    def reset(self, rng):
        qpos, qvel, start = self.reset_state(rng)
        self.b.init(qpos, qvel)
        self.cur_frame = start.copy()
        self.obs = self.b.obs(self.track, self.cur_frame)
        self.done = np.zeros(self.n)
        self.steps = np.zeros(self.n)
        self.truncation = np.zeros(self.n)
        self.first = self.b.state()
        self.first_obs = self.obs.copy()
        return self.obs

    # -- Rodent.step (+ Episode + AutoReset when episode_length is given)
    def step(self, action, force_state=None):
        """`force_state`: dict of [N, .] arrays put into the oracle before stepping (teacher forcing)."""
        if force_state is not None:
            self.b.set_state(force_state)
        if self.episode_length is not None:
            self.steps = np.where(self.done != 0, 0.0, self.steps)
        obs, rew, done, cf, met = self.b.env_step(action, self.track, self.cur_frame, self.n_frames, healthy_z_range=self.z)
        self.cur_frame = cf
        self.reward, self.metrics = rew, met
        if self.episode_length is not None:
            self.steps = self.steps + 1
            over = self.steps >= self.episode_length
            self.truncation = np.where(over, 1 - done, 0.0)
            done = np.where(over, 1.0, done)
            idx = np.nonzero(done)[0]
            if len(idx):
                self.b.set_state(self.first, idx)
                obs[idx] = self.first_obs[idx]
        self.done, self.obs = done, obs
        return obs

    def state(self):
        return self.b.state()
