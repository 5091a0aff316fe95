/* rodent_rr.h -- C ABI of librodent_hip.so: the MI355X-native replacement for the one hot path of
 * talmolab/Brax-Rodent-Run, the batched `Rodent.step()` / `pipeline_step` physics
 * [REF Rodent_Env_Brax.py:87 pipeline_init, :101 pipeline_step, :98-136 step, :138-158 _get_obs].
 *
 * In the reference that path is `PipelineEnv.pipeline_init/pipeline_step` dispatching to the backend
 * module pair `brax.mjx.pipeline.init(sys,q,qd,act,ctrl)` / `step(sys,state,act)` (selected by
 * `backend='mjx'` [REF Rodent_Env_Brax.py:58]).  This header is what a ctypes/cffi binding of a fifth
 * backend would bind (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - Every `float*` / `int32_t*` data argument is a CALLER-OWNED DEVICE pointer (e.g. torch tensor
 *    storage on the batch's HIP device), env-major row-major `[num_envs][width]`, valid until the
 *    work enqueued on the batch's stream has completed.  Nothing is copied to the host.
 *  - All work is enqueued asynchronously on the `hipStream_t` given at rr_batch_create.
 *  - Return value: 0 on success, negative rr_status on error; message via rr_last_error()
 *    (thread-local).  No exceptions cross the ABI; no abort().
 *  - rr_model is immutable and may be shared by batches on several devices; an rr_batch is not
 *    thread-safe.  The library keeps no per-env state: all state lives in the caller's buffers.
 */
#ifndef RODENT_RR_H
#define RODENT_RR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rr_model rr_model;
typedef struct rr_batch rr_batch;

enum rr_status { RR_OK = 0, RR_EINVAL = -1, RR_EIO = -2, RR_EHIP = -3, RR_EUNSUPPORTED = -4 };

/* model dimensions (what `sys.nq/nv/nu`, obs size etc. are in the reference) */
typedef struct rr_dims {
  int32_t nq, nv, nu, na, nbody, njnt, ngeom, nM, ncon, nlimit, nefc, obs_dim;
  int32_t iterations, ls_iterations; /* opt.iterations / opt.ls_iterations [REF Rodent_Env_Brax.py:46-47] */
  int32_t lds_bytes;                 /* dynamic LDS one environment (= one wavefront) uses */
  int32_t dbg_floats;                /* floats per env of the debug dump (rr_debug_layout) */
  float timestep;
  int32_t solver;                    /* 1 = CG, 2 = Newton (mjtSolver; opt.solver [REF Rodent_Env_Brax.py:42-45]) */
  int32_t fixed_instance;            /* 1: stepped by a kernel instance compiled for these dimensions; 0: the generic instance (same results, slower) */
} rr_dims;

/* physics state of the batch: what survives between `pipeline_step` calls (mjx.Data qpos, qvel, act,
 * qacc_warmstart).  All in/out, [N][nq], [N][nv], [N][na], [N][nv]. */
typedef struct rr_state {
  float* qpos;
  float* qvel;
  float* act;
  float* qacc_warmstart;
} rr_state;

/* optional per-step outputs read by the reference env (`data.cinert, data.cvel, data.qfrc_actuator,
 * data.xmat, data.xpos` [REF Rodent_Env_Brax.py:151-155,161]); any pointer may be NULL.
 * Values are those of the LAST forward pass (before the last integration), as in mjx.step. */
typedef struct rr_outputs {
  float* cinert;        /* [N][nbody*10] */
  float* cvel;          /* [N][nbody*6]  */
  float* qfrc_actuator; /* [N][nv]       */
  float* xpos;          /* [N][nbody*3]  */
  float* xmat;          /* [N][nbody*9]  */
  float* subtree_com;   /* [N][3]  (root 0) */
  float* debug;         /* [N][dims.dbg_floats], see rr_debug_layout; NULL in production */
  /* contact geometry of the last forward pass (mjx.Data.contact.{dist,pos,frame}; brax State.contact
   * [NB mjcf.ipynb:917-921]); the integer fields geom1 / geom2 / link_idx are static: rr_model_table */
  float* contact_dist;  /* [N][ncon]   */
  float* contact_pos;   /* [N][ncon*3] */
  float* contact_frame; /* [N][ncon*9]: rows normal, tangent 1, tangent 2 */
} rr_outputs;

/* reference-env epilogue fused into the step kernel [REF Rodent_Env_Brax.py:98-136]; all NULL = physics only */
typedef struct rr_env_io {
  const float* track_pos; /* [T][3] device */
  int32_t track_len;      /* T */
  int32_t* cur_frame;     /* [N] in/out: info['cur_frame'] */
  float* obs;             /* [N][obs_dim] out */
  float* reward;          /* [N] out */
  float* done;            /* [N] out */
  float* metrics;         /* [N][3] out: pos_reward, reward_quadctrl, reward_alive */
  float healthy_reward, ctrl_cost_weight, healthy_z_min, healthy_z_max;
  int32_t terminate_when_unhealthy;
} rr_env_io;

/* -- model ------------------------------------------------------------------------------------- */
/* Load a compiled model blob ('RRM1', written by rodent_amd.mjcf.save_blob).  Replaces
 * mujoco.MjModel.from_xml_path + brax.io.mjcf.load_model -> mjx.put_model [REF Rodent_Env_Brax.py:41,51]. */
int rr_model_load(const char* blob_path, rr_model** out);
int rr_model_dims(const rr_model* m, rr_dims* out);
/* opt.iterations / opt.ls_iterations override [REF Rodent_Env_Brax.py:46-47] (before rr_batch_create) */
int rr_model_set_solver(rr_model* m, int32_t iterations, int32_t ls_iterations);
/* opt.solver = {'cg': 1, 'newton': 2} [REF Rodent_Env_Brax.py:42-45] (before rr_batch_create).  Newton: H = M + J' diag(D active) J
 * has M's tree sparsity for these models and runs through the same factorisation schedules in a second LDS array (csrc/rr_kernel.h
 * newton_hessian); available for the single-rodent models (RR_EUNSUPPORTED otherwise); ~40 KB of LDS per environment. */
int rr_model_set_solver_type(rr_model* m, int32_t solver);
void rr_model_destroy(rr_model* m);
/* Static model tables as the LOADED model holds them (the integer ids the reference exposes through `sys` / `Contact`:
 * "con_geom1", "con_geom2" = Contact.geom1 / geom2 [NB mjcf.ipynb:917-921], "con_body1", "con_body2" (+1 of brax's
 * contact.link_idx), "geom_bodyid", "dof_bodyid", "dof_parentid", "body_parentid", "jnt_qposadr", "jnt_dofadr",
 * "actuator_dofadr", ... -- any entry of the blob by its MuJoCo field name).  Returns a borrowed HOST pointer valid for
 * the model's lifetime; *dtype: 0 = float32, 1 = int32; *count = number of elements.  RR_EINVAL for an unknown name. */
int rr_model_table(const rr_model* m, const char* name, const void** host_ptr, size_t* count, int32_t* dtype);

/* -- batch ------------------------------------------------------------------------------------- */
/* Upload the model tables to `hip_device` and bind `hip_stream` (a hipStream_t, may be NULL = default). */
int rr_batch_create(const rr_model* m, int32_t num_envs, int32_t hip_device, void* hip_stream, rr_batch** out);
void rr_batch_destroy(rr_batch* b);

/* brax.mjx.pipeline.init(sys, q, qd) = make_data + mjx.forward [REF Rodent_Env_Brax.py:87].
 * Reads st->qpos/qvel/act (act normally zero), writes st->qacc_warmstart and the outputs. */
int rr_pipeline_init(rr_batch* b, const rr_state* st, const rr_outputs* out);

/* brax.mjx.pipeline.step(sys, state, act) x n_frames [REF Rodent_Env_Brax.py:101, :53-57]:
 * ctrl [N][nu] is held over the n_frames substeps; st is updated in place. */
int rr_pipeline_step(rr_batch* b, const rr_state* st, const float* ctrl, int32_t n_frames, const rr_outputs* out);

/* Rodent.step fused: pipeline_step + reward/done/obs/metrics + cur_frame increment
 * [REF Rodent_Env_Brax.py:98-136]. `action` plays the role of ctrl. */
int rr_env_step(rr_batch* b, const rr_state* st, const float* action, int32_t n_frames, const rr_env_io* env,
                const rr_outputs* out);

/* Out-of-place variants: read the state from `in` (and info['cur_frame'] from `cur_frame_in`), write the stepped state to
 * `out_state` (and env->cur_frame).  brax states are immutable values [REF Rodent_Env_Brax.py:98-136 returns a new State]:
 * a caller that keeps the previous state (rollout buffers, AutoReset's first state) needs no copies.  `in` and
 * `out_state` may alias field by field (then it is the in-place call). */
int rr_pipeline_step_to(rr_batch* b, const rr_state* in, const rr_state* out_state, const float* ctrl, int32_t n_frames,
                        const rr_outputs* out);
int rr_env_step_to(rr_batch* b, const rr_state* in, const rr_state* out_state, const float* action, int32_t n_frames,
                   const rr_env_io* env, const int32_t* cur_frame_in, const rr_outputs* out);

/* A multi-step rollout in ONE launch: num_steps x [Rodent.step, then EpisodeWrapper + AutoResetWrapper with action_repeat 1] --
 * the `lax.scan` over env.step of brax.training.acting.generate_unroll / a jitted random-action rollout [UP; REF
 * brax_rodent_run_ppo.py:141-151].  actions [num_steps][N][nu].  The environments of the launch do not wait for each other between
 * steps and the state stays on chip; results are those of num_steps calls of rr_env_step_to + rr_wrap_episode_autoreset, bit for
 * bit.  `in` / `out_state` / `env` / `cur_frame_in` as rr_env_step_to (obs, reward, done, metrics, cur_frame of the LAST step are
 * written); `wrap`: the stored first state and first observation (restored where an episode ends), the wrappers' state before the
 * launch (prev_done, steps_in [N]) and after it (steps_out, truncation_out [N]; the final done goes to env->done).
 * Production instance only (no rr_outputs); RR_EUNSUPPORTED for models without a multi-step instance and for the Newton solver. */
/* Models whose contact list is a list of candidate pairs (contacts between two moving bodies, e.g. rodent_cpu.xml [REF models/rodent_cpu.xml]):
 * the kernel keeps the pairs in penetration in 64 contact slots per environment; pairs beyond that are DROPPED for that substep.  *events = the
 * number of (launch, environment) events in which that happened since the batch was created (synchronises the batch's stream); always 0 for
 * the floor-contact models, whose every contact has its own slot. */
int rr_batch_contact_overflow(rr_batch* b, int64_t* events);

/* 1 when this batch's model / solver has a multi-step kernel instance (with_actor != 0: the one with the actor inside), else 0. */
int rr_batch_unroll_supported(const rr_batch* b, int32_t with_actor);
typedef struct rr_unroll_io {
  rr_state first;
  const float* first_obs;
  const float* prev_done;
  const float* steps_in;
  float* steps_out;
  float* truncation_out;
  float episode_length;
} rr_unroll_io;
int rr_env_unroll(rr_batch* b, const rr_state* in, const rr_state* out_state, const float* actions, int32_t num_steps, int32_t n_frames,
                  const rr_env_io* env, const int32_t* cur_frame_in, const rr_unroll_io* wrap);

/* rr_env_unroll with the actor inside: brax.training.acting.generate_unroll -- num_steps x [policy(obs) -> sample -> wrapped env
 * step], transitions recorded -- in ONE launch [UP acting.generate_unroll / actor_step; SURVEY.md a22, a23].  Per env and step the
 * wave evaluates the policy MLP (obs -> 32 x nhidden -> 2A, SiLU; optional normaliser) on the observation of that step, samples
 * the tanh-normal action with the given noise, steps, applies the wrappers, and writes the transition into the learner's trajectory
 * arrays: traj_obs [N][T+1][obs] (row t = observation of step t, row T = the bootstrap observation), traj_raw_action [N][T][A],
 * traj_log_prob / traj_reward / traj_discount (= 1 - done) / traj_truncation [N][T] (with segment_length L < T: U = T / L such
 * blocks one after the other -- a whole rollout phase of U unrolls in one launch); actions_out [T][N][A] receives tanh(raw).
 * Weights: w0 [32][obs] and b0 [32] as torch holds them; hidden_wt[l-1] (l = 1 .. nhidden-1) TRANSPOSED [32 in][32 out];
 * head_wt TRANSPOSED and zero-padded to [32][64], head_b padded to [64]; noise [T][N][A] standard normal draws.
 * The final observation is traj_obs[:, T] (env->obs is not written).  Instances for the single-rodent models (rr_batch_unroll_supported). */
typedef struct rr_actor_io {
  const float* obs_in; const float* mean; const float* std;
  const float* w0; const float* b0;
  const float* hidden_wt[4]; const float* hidden_b[4];
  const float* head_wt; const float* head_b;
  const float* noise;
  float* actions_out;
  float* traj_obs; float* traj_raw_action; float* traj_log_prob; float* traj_reward; float* traj_discount; float* traj_truncation;
  float min_std;
  int32_t nhidden;
  int32_t segment_length;   /* L: num_steps = U * L consecutive steps recorded as U trajectories, the traj_* arrays are [U][N][L(+1)][...]
                             * and the last observation row of a segment is repeated as row 0 of the next; 0 = one segment (L = num_steps) */
} rr_actor_io;
int rr_env_unroll_policy(rr_batch* b, const rr_state* in, const rr_state* out_state, int32_t num_steps, int32_t n_frames, const rr_env_io* env,
                         const int32_t* cur_frame_in, const rr_unroll_io* wrap, const rr_actor_io* actor);

/* obs of Rodent.reset: after rr_pipeline_init, obs = _get_obs(data, 0, cur_frame) [REF :89];
 * implemented as rr_pipeline_init + obs epilogue in one launch. Only env->obs/track_pos/cur_frame are used. */
int rr_env_reset(rr_batch* b, const rr_state* st, const rr_env_io* env, const rr_outputs* out);

/* Generalised advantage estimation of one PPO minibatch, the reverse scan of brax.training.agents.ppo.losses.compute_gae
 * [UP; SURVEY.md Appendix E], one thread per trajectory.  All arrays time-major [T][B] device float32; bootstrap [B];
 * outputs vs [T][B] and advantages [T][B].  `stream` is a hipStream_t (NULL = default stream). */
int rr_compute_gae(const float* truncation, const float* termination, const float* rewards, const float* values,
                   const float* bootstrap_value, int32_t T, int32_t B, float lambda_, float discount, float* vs,
                   float* advantages, void* stream);

/* Fused actor / critic MLP forward on the f32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32; csrc/rr_mlp.h): what
 * `ppo.networks.make_inference_fn` (normalise, policy MLP) and the forward half of `ppo.losses.compute_ppo_loss` (policy and
 * value MLP on the same observations) compute [UP brax.training; REF brax_rodent_run_ppo.py:97-114], for
 * the `make_ppo_networks` default shapes: policy obs -> 32 x (nlayers-1) -> out (<= 64), value obs -> 256 x (nlayers-1) -> 1,
 * SiLU on hidden layers, float32 throughout.  A network is described by HOST arrays of DEVICE pointers: weights[l] is
 * [sizes[l+1]][sizes[l]] row-major (torch.nn.Linear.weight), biases[l] is [sizes[l+1]]; sizes has nlayers + 1 entries.
 * Either network may be NULL (skipped).  obs_rows (device int64 [M], nullable): sample m is row obs_rows[m] of `obs` (a minibatch
 * addressed inside the unroll buffer, no gathered copy).  mean / std (device, [K]) may both be NULL (no normalisation), else
 * x <- (x - mean) / std.  Outputs: policy_out [M][sizes[nlayers]], value_out [M]; optional PRE-activation dumps of the
 * hidden layers (for a backward pass): policy_pre [nlayers-1][M][32], value_pre [nlayers-1][M][256] (NULL = not written).
 * RR_EUNSUPPORTED for other shapes. */
typedef struct rr_mlp_net {
  const float* const* weights;
  const float* const* biases;
  const int32_t* sizes;
  int32_t nlayers;
} rr_mlp_net;
int rr_mlp_forward(const float* obs, const int64_t* obs_rows, int32_t M, int32_t K, const float* mean, const float* std, const rr_mlp_net* policy,
                   const rr_mlp_net* value, float* policy_out, float* value_out, float* policy_pre, float* value_pre, void* stream);


/* The loss half of `brax.training.agents.ppo.losses.compute_ppo_loss` [UP; REF
 * brax_rodent_run_ppo.py:97-114] and its gradient with respect to the network outputs, in three launches without atomics
 * (csrc/rr_ppo.h): truncation-aware GAE, device-local advantage normalisation (population std + 1e-8), tanh-normal log-prob
 * (scale = softplus + min_std), clipped surrogate, value loss 0.25 mean((vs - v)^2), entropy estimated with one normal draw
 * per action dimension.  Inputs (device float32): policy_logits [(T+1)*B][2A] and values [(T+1)*B], the networks' outputs on
 * the gathered minibatch, TIME-major (row t*B + b; rows T*B.. belong to the bootstrap observation); the batch leaves
 * raw_action [R][T][A], log_prob / reward / discount / truncation [R][T] batch-major as collected, addressed through idx [B]
 * (int64 rows of the minibatch; NULL = rows 0..B-1); noise [T*B][A] standard normal draws.  Outputs: grad_logits
 * [(T+1)*B][2A] and grad_values [(T+1)*B] (d total_loss / d output; bootstrap rows zero), metrics[4] = total_loss,
 * policy_loss, v_loss, entropy_loss.  workspace: rr_ppo_loss_workspace_bytes(T, B) bytes of device memory, 8-byte aligned. */
typedef struct rr_ppo_cfg {
  float entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon, min_std;
  int32_t normalize_advantage;
} rr_ppo_cfg;
size_t rr_ppo_loss_workspace_bytes(int32_t T, int32_t B);
int rr_ppo_loss(const float* policy_logits, const float* values, const float* raw_action, const float* log_prob, const float* reward,
                const float* discount, const float* truncation, const int64_t* idx, const float* noise, int32_t T, int32_t B, int32_t A,
                const rr_ppo_cfg* cfg, float* grad_logits, float* grad_values, float* metrics, void* workspace, size_t workspace_bytes,
                void* stream);

/* The rollout's actor step, acting.actor_step -> make_inference_fn [UP; SURVEY.md a22], in two launches: observations (optionally
 * through obs_rows, optionally normalised by mean / std) -> policy network (as rr_mlp_forward takes it: 32-wide hidden layers,
 * head 2 x action_size <= 64) -> tanh-normal head.  noise [M][A] standard normal draws: raw_action = loc + (softplus(scale) + min_std)
 * * noise, action = tanh(raw_action), log_prob [M]; noise NULL: the deterministic policy, action = tanh(loc).  raw_action, log_prob,
 * logits [M][2A] are optional outputs (NULL = not written).  The first layer is split over the observation width across
 * workgroups (the batch alone is 64 row tiles for 256 CUs).  workspace: rr_policy_act_workspace_bytes(M) bytes of device memory. */
size_t rr_policy_act_workspace_bytes(int32_t M);
int rr_policy_act(const float* obs, const int64_t* obs_rows, int32_t M, int32_t K, const float* mean, const float* std,
                  const rr_mlp_net* policy, const float* noise, float min_std, float* action, float* raw_action, float* log_prob,
                  float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* The actor's head on the rollout path, `NormalTanhDistribution` of brax.training.distribution as acting.actor_step uses it [UP;
 * SURVEY.md a22]: logits [N][2A] = (loc | pre-softplus scale), noise [N][A] standard normal draws ->
 * raw_action = loc + (softplus(scale) + min_std) * noise, action = tanh(raw_action), log_prob [N] of raw_action under the
 * distribution (Normal log-density minus the tanh log-det-Jacobian, summed over the action dimensions).  One launch. */
int rr_policy_sample(const float* logits, const float* noise, int32_t N, int32_t A, float min_std, float* action, float* raw_action,
                     float* log_prob, void* stream);

/* Backward pass of the policy network's hidden stack (32-wide SiLU layers) in one launch: delta_{nh-1} = (g W_head) *
 * silu'(z_{nh-1}), delta_{j-1} = (delta_j W_j) * silu'(z_{j-1}).  grad_logits [M][P] (P <= 64) = d loss / d logits; head_weight
 * [P][32]; hidden_weights: HOST array of nhidden device pointers, entry j (1 <= j < nhidden) = W_j [32 out][32 in]
 * (torch.nn.Linear.weight), entry 0 unused; pre_act [nhidden][pre_act_rows >= M][32] = rr_mlp_forward's policy_pre, rows 0..M-1 of
 * each layer overwritten by silu(z) (a minibatch may carry bootstrap rows, which have no policy gradient, behind the M used ones);
 * delta [nhidden][M][32] out; bias_grads: HOST array of nhidden device pointers, db_j [32] (fixed-order sums).  The weight
 * gradients (rr_mlp_weight_grad on these outputs) stay with the caller. */
size_t rr_policy_backward_workspace_bytes(int32_t M, int32_t nhidden);
int rr_policy_backward(const float* grad_logits, const float* head_weight, const float* const* hidden_weights, int32_t nhidden, int32_t M,
                       int32_t P, float* pre_act, int32_t pre_act_rows, float* delta, float* const* bias_grads, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Elementwise half of the backward pass of one hidden SiLU layer of the networks above (the matrix products stay with the
 * caller): with g = delta_l W_l [M][H] and the layer's pre-activations z [M][H] (rr_mlp_forward's dumps),
 * delta = g * silu'(z), h = silu(z) (the operand of dW_l = delta_l' h) and bias_grad[n] = sum_m delta[m][n], in one pass
 * (plus a fixed-order column reduction; no atomics).  `delta` may alias `g` and `h` may alias `z`.  H must divide 256.
 * workspace: rr_mlp_silu_backward_workspace_bytes(M, H) bytes of device memory. */
size_t rr_mlp_silu_backward_workspace_bytes(int32_t M, int32_t H);
int rr_mlp_silu_backward(const float* g, const float* z, int32_t M, int32_t H, float* delta, float* h, float* bias_grad,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Backward pass of the value network's hidden stack (256-wide SiLU layers, as rr_mlp_forward takes them) on the f32 matrix
 * cores: delta_{nh-1} = g w_head * silu'(z_{nh-1}), delta_{j-1} = (delta_j W_j) * silu'(z_{j-1}) with the delta tile resident in
 * LDS between layers.  grad_value [M] = d loss / d value; head_weight [256]; hidden_weights_t: HOST array of nhidden device
 * pointers, entry j (1 <= j < nhidden) = W_j TRANSPOSED ([in][out] row-major), entry 0 unused; pre_act [nhidden][M][256] =
 * rr_mlp_forward's value_pre, overwritten by silu(z) (the operands h_j of dW_{j+1} = delta_{j+1}' h_j); delta [nhidden][M][256]
 * out; bias_grads: HOST array of nhidden device pointers, db_j [256] = column sums of delta_j (fixed-order reduction).  The
 * weight gradients are matrix products of these outputs (dW_0 = delta_0' x) and stay with the caller. */
size_t rr_mlp_value_backward_workspace_bytes(int32_t M, int32_t nhidden);
int rr_mlp_value_backward(const float* grad_value, const float* head_weight, const float* const* hidden_weights_t, int32_t nhidden,
                          int32_t M, float* pre_act, float* delta, float* const* bias_grads, void* workspace, size_t workspace_bytes,
                          void* stream);

/* Weight gradient of one layer, grad[o][i] = sum_m delta[m][o] * x[m][i], on the f32 matrix cores with the row range split
 * over workgroups (the output is small, the reduction ~2e4 rows long) and a fixed-order sum of the partial tiles.  delta
 * [M][O]; act: the layer's input, [M][I] (h = silu(z)) -- or, for the first layer, the RAW observations with act_rows [M]
 * (int64, nullable) = the row of sample m inside `act` (the minibatch addressed in place, no gathered copy) and mean / std
 * [I] (nullable, together with delta_colsum [O] = sum_m delta[m][o], the layer's bias gradient) for x = (act - mean) / std,
 * applied to the sum: grad = (delta' act - delta_colsum mean') / std.  grad [O][I] row-major (torch.nn.Linear.weight.grad).
 * workspace: rr_mlp_weight_grad_workspace_bytes(M, O, I) bytes of device memory. */
size_t rr_mlp_weight_grad_workspace_bytes(int32_t M, int32_t O, int32_t I);
int rr_mlp_weight_grad(const float* delta, const float* act, const int64_t* act_rows, const float* mean, const float* std,
                       const float* delta_colsum, int32_t M, int32_t O, int32_t I, float* grad, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Several weight gradients in one call (up to 12: the eleven layers of the two networks): the products of one tile shape share
 * a launch (item = grid z) and all partial tiles are summed by ONE reduction launch -- 4 launches instead of 22 per minibatch.
 * Items as rr_mlp_weight_grad's arguments. */
typedef struct rr_dw_item {
  const float* delta; const float* act; const int64_t* act_rows; const float* mean; const float* std; const float* delta_colsum;
  int32_t M, O, I;
  float* grad;
} rr_dw_item;
size_t rr_mlp_weight_grad_batch_workspace_bytes(const rr_dw_item* items, int32_t n);
int rr_mlp_weight_grad_batch(const rr_dw_item* items, int32_t n, void* workspace, size_t workspace_bytes, void* stream);

/* The sums of the observation normaliser's update, brax.training.acme.running_statistics.update [UP; SURVEY.md a24, Appendix E;
 * normalize_observations=True in REF brax_rodent_run_ppo.py:103], in one pass over the transitions of a training step: per observation
 * column k,  sums[k] = sum_x (x_k - mean_k)  and  sums[K + k] = sum_x (x_k - mean_k)^2  (double), over the rows t < T of every
 * sequence of `obs` [nseq][Tp1][K] (the unroll buffer: row T of a sequence is the bootstrap observation, not a transition; Tp1 = T for
 * a plain [rows][K] batch).  The update then is  mean' = mean + S1 / count',  summed_variance += S2 - (mean' - mean) S1  -- what upstream
 * forms as sum (x - mean)(x - mean') -- with S1, S2 all-reduced over the ranks.  Fixed-order reductions, no atomics.
 * workspace: rr_obs_moments_workspace_bytes(nseq, T, K) bytes of device memory, 8-byte aligned; sums: device double [2 K]. */
size_t rr_obs_moments_workspace_bytes(int64_t nseq, int32_t T, int32_t K);
int rr_obs_moments(const float* obs, int64_t nseq, int32_t Tp1, int32_t T, int32_t K, const float* mean, double* sums, void* workspace,
                   size_t workspace_bytes, void* stream);

/* brax.envs.wrappers.training.EpisodeWrapper + AutoResetWrapper [UP; SURVEY.md 3.4] after an env step, in one launch:
 * steps' = (prev_done ? 0 : prev_steps) + action_repeat; over = steps' >= episode_length; done <- over ? 1 : done;
 * truncation = over ? 1 - done_env : 0; and for every env with done != 0 the rows of the `narr` (<= 12) arrays `cur[i]`
 * ([N][widths[i]], overwritten in place) are replaced by the rows of `first[i]` (the state stored at reset).  `first`, `cur`,
 * `widths` are HOST arrays of device pointers / ints; everything else is device memory.  `info` is not restored (as upstream). */
int rr_wrap_episode_autoreset(int32_t num_envs, int32_t narr, const float* const* first, float* const* cur, const int32_t* widths,
                              const float* prev_done, const float* prev_steps, float* done, float* steps, float* truncation,
                              float episode_length, float action_repeat, void* stream);

/* Scheduling of the environments on the GPU (results are unaffected: environments are independent).  `env_map` (device,
 * [N] int32, a permutation of 0..N-1, or NULL = identity): workgroup w steps environment env_map[w]; `cost_cycles` (device,
 * [N] uint32, or NULL): every launch writes a work estimate of environment e (line-search point evaluations, the quantity
 * that separates slow from fast environments; NOT cycles, which depend on the SIMD neighbour) to cost_cycles[e].  With N equal to
 * one resident round (2 waves per SIMD: N = 8 x number of CUs) workgroups w and w + N/2 share a SIMD, and a launch lasts as
 * long as its slowest environment: pairing the costliest environments of the previous step with the cheapest ones shortens
 * it (rodent_amd/envs/base.py: PipelineEnv._rebalance).  Both pointers are read at every later launch of this batch. */
int rr_batch_set_schedule(rr_batch* b, const int32_t* env_map, uint32_t* cost_cycles);

/* Debug dump layout: names[i] begins at float offsets[i] of each env's debug row; returns the field count. */
int rr_debug_layout(const rr_batch* b, const char*** names, const int32_t** offsets, const int32_t** sizes);

/* ms of the most recent step-kernel launches on this batch measured with hipEvents on its stream
 * (enable with rr_batch_set_timing(b,1); each launch is then bracketed by events) */
int rr_batch_set_timing(rr_batch* b, int32_t enable);
int rr_batch_kernel_time(rr_batch* b, double* total_ms, int64_t* launches);

/* Diagnostic only (never in a timed run): route launches to the s_memtime-instrumented build of the kernel, which
 * writes per-phase cycle sums of each env into dev_cycles [N][16] (uint64, device).  NULL switches back. */
int rr_batch_set_profile(rr_batch* b, uint64_t* dev_cycles);

/* Build-consistency probe (tests): byte offset and size of the I/O block inside the step kernel's argument segment as the
 * HOST code assumes them (the kernel re-reads that block through the kernarg segment pointer; tools/kernel_meta.py reads the
 * offsets the device compiler actually assigned from the code object's metadata and tests/test_abi_and_oracle.py compares). */
int rr_kernarg_layout(int32_t* io_offset, int32_t* io_size, int32_t* total_size);

const char* rr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
