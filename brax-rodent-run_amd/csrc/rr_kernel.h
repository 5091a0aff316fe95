// rr_kernel.h -- fused per-environment physics step for gfx950 (MI355X): ONE 64-lane wavefront per
// environment, the whole working set of an environment resident in LDS for all n_frames substeps.
//
// Path replaced: brax.mjx.pipeline.step -> mujoco.mjx.step (kinematics, com_pos, crb, factor_m,
// collision, make_constraint, com_vel, passive, rne, fwd_actuation, fwd_acceleration, CG solve, euler)
// [REF Rodent_Env_Brax.py:101 -> UP mjx; SURVEY.md Appendix A], plus the env epilogue
// [REF Rodent_Env_Brax.py:103-158].
//
// Mapping: lanes run over bodies / dofs / contacts (element e -> lane e%64, slot e/64,
// slot counts are template parameters so per-lane arrays stay in registers).  Kinematic-tree
// recursions are pointer-doubling / DFS-range sweeps; the sparse L'DL factorisation and its explicit
// inverse run as list-scheduled row programs (rodent_amd/levelsched.py), the solves and M*x as balanced per-lane jobs
// (rodent_amd/ktables.py); J*x and J'f are Jacobian-free.  No LDS atomics: every update has one owner
// lane.  Reductions over dofs / rows are DPP wave sums in a fixed order.  Inactive constraint rows
// contribute exactly zero in the reference formulation, so they are skipped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RR_LANES 64
#define RR_DOFI 12   // ints per dof in k_dof_i
#define RR_BODYI 12  // ints per body in k_body_i
#ifndef RR_WIDE
#define RR_WIDE 1
#endif
#ifndef RR_RING
#define RR_RING 6    // table rows of a row program in flight (<= ktables RING = 8, the empty rows behind a program); round 2: 8 (12 / 16: 2-3 % slower);
                     // with the shorter round-3 programs 6 is best under the driver protocol (2: +0.2 %, 3-5: +1.0 %, 6: +1.5 % against 8)
#endif

#define RR_NPH 24    // phases of the diagnostic (s_memtime) build
#ifndef RR_W1                // contacts in penetration from which an environment's waves run at priority 1 / 2 / 3
#define RR_W1 2
#define RR_W2 6
#define RR_W3 12
#endif
#ifndef RR_ENV_PRIO
#define RR_ENV_PRIO 1      // graded wave priority by contacts in penetration (see the kernel body)
#endif
#ifndef RR_FACTOR_PRIO
#define RR_FACTOR_PRIO 1   // top priority during the level schedules
#endif
#ifndef RR_DPP_BLOCK
#define RR_DPP_BLOCK 1     // one s_nop per row-broadcast stage instead of one per value (-0.6 % launch time, bit-identical)
#endif
#define RR_MINVAL 1e-15f
#define RR_MINIMP 0.0001f
#define RR_MAXIMP 0.9999f

struct RRDims {
  int nq, nv, nu, nbody, njnt, nM, ncon, dmax, nroot;
  int obs_dim, iterations, ls_iterations, nfac, nround, ninv, lmax, cmax, rmax;
  float dt, gx, gy, gz, tolerance, ls_tolerance, meaninertia;
  // LDS offsets (floats)
  int o_qpos, o_qvel, o_act, o_ctrl, o_xpos, o_xquat, o_cinert, o_cdof, o_cvel,
      o_qLD, o_vec, o_x, o_arm, o_warm, o_qact, o_jlist, lds_floats;
  // Newton instance only (solver == 2): pair arrays of the Hessian and of a copy of M, ancestor-id bytes; 0 otherwise
  int o_H, o_Mp, o_anc, solver;
  // nv_scale: the dof count the solver's tolerances are scaled by (= nv; a PAIR instance runs one replica of nv dofs per wavefront of a
  // 2 nv-dof model).  fac_stride / inv_stride (ints): PAIR instances hold a second copy of each level schedule behind the first, its
  // LDS addresses moved to the second wavefront's region.  lds_bytes_rep: LDS bytes of one replica's region
  int nv_scale, fac_stride, inv_stride, lds_bytes_rep;
  int nalias;      // alias cells (pairs) of the factorisation schedule, at the head of the pose cells (levelsched.py); 0 for the Newton instances
  // debug dump offsets (floats)
  int g_xpos, g_xquat, g_xmat, g_com, g_cinert, g_crb, g_cdof, g_cvel, g_cfrc, g_qM, g_qLD, g_dinv, g_bias, g_passive,
      g_actuator, g_smooth, g_qacc_smooth, g_con_dist, g_con_pos, g_con_frame, g_con_D, g_con_aref, g_lim, g_qacc,
      g_qfrc_constraint, g_misc, g_kaok, dbg_floats;
};

// LDS layout of one environment (float offsets).  One constexpr function serves the host (rr_api.hip layout) and the
// kernel instance compiled for fixed model dimensions.
struct RRLayout {
  int o_qpos, o_qvel, o_act, o_ctrl, o_xpos, o_xquat, o_cinert, o_cdof, o_cvel, o_qLD, o_vec, o_x, o_arm, o_warm, o_qact, o_jlist, lds_floats;
  int o_H, o_Mp, o_anc;
};
constexpr int rr_imax(int a, int b) { return a > b ? a : b; }
constexpr int rr_up4(int n) { return (n + 3) & ~3; }
constexpr RRLayout rr_layout(int nq, int nv, int nu, int nbody, int nM, int ncon, bool newton = false) {
  RRLayout k{};
  int o = 0;
  k.o_qpos = o; o += rr_up4(nq);
  k.o_qvel = o; o += rr_up4(nv + 1);         // cell nv holds 0 (padding of the J*x jobs)
  k.o_act = o; o += rr_up4(nu);
  k.o_ctrl = o; o += rr_up4(nu);
  // pose cells (xpos | xquat) are recycled as the 6*nv scratch of the mass-matrix build and as solver staging
  k.o_xpos = o; o += rr_up4(rr_imax(7 * nbody + 4, 6 * nv));
  k.o_xquat = k.o_xpos + rr_up4(3 * nbody);
  k.o_cinert = o; o += rr_up4(10 * nbody);      // composite inertia accumulates in place
  k.o_cdof = o; o += rr_up4(6 * nv + 6);     // + a zero motion vector for dof id nv (padding of the J*x jobs)
  k.o_cvel = o; o += rr_up4(6 * nbody);
  // sparse-matrix array of PAIRS (M | M + dt*diag(damping), then their factors, then their inverse factors): nM entries,
  // the cells ZERO, ONE, TRASH, MINUS_ONE of the row programs, 16 zero cells padded row jobs read on.  Before the mass
  // matrix is built its first cells hold cacc | cfrc and the sin/cos scratch.
  k.o_qLD = o; o += rr_up4(rr_imax(rr_imax(2 * (nM + 20), 12 * nbody), 2 * nv));
  k.o_vec = o; o += rr_up4(nv + 16);         // vector cells nv.. hold 0 (padding of the job descriptors; padded column steps read on)
  k.o_x = o; o += rr_up4(nv + 16);
  k.o_arm = o; o += rr_up4(2 * nv);
  k.o_warm = o; o += rr_up4(nv);
  k.o_qact = o; o += rr_up4(nv);
  k.o_jlist = o; o += rr_up4(ncon);      // ids of the contacts in penetration, by rank (J*x jobs)
  if (newton) {   // Newton solver: the Hessian H = M + J'DJ has M's tree sparsity (every constraint row lives on ONE ancestor chain),
                  // so it is held, factorised and inverted exactly like M: a second pair array; plus M itself (for M*search)
    k.o_H = o; o += rr_up4(rr_imax(rr_imax(2 * (nM + 20), 12 * nbody), 2 * nv));
    k.o_Mp = o; o += rr_up4(rr_imax(rr_imax(2 * (nM + 20), 12 * nbody), 2 * nv));
    k.o_anc = o; o += rr_up4((nM + 3) / 4 + 1);       // p-th ancestor of dof i at byte Madr[i] + p
  }
  k.lds_floats = o;
  return k;
}

// Dimensions of the benchmark model family (rodent_optimized.xml) as compile-time constants: the instance built on them
// has its loop bounds, bounds checks and LDS addresses as immediates instead of ~40 scalar registers (the generic instance
// spills hundreds of SGPRs to VGPR lanes).  Members of the same name hide the run-time fields of RRDims; everything else
// (solver options, table row counts, debug offsets) stays run-time.  The host selects it only when every constant matches.
template <int NBODY_, int NCON_, int OBS_>
struct RRDimsFixed : RRDims {
  static constexpr int nq = 74, nv = 73, nu = 30, nbody = NBODY_, njnt = 68, nM = 1119, ncon = NCON_, dmax = 35, nroot = 1, obs_dim = OBS_, nround = 6, lmax = 12, cmax = 6, rmax = 3;
  static constexpr RRLayout LY = rr_layout(nq, nv, nu, nbody, nM, ncon);
  static constexpr int o_qpos = LY.o_qpos, o_qvel = LY.o_qvel, o_act = LY.o_act, o_ctrl = LY.o_ctrl, o_xpos = LY.o_xpos, o_xquat = LY.o_xquat,
                       o_cinert = LY.o_cinert, o_cdof = LY.o_cdof, o_cvel = LY.o_cvel, o_qLD = LY.o_qLD, o_vec = LY.o_vec,
                       o_x = LY.o_x, o_arm = LY.o_arm, o_warm = LY.o_warm, o_qact = LY.o_qact, o_jlist = LY.o_jlist,
                       lds_floats = LY.lds_floats;
  __host__ __device__ RRDimsFixed(const RRDims& d) : RRDims(d) {}
  static bool matches(const RRDims& d) {
    const RRDims& r = d;
    return r.nq == nq && r.nv == nv && r.nu == nu && r.nbody == nbody && r.njnt == njnt && r.nM == nM && r.ncon == ncon && r.dmax == dmax &&
           r.nroot == nroot && r.obs_dim == obs_dim && r.nround == nround && r.lmax == lmax && r.cmax == cmax && r.rmax == rmax && r.o_qpos == o_qpos && r.o_qvel == o_qvel && r.o_act == o_act &&
           r.o_ctrl == o_ctrl && r.o_xpos == o_xpos && r.o_xquat == o_xquat && r.o_cinert == o_cinert && r.o_cdof == o_cdof &&
           r.o_cvel == o_cvel && r.o_qLD == o_qLD && r.o_vec == o_vec && r.o_x == o_x && r.o_arm == o_arm &&
           r.o_warm == o_warm && r.o_qact == o_qact && r.o_jlist == o_jlist && r.lds_floats == lds_floats;
  }
};
typedef RRDimsFixed<66, 59, 1263> RRDimsRodent;       // rodent_optimized.xml (the benchmark model)
typedef RRDimsFixed<67, 57, 1279> RRDimsRodentNew;    // rodent_new.xml (the env's default model [REF Rodent_Env_Brax.py:16]) == one replica of rodent_pair.xml

// Table pointers carry the global address space in their type, so every table access is a global_load (never flat).
typedef const int __attribute__((address_space(1)))* rr_gi;
typedef const float __attribute__((address_space(1)))* rr_gf;
struct RRTables {
  rr_gi factor3, linv, coljob, rowjob, jobown, body_i, jnt_i, dof_i, M_ij_k, body_anc, con_chain_rows, con_i, anc4;
  rr_gf body_f, jnt_f, dof_f, act_f, con_f, root_mass;
  rr_gi act_i, act_m_i;      // transmission (DYN instances): per actuator (first moment entry, count); per entry (dof, qpos address)
  rr_gf act_m_f;             // ... coefficient per entry
};

struct RRIO {
  float *qpos, *qvel, *act, *warm;                         // state written by the launch
  const float *qpos_in, *qvel_in, *act_in, *warm_in;       // state read (== the above for an in-place step)
  const int* cur_frame_in;
  const float* ctrl;
  float *o_cinert, *o_cvel, *o_qfrc_actuator, *o_xpos, *o_xmat, *o_com, *dbg;
  float *o_cdist, *o_cpos, *o_cframe;                      // contact geometry of the last forward pass (optional)
  // env epilogue
  const float* track_pos;
  int track_len;
  int* cur_frame;
  float *obs, *reward, *done, *metrics;
  float healthy_reward, ctrl_cost_weight, z_min, z_max;
  int terminate_when_unhealthy;
  unsigned long long* prof;  // diagnostic build only: [N][RR_NPH] cycle sums per phase
  const int* env_map;        // nullable [N]: workgroup -> environment (SIMD pairing of heavy with light environments, rr_batch_set_schedule)
  unsigned* cost;            // nullable [N]: work estimate of this launch per environment (line-search point evaluations x row blocks)
  // multi-step rollout (UNROLL instances, rr_env_unroll): unroll_T env steps in one launch, the Episode + AutoReset training wrappers
  // applied in place between them; ctrl is then [unroll_T][N][nu]
  const float *first_qpos, *first_qvel, *first_act, *first_warm, *first_obs;   // the stored first state (restored where done)
  const float *prev_done, *steps_in;                                           // wrapper state before the launch [N]
  float *steps_out, *trunc_out;                                                // ... and after it
  float episode_length;
  int unroll_T;
  // ... with the actor inside (ACTOR instances, rr_env_unroll_policy): the action of step t is sampled in-kernel from the policy MLP
  // on the observation of step t, and the transitions go straight into the trajectory buffers [N][T(+1)][...] of the learner
  const float *a_obs_in;                          // [N][obs]  observation of the state before the launch
  const float *a_mean, *a_std;                    // nullable: observation normaliser
  const float *a_W0, *a_b0;                       // first layer [32][obs] (torch layout), [32]
  const float *a_Wt[4], *a_b[4];                  // hidden layers l = 1 .. a_nh-1: TRANSPOSED [32 in][32 out], [32]
  const float *a_Wth, *a_bh;                      // head TRANSPOSED and padded [32 in][64], [64]
  const float *a_noise;                           // [T][N][A] standard normal draws
  float *a_actions;                               // [T][N][nu] out: the actions taken (also what the step reads as ctrl)
  float *t_obs, *t_raw, *t_logp, *t_reward, *t_discount, *t_trunc;     // [N][T+1][obs], [N][T][A], [N][T] x 4
  float a_min_std;
  int a_nh;
  int a_seg, a_pad;                               // trajectory segment length L (unroll_T = U * L): the buffers are [U][N][L(+1)][...]
  // PACING of a multi-step launch (nullable): one counter per launch, zeroed by the host; every environment adds 1 per finished env
  // step, so counter / num_envs is the launch's average progress.  An environment behind it by more than a fraction of a step raises
  // its wave priority (Wave::env_prio): the launch ends when its SLOWEST environment does, and a wave that outranks its SIMD partner
  // runs at close to single-wave speed while the partner, which is ahead, has slack.  Timing only -- results are unaffected.
  unsigned* progress;
  unsigned* dyn_overflow;            // DYN instances (nullable): counts (env, substep) events with more pairs in penetration than contact slots
  float pace_t1, pace_t2, pace_t3;   // env steps behind the average for priority levels 1, 2, 3
  int pace_mode;                     // bits 0-1: 0 max(weight level, lag level), 1 lag level only, 2 sum capped at 3; bit 2 (4): progress counted
                                     // per SUBSTEP (ten times finer); bit 3 (8): the factor-phase priority is 3 for laggards, 2 otherwise
  int mode;  // 0 = forward only (pipeline_init), 1 = step; bit 1 (2) = env epilogue as reset (obs only)
  int pad_;
};

// ------------------------------------------------------------------------------------------ small math
struct v3 { float x, y, z; };
__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 ld3(const float __attribute__((address_space(1)))* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

__device__ __forceinline__ void quat_mul(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void quat_to_mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ v3 mat_vec(const float* m, v3 v) {
  return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
__device__ __forceinline__ void quat_normalize(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < RR_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  const float inv = 1.0f / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
// spatial inertia (xx yy zz xy xz yz, m*off(3), m) times motion vector (ang; lin)
__device__ __forceinline__ void mul_inert_vec(float* res, const float* i, const float* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float* res, const float* vel, const float* v) {
  res[0] = -vel[2] * v[1] + vel[1] * v[2];
  res[1] = vel[2] * v[0] - vel[0] * v[2];
  res[2] = -vel[1] * v[0] + vel[0] * v[1];
  res[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  res[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  res[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void cross_force(float* res, const float* vel, const float* f) {
  res[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  res[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  res[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  res[3] = -vel[2] * f[4] + vel[1] * f[5];
  res[4] = vel[2] * f[3] - vel[0] * f[5];
  res[5] = -vel[1] * f[3] + vel[0] * f[4];
}
__device__ __forceinline__ float dot6(const float* a, const float* b) {
  float s = a[0] * b[0];
  s += a[1] * b[1]; s += a[2] * b[2]; s += a[3] * b[3]; s += a[4] * b[4]; s += a[5] * b[5];
  return s;
}
// Wavefront sum, result in every lane.  Four DPP adds give each 16-lane row its total (no LDS crossbar), two row
// broadcasts fold the rows into lane 63, one v_readlane returns it: a fixed order (deterministic), 7 issues per value.
__device__ __forceinline__ float dpp_add(float v, const int ctrl_tag) {
  int x = __float_as_int(v), y;
  switch (ctrl_tag) {
    case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1,0,3,2]
    case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2,3,0,1]
    case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;   // row_half_mirror
    case 3: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break;   // row_mirror: every lane holds its row's sum
    // the two row broadcasts as single DPP adds (the compiler emits v_mov_b32_dpp + v_add_f32 for the masked form):
    // enabled rows add the broadcast lane, the other rows keep their value; s_nop covers the VALU-write -> DPP-read hazard
    case 4: asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v)); return v;   // rows 1, 3 += rows 0, 2
    case 5: asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v)); return v;   // rows 2, 3 += rows 0+1
    default: return v;
  }
  return v + __int_as_float(y);
}
// sum over the 64 lanes, returned wave-uniform: six DPP adds leave the total in lane 63
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3); v = dpp_add(v, 4); v = dpp_add(v, 5);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// the two row-broadcast steps of K >= 3 values as ONE asm block each: a single s_nop covers the VALU-write -> DPP-read hazard of
// the first value, the K - 1 >= 2 instructions in between cover it for the others (dpp_add pays one s_nop per value)
template <int K>
__device__ __forceinline__ void dpp_bcast_stage(float* v) {
  static_assert(K == 3 || K == 4 || K == 6, "instantiated widths");
  if constexpr (K == 3)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
  else if constexpr (K == 4)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
  else
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
}
template <int K>
__device__ __forceinline__ void wave_sum_n(float* v) {
  constexpr bool BLOCK = RR_DPP_BLOCK && (K == 3 || K == 4 || K == 6);
#pragma unroll
  for (int st = 0; st < (BLOCK ? 4 : 6); ++st)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = dpp_add(v[k], st);
  if constexpr (BLOCK) dpp_bcast_stage<K>(v);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
}

// solver impedance [UP mjx constraint._kbi]
__device__ __forceinline__ void kbi(float dt, float sr0, float sr1, const float* si, float pos, float& k, float& b, float& imp) {
  float timeconst = fmaxf(sr0, 2.0f * dt);
  float dmin = fminf(fmaxf(si[0], RR_MINIMP), RR_MAXIMP), dmax = fminf(fmaxf(si[1], RR_MINIMP), RR_MAXIMP);
  float width = fmaxf(RR_MINVAL, si[2]);
  float mid = fminf(fmaxf(si[3], RR_MINIMP), RR_MAXIMP);
  float power = fmaxf(1.0f, si[4]);
  k = 1.0f / (dmax * dmax * timeconst * timeconst * sr1 * sr1);
  b = 2.0f / (dmax * timeconst);
  if (sr0 <= 0) k = -sr0 / (dmax * dmax);
  if (sr1 <= 0) b = -sr1 / dmax;
  float x = fabsf(pos) / width;
  float a_, bb;
  if (power == 2.0f) {            // the exponent of every constraint of the rodent models: squares instead of four powf
    a_ = (1.0f / mid) * (x * x);
    bb = 1.0f - (1.0f / (1.0f - mid)) * ((1.0f - x) * (1.0f - x));
  } else if (power == 1.0f) {
    a_ = x;
    bb = 1.0f - (1.0f - x);
  } else {
    a_ = (1.0f / powf(mid, power - 1.0f)) * powf(x, power);
    bb = 1.0f - (1.0f / powf(1.0f - mid, power - 1.0f)) * powf(1.0f - x, power);
  }
  float y = x < mid ? a_ : bb;
  imp = dmin + y * (dmax - dmin);
  imp = fminf(fmaxf(imp, dmin), dmax);
  if (x > 1.0f) imp = dmax;
}

struct LSPoint { float alpha, cost, d0, d1; };
// a / b by v_rcp_f32 and one Newton step (<= 1 ulp from the IEEE quotient; the IEEE sequence is 13 instructions and the
// line search, which sets the length of the slowest environments, does two per bracketing iteration)
__device__ __forceinline__ float div_nr(float a, float b) { float r = __builtin_amdgcn_rcpf(b); r = r * (2.0f - b * r); return a * r; }

// Table loads in the hot loops must be GLOBAL loads: when the optimiser loses the address space of a table pointer it
// emits flat_load, which also counts on lgkmcnt -- every LDS wait would then drain the table prefetch as well.
__device__ __forceinline__ int g_int(const int __attribute__((address_space(1)))* base, int idx) { return base[idx]; }

// Identity the optimiser cannot see through: stops loop-invariant code motion from unpacking every packed index
// table entry once, ahead of the solver loops, and keeping hundreds of unpacked indices / addresses alive in
// registers (the unpack is 2 VALU ops; the registers are what limits residency).
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// Per-body model constants of the tree sweeps.  For bodies 0..63 (slot 0) they live in registers for the whole
// launch; a body's 2nd/3rd joint (a dozen bodies have one) and bodies >= 64 reload from the L2-resident tables.
struct BodyC {
  int parent, depth, sib, dofadr, dofnum, jn, jadr;
  int jtype0, jqa0, jda0;
  float pos[3], quat[4], jpos0[3], jaxis0[3], jq00;
};
__device__ __forceinline__ BodyC load_bodyc(const RRTables& T, int b, int nbody) {
  BodyC c;
  const bool ok = b >= 1 && b < nbody;
  auto bi = T.body_i + RR_BODYI * (ok ? b : 0);
  auto bf = T.body_f + 18 * (ok ? b : 0);
  c.parent = bi[0]; c.jadr = bi[1]; c.jn = ok ? bi[2] : 0; c.dofadr = bi[3]; c.dofnum = bi[4];
  c.depth = ok ? bi[8] : -1; c.sib = bi[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) c.pos[k] = bf[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) c.quat[k] = bf[3 + k];
  const int j = c.jn > 0 ? c.jadr : 0;
  auto ji = T.jnt_i + 4 * j;
  auto jf = T.jnt_f + 8 * j;
  c.jtype0 = c.jn > 0 ? ji[0] : 3; c.jqa0 = ji[1]; c.jda0 = ji[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) { c.jpos0[k] = jf[k]; c.jaxis0[k] = jf[3 + k]; }
  c.jq00 = jf[6];
  return c;
}

// ------------------------------------------------------------------------------------------ the wave
// ---- diagnostic builds only (tools/rep_bench.py): run a phase REP extra times to read its marginal in-situ cost
#ifndef RR_REP_SOLVE
#define RR_REP_SOLVE 0
#endif
#ifndef RR_REP_LS
#define RR_REP_LS 0
#endif
#ifndef RR_REP_UC
#define RR_REP_UC 0
#endif
#ifndef RR_REP_JAC
#define RR_REP_JAC 0
#endif
#ifndef RR_REP_MM
#define RR_REP_MM 0
#endif
#ifndef RR_REP_KIN
#define RR_REP_KIN 0
#endif
typedef float rr_f2 __attribute__((ext_vector_type(2)));
// PAIR: the model is two identical trees that share no constraint (rodent_pair.xml: M block-diagonal, floor contacts only); the
// workgroup is TWO wavefronts, wave r steps replica r on the tables of one replica in its own LDS region, and the only coupling is the
// CG solver's scalars (cost, gradient norm, line-search sums, Polak-Ribiere beta): every such wave sum is followed by an exchange
// through LDS (solver_sum_n), after which both waves hold the same totals and take the same branches.
// DYN (rodent_cpu.xml): the model's contact list is a list of CANDIDATE pairs of two moving geoms (sphere / capsule); every
// substep the wave scans it, keeps the pairs in penetration in its 64 * NCS contact slots (in list order, by ballot), and works on those:
// J = jac(body2) - jac(body1) through SIGNED dof chains (the dofs on exactly one of the two ancestor chains; J x) and a two-interval
// membership test (J' f); contacts of condim 1 carry one row (rows 1..3 of the slot get D = 0); transmissions with several joints (fixed
// tendons) go through per-actuator sums.  Production physics + env epilogue only.
template <int NBS, int NVS, int NCS, class DT, bool NEWTON = false, bool PAIR = false, bool DYN = false>
struct Wave {
  const DT& D;
  const RRTables& T;
  int lane;               // re-derived (opaquely) at the head of every substep: see RR_FRAME_LOCAL in the kernel
  float* const lds;
  int rep = 0;            // PAIR: this wave's replica (wave-uniform)
  int lag_mode = 0;       // how lag_prio combines with the weight-based level (RRIO::pace_mode)
  int lag_prio = 0;       // multi-step launches: priority level of an environment that is behind the launch's average progress (wave-uniform)
  int xpar = 0;           // PAIR: parity of the next exchange (two buffers: a wave may be one exchange ahead of its partner)
  float* s_xc = nullptr;  // PAIR: exchange cells [2 parities][2 waves][8] behind the two replicas' regions
  // LDS regions.  Aliases (liveness): s_crb == s_cinert (accumulated in place once cinert has been
  // consumed / written out), s_cacc|s_cfrc and the sin/cos scratch live in the region that later holds qLD,
  // s_buf reuses xpos|xquat after the contact geometry has been taken.
  float *s_qpos, *s_qvel, *s_act, *s_ctrl, *s_xpos, *s_xquat, *s_cinert, *s_crb, *s_cdof, *s_cvel, *s_cacc,
      *s_cfrc, *s_buf, *s_sc, *s_qLD, *s_vec, *s_x, *s_arm, *s_warm, *s_qact;
  float *s_H, *s_Mp;      // Newton instance: Hessian pairs, copy of the M pairs
  unsigned char* s_anc;   // Newton instance: ancestor ids along the rows of M
  float dinvH[NVS];       // Newton instance: 1/D of the Hessian's factor
  int* s_jlist;           // contact ids by rank (J*x jobs)

  static constexpr int W = NVS * RR_LANES;
  static constexpr int WC = NCS * RR_LANES;

  // ---- model constants held in registers for the whole launch (loaded once, reused by all substeps)
  BodyC bc0, bc1;         // constants of bodies `lane` (slot 0) and `lane + 64` (slot 1); further slots are loaded at use
  int banc[NBS][2];       // 2^k-th ancestors of the slot's body, k = 0..7, one byte each (0 = none)
  int blast[NBS];         // last body of the subtree (bodies are in DFS order)
  int dofc0[NVS], dofc1[NVS];   // packed per-dof constants: depth | kind<<8 | root<<12 | body<<16 | parent-of-body<<24 ; Madr | last_desc<<16
  // J*x jobs (contact_jobs): the ancestor chains of the contacts in penetration are cut into pieces, one per lane
  static constexpr int JW = DYN ? 10 : 9;       // ints of a chain row (4 ids each)
  int jch[NCS][JW];       // dof ids of this lane's piece, 4 per register (DYN: bit 7 = the dof enters with a minus sign)
  int con_rank[NCS];      // rank of this lane's contact among the contacts in penetration
  int con_leaf[NCS];      // last dof of the contact's chain
  int con_leaf1[NCS];     // DYN: last dof of body1's chain (-1: world)
  int con_pid[NCS];       // DYN: candidate pair held by this slot
  int con_nrow[NCS];      // DYN: rows of the slot's contact (4 = pyramid, 1 = frictionless)
  int dyn_overflow = 0;   // DYN: pairs in penetration beyond the slots (dropped; reported through RRIO::cost bit 31)
  int jP, jLp, jnact;     // wave-uniform: lanes per contact (4 / 2 / 1), ids per piece (12 / 20 / 36), contacts in penetration
  // per-dof registers (slot s -> dof lane + 64 s)
  float dinv[NVS], dinvB[NVS];   // 1/D of M's factor, and of the eulerdamp matrix M + dt*diag(damping)
  float Ma_warm[NVS];            // M * qacc_warmstart, taken before the cells of qM are given to the second factor
  float qfrc_smooth[NVS], qacc_smooth[NVS];
  float qacc[NVS], Ma[NVS], grad[NVS], Mgrad[NVS], search[NVS], mv[NVS], qfrc_con[NVS];
  // limit rows (one per limited hinge dof)
  float lim_sign[NVS], lim_D[NVS], lim_aref[NVS], lim_jar[NVS], lim_jv[NVS];
  bool lim_act[NVS];
  // contact rows (4 pyramid rows per contact slot)
  bool con_act[NCS];
  float con_mu[NCS], con_D[NCS], con_aref[NCS][4], con_jar[NCS][4], con_jv[NCS][4];
  float con_kk[NCS], con_b[NCS];                 // k*imp*dist and b of the contact's reference acceleration
  float con_off[NCS][3], con_fr[NCS][9];         // contact point relative to the tree's COM; frame rows n, t1, t2
  int con_nanc[NCS];
  float com0[3], com1[3];
  float gauss, cost, prev_cost;
  int work;               // wave-uniform count of line-search point evaluations, weighted by row blocks: the scheduling cost estimate (rr_batch_set_schedule)
  unsigned long long pt_last, pt[RR_NPH];
  template <bool PROF> __device__ __forceinline__ void stamp(int i) {
    if (PROF) { const unsigned long long t = __builtin_readcyclecounter(); pt[i] += t - pt_last; pt_last = t; }
  }

  __device__ Wave(const DT& d, const RRTables& t, float* l)
      : D(d), T(t), lane(threadIdx.x & (RR_LANES - 1)), lds(l) {
    s_qpos = l + d.o_qpos; s_qvel = l + d.o_qvel; s_act = l + d.o_act; s_ctrl = l + d.o_ctrl;
    s_xpos = l + d.o_xpos; s_xquat = l + d.o_xquat; s_cinert = l + d.o_cinert; s_crb = s_cinert;
    s_cdof = l + d.o_cdof; s_cvel = l + d.o_cvel; s_qLD = l + d.o_qLD;
    s_cacc = s_qLD; s_cfrc = s_qLD + 6 * d.nbody; s_sc = s_qLD; s_buf = s_xpos;
    s_vec = l + d.o_vec; s_x = l + d.o_x; s_arm = l + d.o_arm; s_warm = l + d.o_warm; s_qact = l + d.o_qact;
    s_jlist = (int*)(l + d.o_jlist);
    s_H = l + d.o_H; s_Mp = l + d.o_Mp; s_anc = (unsigned char*)(l + d.o_anc);
  }

  // One wavefront owns the environment: its LDS instructions execute in program order, so a cross-lane hand-off through LDS
  // needs no s_barrier and no lgkmcnt drain -- only that the compiler keeps the program order of the LDS accesses around this
  // point ("memory" clobber).  Rounds 1-2 also retired the wave's LDS operations here (s_waitcnt lgkmcnt(0), a leftover of the
  // float atomics); without it the results are bit-identical and the launch 1.2 % shorter (tools/variant_bench.py).
#ifndef RR_SYNC_WAIT
#define RR_SYNC_WAIT 0      // 1: also retire the LDS operations at every hand-off (round 1-2 behaviour; bit-identical results, +1.2 % launch time)
#endif
  __device__ __forceinline__ void sync() {
#if RR_SYNC_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ v3 get_com(int r) const {
    return mk3(r ? com1[0] : com0[0], r ? com1[1] : com0[1], r ? com1[2] : com0[2]);
  }
  // wave priority of the throughput-bound phases: by the environment's weight (contacts in penetration), see the kernel body
  __device__ __forceinline__ void env_prio() const {
#if !RR_ENV_PRIO
    __builtin_amdgcn_s_setprio(0);
    return;
#endif
    // ... raised for an environment that has fallen behind the others of a multi-step launch (lag_prio, set by the kernel per env step)
    const int by_weight = jnact >= RR_W3 ? 3 : (jnact >= RR_W2 ? 2 : (jnact >= RR_W1 ? 1 : 0));
    int p = by_weight > lag_prio ? by_weight : lag_prio;
    if ((lag_mode & 3) == 1) p = lag_prio;                                         // progress only
    else if ((lag_mode & 3) == 2) p = by_weight + lag_prio > 3 ? 3 : by_weight + lag_prio;
    if (p >= 3) __builtin_amdgcn_s_setprio(3);          // (s_setprio takes an immediate)
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
  // wave-uniform predicate -> scalar branch
  static __device__ __forceinline__ bool uni(bool p) { return __builtin_amdgcn_readfirstlane((int)p) != 0; }
  // A sum the SOLVER branches on (cost, gradient norm, line-search sums, beta): the wave sum, and for PAIR instances the total of the two
  // replicas' wave sums, formed as [wave 0] + [wave 1] by BOTH waves (identical bits, hence identical branches and matching barriers).
  // One s_barrier per exchange: the cells alternate between two buffers, and a wave can reach exchange k + 2 (same buffer as k) only
  // after its partner has passed the barrier of k + 1, i.e. has read the cells of k.
  template <int K>
  __device__ __forceinline__ void solver_sum_n(float* v) {
    wave_sum_n<K>(v);
    if (PAIR) {
      static_assert(K <= 8, "exchange cells");
      float* mine = s_xc + 16 * xpar + 8 * rep;
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) mine[k] = v[k];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const float* c = s_xc + 16 * xpar;
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = c[k] + c[8 + k];
      xpar ^= 1;
    }
  }
  __device__ __forceinline__ float solver_sum(float v) { float t[1] = {v}; solver_sum_n<1>(t); return t[0]; }

  // ---------------------------------------------------------------- A-1 kinematics (pointer doubling)
  // Every body first builds its LOCAL transform (parent frame -> body, all joints applied) in parallel; world
  // poses are then the prefix products along the tree, formed in ceil(log2(depth)) pointer-doubling rounds
  // T[b] <- T[anc_k(b)] o T[b], anc_{k+1} = anc_k o anc_k (ancestor tables in registers), all bodies busy every
  // round, instead of one serial step per tree level (38-39 of them, a handful of active lanes each).
  // Joint anchors / axes are kept in the parent frame (raw, in the cdof cells) and mapped to the world in com_pos.
  __device__ __forceinline__ BodyC bodyc(int s) const { return s == 0 ? bc0 : (s == 1 ? bc1 : load_bodyc(T, lane + RR_LANES * s, D.nbody)); }
  // k is a run-time round counter: select the word instead of indexing the register array (a dynamic index would
  // push the whole object into scratch memory)
  __device__ __forceinline__ int anc_at(int s, int k) const { const int wd = k < 4 ? banc[s][0] : banc[s][1]; return (wd >> (8 * (k & 3))) & 255; }

  __device__ __forceinline__ void kinematics() {
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      if (s > 0 && !__any(b < D.nbody)) continue;
      const BodyC c = bodyc(s);
      if (b >= 1 && b < D.nbody) {
        float quat[4], mat[9];
        v3 pos = mk3(c.pos[0], c.pos[1], c.pos[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) quat[k] = c.quat[k];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {     // at most 3 joints per body (checked by the model compiler); unrolled so the
          if (jj >= c.jn) break;             // table reads of joints 1, 2 are requested together, ahead of joint 0's arithmetic
          int jt, qa, da;
          v3 jp, ja;
          float q0;
          if (jj == 0) {
            jt = c.jtype0; qa = c.jqa0; da = c.jda0; q0 = c.jq00;
            jp = mk3(c.jpos0[0], c.jpos0[1], c.jpos0[2]); ja = mk3(c.jaxis0[0], c.jaxis0[1], c.jaxis0[2]);
          } else {   // 2nd / 3rd joint of a multi-joint body: parameters from the tables
            auto ji = T.jnt_i + 4 * (c.jadr + jj);
            auto jf = T.jnt_f + 8 * (c.jadr + jj);
            jt = ji[0]; qa = ji[1]; da = ji[2]; q0 = jf[6];
            jp = ld3(jf); ja = ld3(jf + 3);
          }
          if (jt == 0) {  // free joint: the pose is the generalised coordinate itself (parent = world)
            pos = ld3(s_qpos + qa);
#pragma unroll
            for (int k = 0; k < 4; ++k) quat[k] = s_qpos[qa + 3 + k];
            quat_normalize(quat);
          } else {  // hinge
            quat_to_mat(mat, quat);
            const v3 anchor = mat_vec(mat, jp) + pos;
            const v3 axis = mat_vec(mat, ja);
            st3(s_cdof + 6 * da, axis);        // raw, parent frame: axis ; anchor
            st3(s_cdof + 6 * da + 3, anchor);
            const float ang = (s_qpos[qa] - q0) * 0.5f;
            float sn, cs;
            sincosf(ang, &sn, &cs);
            float ql[4] = {cs, ja.x * sn, ja.y * sn, ja.z * sn}, qn[4];
            quat_mul(qn, quat, ql);
#pragma unroll
            for (int k = 0; k < 4; ++k) quat[k] = qn[k];
            quat_to_mat(mat, quat);
            pos = anchor - mat_vec(mat, jp);
          }
        }
        st3(s_xpos + 3 * b, pos);
#pragma unroll
        for (int k = 0; k < 4; ++k) s_xquat[4 * b + k] = quat[k];
      }
    }
    sync();
    for (int k = 0; k < D.nround; ++k) {
      float np[NBS][3], nq[NBS][4];
      bool upd[NBS];
#pragma unroll
      for (int s = 0; s < NBS; ++s) {
        const int b = lane + RR_LANES * s;
        const int a = anc_at(s, k);
        upd[s] = b >= 1 && b < D.nbody && a != 0;
        if (upd[s]) {
          float qa_[4], qb[4], mat[9];
#pragma unroll
          for (int i = 0; i < 4; ++i) { qa_[i] = s_xquat[4 * a + i]; qb[i] = s_xquat[4 * b + i]; }
          quat_to_mat(mat, qa_);
          const v3 p = ld3(s_xpos + 3 * a) + mat_vec(mat, ld3(s_xpos + 3 * b));
          np[s][0] = p.x; np[s][1] = p.y; np[s][2] = p.z;
          quat_mul(nq[s], qa_, qb);
        }
      }
      sync();
#pragma unroll
      for (int s = 0; s < NBS; ++s) {
        if (upd[s]) {
          const int b = lane + RR_LANES * s;
          st3(s_xpos + 3 * b, mk3(np[s][0], np[s][1], np[s][2]));
#pragma unroll
          for (int i = 0; i < 4; ++i) s_xquat[4 * b + i] = nq[s][i];
        }
      }
      sync();
    }
#pragma unroll
    for (int s = 0; s < NBS; ++s) {   // xquat is kept normalised, as the reference does per body
      const int b = lane + RR_LANES * s;
      if (b >= 1 && b < D.nbody) {
        float q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = s_xquat[4 * b + i];
        quat_normalize(q);
#pragma unroll
        for (int i = 0; i < 4; ++i) s_xquat[4 * b + i] = q[i];
      }
    }
    sync();
  }

  // ---------------------------------------------------------------- A-2 com_pos: subtree COM per root, cinert, cdof
  __device__ __forceinline__ void com_pos() {
    float acc[2][3] = {{0, 0, 0}, {0, 0, 0}};
    float xip[NBS][3];
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      xip[s][0] = xip[s][1] = xip[s][2] = 0;
      if (b >= 1 && b < D.nbody) {
        auto bf = T.body_f + 18 * b;
        float bq[4], R[9];
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = s_xquat[4 * b + k];
        quat_to_mat(R, bq);
        v3 xi = ld3(s_xpos + 3 * b) + mat_vec(R, ld3(bf + 7));
        xip[s][0] = xi.x; xip[s][1] = xi.y; xip[s][2] = xi.z;
        const float mass = bf[14];
        if (T.body_i[RR_BODYI * b + 5] == 0) { acc[0][0] += mass * xi.x; acc[0][1] += mass * xi.y; acc[0][2] += mass * xi.z; }
        else                { acc[1][0] += mass * xi.x; acc[1][1] += mass * xi.y; acc[1][2] += mass * xi.z; }
      }
    }
    wave_sum_n<6>(&acc[0][0]);
    {
      const float rm0 = T.root_mass[0], rm1 = D.nroot > 1 ? T.root_mass[1] : 1.0f;
      for (int k = 0; k < 3; ++k) { com0[k] = acc[0][k] / rm0; com1[k] = acc[1][k] / rm1; }
    }
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      if (b >= 1 && b < D.nbody) {
        auto bf = T.body_f + 18 * b;
        float q[4], bq[4], iq[4], R[9];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bq[k] = s_xquat[4 * b + k]; iq[k] = bf[10 + k]; }
        quat_mul(q, bq, iq);
        quat_to_mat(R, q);
        const float mass = bf[14];
        const float I0 = bf[15], I1 = bf[16], I2 = bf[17];
        const v3 cm = get_com(T.body_i[RR_BODYI * b + 5]);
        const float d0 = xip[s][0] - cm.x, d1 = xip[s][1] - cm.y, d2 = xip[s][2] - cm.z;
        float t[9];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) { t[3 * rr] = R[3 * rr] * I0; t[3 * rr + 1] = R[3 * rr + 1] * I1; t[3 * rr + 2] = R[3 * rr + 2] * I2; }
        float* c = s_cinert + 10 * b;
        c[0] = t[0] * R[0] + t[1] * R[1] + t[2] * R[2] + mass * (d1 * d1 + d2 * d2);
        c[1] = t[3] * R[3] + t[4] * R[4] + t[5] * R[5] + mass * (d0 * d0 + d2 * d2);
        c[2] = t[6] * R[6] + t[7] * R[7] + t[8] * R[8] + mass * (d0 * d0 + d1 * d1);
        c[3] = t[0] * R[3] + t[1] * R[4] + t[2] * R[5] - mass * d0 * d1;
        c[4] = t[0] * R[6] + t[1] * R[7] + t[2] * R[8] - mass * d0 * d2;
        c[5] = t[3] * R[6] + t[4] * R[7] + t[5] * R[8] - mass * d1 * d2;
        c[6] = mass * d0; c[7] = mass * d1; c[8] = mass * d2; c[9] = mass;
      }
    }
    // cdof [mju_dofCom]: hinge raw (axis; anchor) is in the parent frame of the joint's body -> world, then
    // (axis; axis x (com - anchor)); free joint: translations along the world axes, rotations about the body axes
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) {
        const int kind = ((opaque(dofc0[s]) >> 8) & 15);
        float* c = s_cdof + 6 * d;
        if (kind < 3) {
          c[0] = c[1] = c[2] = 0;
          c[3] = kind == 0; c[4] = kind == 1; c[5] = kind == 2;
        } else {
          const int fb = kind < 6 ? ((opaque(dofc0[s]) >> 16) & 255) : ((opaque(dofc0[s]) >> 24) & 255);     // frame the raw data is expressed in
          float q[4], R[9];
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = s_xquat[4 * fb + k];
          quat_to_mat(R, q);
          v3 ax, anchor;
          if (kind < 6) {
            ax = mk3(R[kind - 3], R[3 + kind - 3], R[6 + kind - 3]);
            anchor = ld3(s_xpos + 3 * fb);
          } else {
            ax = mat_vec(R, ld3(c));
            anchor = ld3(s_xpos + 3 * fb) + mat_vec(R, ld3(c + 3));
          }
          st3(c, ax);
          st3(c + 3, cross(ax, get_com(((opaque(dofc0[s]) >> 12) & 15)) - anchor));
        }
      }
    }
    sync();
  }

  // ---------------------------------------------------------------- A-6 com_vel + rne (prefix sums over the tree)
  // cvel[b] = sum over the ancestor chain of each body's own sum(cdof * qvel): an inclusive tree prefix sum by the
  // same pointer-doubling rounds as the kinematics; cdof_dot then needs only the parent's cvel (lane-local loop over
  // the body's dofs), and cacc is a second prefix sum of sum(cdof_dot * qvel) on top of the world's -gravity.
  __device__ __forceinline__ void tree_prefix6(float* arr) {
    for (int k = 0; k < D.nround; ++k) {
      float add[NBS][6];
      bool upd[NBS];
#pragma unroll
      for (int s = 0; s < NBS; ++s) {
        const int b = lane + RR_LANES * s;
        const int a = anc_at(s, k);
        upd[s] = b >= 1 && b < D.nbody && a != 0;
        if (upd[s]) {
#pragma unroll
          for (int i = 0; i < 6; ++i) add[s][i] = arr[6 * a + i];
        }
      }
      sync();
#pragma unroll
      for (int s = 0; s < NBS; ++s) {
        if (upd[s]) {
          const int b = lane + RR_LANES * s;
#pragma unroll
          for (int i = 0; i < 6; ++i) arr[6 * b + i] += add[s][i];
        }
      }
      sync();
    }
  }

  __device__ __forceinline__ void velocity_sweep() {
    // own contribution of every body
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      if (s > 0 && !__any(b < D.nbody)) continue;
      const BodyC c = bodyc(s);
      if (b >= 1 && b < D.nbody) {
        float v[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < c.dofnum; ++k) {
          const float qv = s_qvel[c.dofadr + k];
#pragma unroll
          for (int i = 0; i < 6; ++i) v[i] += s_cdof[6 * (c.dofadr + k) + i] * qv;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) s_cvel[6 * b + i] = v[i];
      }
    }
    sync();
    tree_prefix6(s_cvel);
    // cdof_dot (lane-local) -> own acceleration contribution
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      if (s > 0 && !__any(b < D.nbody)) continue;
      const BodyC c = bodyc(s);
      if (b >= 1 && b < D.nbody) {
        const int p = c.parent, da = c.dofadr, dn = c.dofnum;
        float v[6], a[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = s_cvel[6 * p + k];
        const bool is_free = c.jn > 0 && c.jtype0 == 0;
        if (is_free) {
          for (int k = 0; k < 3; ++k) {
            const float qv = s_qvel[da + k];
            for (int i = 0; i < 6; ++i) v[i] += s_cdof[6 * (da + k) + i] * qv;
          }
          for (int k = 0; k < 3; ++k) {   // all three rotational cdof_dot use the velocity after the translations
            float cd[6];
            cross_motion(cd, v, s_cdof + 6 * (da + 3 + k));
            const float qv = s_qvel[da + 3 + k];
            for (int i = 0; i < 6; ++i) a[i] += cd[i] * qv;
          }
        } else {
          for (int k = 0; k < dn; ++k) {
            float cd[6];
            cross_motion(cd, v, s_cdof + 6 * (da + k));
            const float qv = s_qvel[da + k];
            for (int i = 0; i < 6; ++i) { v[i] += s_cdof[6 * (da + k) + i] * qv; a[i] += cd[i] * qv; }
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) s_cacc[6 * b + i] = a[i];
      }
    }
    sync();
    tree_prefix6(s_cacc);
    // body forces: cfrc = I (cacc - g) + cvel x* (I cvel)
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      if (b >= 1 && b < D.nbody) {
        float v[6], a[6], t[6], t1[6], t2[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { v[k] = s_cvel[6 * b + k]; a[k] = s_cacc[6 * b + k]; }
        a[3] -= D.gx; a[4] -= D.gy; a[5] -= D.gz;
        mul_inert_vec(t, s_cinert + 10 * b, v);
        cross_force(t1, v, t);
        mul_inert_vec(t2, s_cinert + 10 * b, a);
#pragma unroll
        for (int k = 0; k < 6; ++k) s_cfrc[6 * b + k] = t2[k] + t1[k];
      }
    }
    sync();
  }

  // ---------------------------------------------------------------- crb + cfrc: subtree sums over DFS ranges
  // Bodies are in depth-first order, so the subtree of b is the contiguous range [b, blast[b]]: every lane sums its
  // own range straight from LDS (no level steps, no hand-offs), keeps the 16 sums in registers, and the results are
  // written back in place after all reads.
  __device__ __forceinline__ void backward_sweep() {
    // Bodies are in DFS order, so a subtree is the contiguous range [b, blast]; the root's range is the whole tree (65
    // terms on one lane while most lanes have a handful).  Two levels: sums of aligned blocks of 8 bodies first (all lanes
    // busy, into the dead pose cells), then every lane adds <= 7 head bodies, <= nbody/8 whole blocks and <= 7 tail bodies.
    constexpr int BS = 8;
    const int nblk = (D.nbody + BS - 1) / BS;
    for (int p = lane; p < 10 * nblk; p += RR_LANES) {
      const int blk = p / 10, k = p - 10 * blk;
      float sum = 0.0f;
#pragma unroll
      for (int j = 0; j < BS; ++j) { const int d = blk * BS + j; if (d >= 1 && d < D.nbody) sum += s_cinert[10 * d + k]; }
      s_buf[16 * blk + k] = sum;
    }
    for (int p = lane; p < 6 * nblk; p += RR_LANES) {
      const int blk = p / 6, k = p - 6 * blk;
      float sum = 0.0f;
#pragma unroll
      for (int j = 0; j < BS; ++j) { const int d = blk * BS + j; if (d >= 1 && d < D.nbody) sum += s_cfrc[6 * d + k]; }
      s_buf[16 * blk + 10 + k] = sum;
    }
    sync();
    float acc[NBS][16];
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[s][k] = 0.0f;
      if (b >= 1 && b < D.nbody) {
        const int e = blast[s] + 1;                                   // range [b, e)
        int d = b;
        const int hb = ((b + BS - 1) & ~(BS - 1)) < e ? ((b + BS - 1) & ~(BS - 1)) : e;
        for (; d < hb; ++d) {
#pragma unroll
          for (int k = 0; k < 10; ++k) acc[s][k] += s_cinert[10 * d + k];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[s][10 + k] += s_cfrc[6 * d + k];
        }
        for (; d + BS <= e; d += BS) {
          const float* S = s_buf + 2 * d;                             // 16 * (d / 8)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[s][k] += S[k];
        }
        for (; d < e; ++d) {
#pragma unroll
          for (int k = 0; k < 10; ++k) acc[s][k] += s_cinert[10 * d + k];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[s][10 + k] += s_cfrc[6 * d + k];
        }
      }
    }
    sync();
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      const int b = lane + RR_LANES * s;
      if (b >= 1 && b < D.nbody) {
#pragma unroll
        for (int k = 0; k < 10; ++k) s_crb[10 * b + k] = acc[s][k];
#pragma unroll
        for (int k = 0; k < 6; ++k) s_cfrc[6 * b + k] = acc[s][10 + k];
      }
    }
    sync();
  }

  static constexpr int NME = NVS == 1 ? 10 : (NVS == 2 ? 18 : 35);   // sparse-M entries per lane (nM <= 64*NME)
  __device__ __forceinline__ void load_ment(int* ment) {
    const int ol = opaque(lane);
#pragma unroll
    for (int it = 0; it < NME; ++it) ment[it] = g_int(T.M_ij_k, ol + RR_LANES * it);
  }
  // ---------------------------------------------------------------- A-3 qM (sparse) from crb and cdof
  __device__ __forceinline__ void mass_matrix() {
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) mul_inert_vec(s_buf + 6 * d, s_crb + 10 * ((opaque(dofc0[s]) >> 16) & 255), s_cdof + 6 * d);
    }
    int ment[NME];          // entry e = lane + 64*it of qM: row i | col j << 8, -1 beyond nM (table padded to whole rows)
    load_ment(ment);
    sync();
#pragma unroll
    for (int it = 0; it < NME; ++it) {
      const int ij = ment[it];
      if (ij >= 0) {
        const int i = ij & 255, j = ij >> 8;
        float v = dot6(s_cdof + 6 * j, s_buf + 6 * i);
        if (i == j) v = s_arm[i] + v;
        *(rr_f2*)(s_qLD + 2 * (lane + RR_LANES * it)) = rr_f2{v, v};     // pair: M | (damped copy, see factor); one 8-byte store
      }
    }
    sync();
  }

  // Executor of the row schedules of the factorisation and the inversion (rodent_amd/levelsched.py).  A table row is 64
  // independent QUAD operations on the sparse-matrix array s_qLD: four targets d_j -= a * b_j [/ piv], b_j four consecutive
  // entries (the targets of one source row are consecutive entries of an ancestor row, so one shared operand, one reciprocal
  // and one run of four feed four multiply-adds), applied as plain read-modify-writes: no two operations of a row share a
  // target.  NO HAND-OFF BETWEEN ROWS: the LDS instructions of a wavefront execute in program order, so the reads of a row see
  // every write of the rows before it, and all reads of a row are issued before its writes -- the host list-schedules the
  // operations of a whole factorisation as one dependency graph under exactly these two rules (rounds 1-2 ran them level by
  // level of the tree with a hand-off per level: 97 + 63 rows for the rodent, now 57 + 44).  The side branches of a fork
  // accumulate into ALIAS copies of their ancestors' rows (pose cells, dead here), which one merge operation per quad folds
  // in before the row is eliminated -- otherwise the free joint's rows, which every dof updates, serialise the schedule.
  // ATOMIC-FREE: an LDS float atomic costs ~10x a plain read-modify-write on gfx950 and >1000 cycles with every wave of the
  // CU issuing them.  PREDICATE-FREE: the host turns the table's element indices into LDS byte addresses at upload, and
  // empty operations / the unused targets of short runs address the cells ZERO (0.0), ONE (1.0) and TRASH kept behind the
  // nM entries (MINUS_ONE (-1.0) is the shared operand of a merge).
  // 1/piv: v_rcp_f32 refined by one Newton step (the row scaling by 1/D after the sweep uses the exact quotient).
  // Rows (16 B per lane) are prefetched RR_RING ahead; the table ends with 8 >= RR_RING empty rows.
  typedef float __attribute__((address_space(3)))* rr_lf;
  static __device__ __forceinline__ float lds_ld(int byte_adr) { return *(rr_lf)(size_t)(unsigned)byte_adr; }
  static __device__ __forceinline__ void lds_st(int byte_adr, float v) { *(rr_lf)(size_t)(unsigned)byte_adr = v; }
  typedef int rr_v4i __attribute__((ext_vector_type(4)));
  // PAIRS: the step factorises two matrices of identical sparsity every substep -- M for the solver and
  // M + dt*diag(damping) for eulerdamp.  They are stored interleaved (float2 per entry), so one 8-byte LDS operation
  // serves both: the table stream, the address extraction and the LDS round trip of a row are shared.
  typedef rr_f2 __attribute__((address_space(3)))* rr_lf2;
  static __device__ __forceinline__ rr_f2 lds_ld2(int byte_adr) { return *(rr_lf2)(size_t)(unsigned)byte_adr; }
  static __device__ __forceinline__ void lds_st2(int byte_adr, rr_f2 v) { *(rr_lf2)(size_t)(unsigned)byte_adr = v; }
  // SHIFT: the schedules' LDS addresses are baked for the pair array at o_qLD; `delta` (bytes) moves them to another pair array
  // (alias-free schedules only: the alias cells are not part of the array)
  struct RowAdr { int a8, p8, b8, d8[4]; };
  template <bool DIV, bool SHIFT>
  static __device__ __forceinline__ RowAdr row_adr(const rr_v4i& e, int delta) {
    const int sh = SHIFT ? delta : 0;
    RowAdr r;
    r.a8 = (e.x & 0xFFFF) + sh; r.b8 = (int)((unsigned)e.x >> 16) + sh;
    r.p8 = DIV ? r.a8 - e.w + 8 : 0;
    r.d8[0] = (e.y & 0xFFFF) + sh; r.d8[1] = (int)((unsigned)e.y >> 16) + sh; r.d8[2] = (e.z & 0xFFFF) + sh; r.d8[3] = (int)((unsigned)e.z >> 16) + sh;
    return r;
  }
  template <bool DIV, bool SHIFT = false>
  __device__ __forceinline__ void run_levels(rr_gi table, int nrows, int delta = 0) {
    typedef const rr_v4i __attribute__((address_space(1)))* rr_gv4;
    typedef const char __attribute__((address_space(1)))* rr_gc;
    const rr_gc tab = (rr_gc)table;                         // wave-uniform base + 16 * lane: the row's address needs no vector arithmetic
    const unsigned lane16 = 16u * (unsigned)lane;
    constexpr int ROWB = RR_LANES * 16;
    rr_v4i ring[RR_RING];
#pragma unroll
    for (int u = 0; u < RR_RING; ++u) ring[u] = *(rr_gv4)(tab + u * ROWB + lane16);
    RowAdr cur = row_adr<DIV, SHIFT>(ring[0], delta);
    for (int r0 = 0; r0 < nrows; r0 += RR_RING) {
      const rr_gc nxt_rows = tab + (size_t)(r0 + RR_RING) * ROWB;
#pragma unroll
#if RR_WIDE == 2
      // EXPERIMENT: table rows in pairs -- the reads of both rows before the writes of either (a row of 128 operations, two per lane)
      for (int u = 0; u < RR_RING; u += 2) {
        if (u && r0 + u >= nrows) break;
        const RowAdr c0 = cur, c1 = row_adr<DIV, SHIFT>(ring[u + 1], delta);
        const rr_f2 va0 = lds_ld2(c0.a8), va1 = lds_ld2(c1.a8);
        rr_f2 vp0 = {1.0f, 1.0f}, vp1 = {1.0f, 1.0f};
        if (DIV) { vp0 = lds_ld2(c0.p8); vp1 = lds_ld2(c1.p8); }
        rr_f2 vb0[4], vo0[4], vb1[4], vo1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vb0[j] = lds_ld2(c0.b8 + 8 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) vo0[j] = lds_ld2(c0.d8[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) vb1[j] = lds_ld2(c1.b8 + 8 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) vo1[j] = lds_ld2(c1.d8[j]);
        ring[u] = *(rr_gv4)(nxt_rows + u * ROWB + lane16);
        ring[u + 1] = *(rr_gv4)(nxt_rows + (u + 1) * ROWB + lane16);
        const RowAdr nxt = row_adr<DIV, SHIFT>(ring[(u + 2) % RR_RING], delta);
        rr_f2 t0 = va0, t1 = va1;
        if (DIV) {
          rr_f2 r0_ = {__builtin_amdgcn_rcpf(vp0.x), __builtin_amdgcn_rcpf(vp0.y)}, r1_ = {__builtin_amdgcn_rcpf(vp1.x), __builtin_amdgcn_rcpf(vp1.y)};
          r0_ = r0_ * (2.0f - vp0 * r0_); r1_ = r1_ * (2.0f - vp1 * r1_);
          t0 *= r0_; t1 *= r1_;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_st2(c0.d8[j], vo0[j] - vb0[j] * t0);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_st2(c1.d8[j], vo1[j] - vb1[j] * t1);
        cur = nxt;
      }
      if (true) continue;
#endif
      for (int u = 0; u < RR_RING; ++u) {
        if (u && r0 + u >= nrows) break;            // wave-uniform: the schedule need not fill its last ring
        const rr_f2 va = lds_ld2(cur.a8);
        rr_f2 vp = {1.0f, 1.0f};
        if (DIV) vp = lds_ld2(cur.p8);
        rr_f2 vb[4], vo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j] = lds_ld2(cur.b8 + 8 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) vo[j] = lds_ld2(cur.d8[j]);
        // under the latency of these reads: the ring's refill and the NEXT row's addresses (ring[u + 1] arrived rows ago)
        ring[u] = *(rr_gv4)(nxt_rows + u * ROWB + lane16);
        const RowAdr nxt = row_adr<DIV, SHIFT>(ring[(u + 1) % RR_RING], delta);
        rr_f2 t = va;
        if (DIV) {
          rr_f2 r = {__builtin_amdgcn_rcpf(vp.x), __builtin_amdgcn_rcpf(vp.y)};
          r = r * (2.0f - vp * r);
          t *= r;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_st2(cur.d8[j], vo[j] - vb[j] * t);
        cur = nxt;
      }
    }
  }

  // sparse L'DL in place in s_qLD  [MuJoCo mj_factorM]; damp = dt for the eulerdamp matrix M + dt*diag(damping).
  // The rank-1 updates M[anc_p][anc_q] -= M[k][anc_p] M[k][anc_q] / M[k][k] of all dofs k, as quad operations in the row order
  // of k_factor3 (run_levels); rows are scaled by 1/D afterwards.
  // Both matrices at once: entry e of the array is the pair (M_e, M_e + dt*damping on the diagonal); mass_matrix has
  // written both halves, the damping is added here.  (M itself is not needed afterwards: the one product with M of the
  // substep, M * qacc_warmstart, is taken before.)
  __device__ __forceinline__ void factor() {
    int ment[NME];          // for the row scaling at the end; requested now, local to this call (not held across the solver)
    load_ment(ment);
    if (lane < 8) s_qLD[2 * D.nM + lane] = (lane >> 1) == 1 ? 1.0f : ((lane >> 1) == 3 ? -1.0f : 0.0f);     // cells ZERO, ONE, TRASH, MINUS_ONE of the row schedules
    for (int c = lane; c < 2 * D.nalias; c += RR_LANES) s_buf[c] = 0.0f;      // alias copies of the hot rows (levelsched.py): pose cells, dead from here to the row scaling
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) s_qLD[2 * (opaque(dofc1[s]) & 0xFFFF) + 1] += D.dt * T.dof_f[16 * d + 1];
    }
    sync();
    run_levels<true>(PAIR ? T.factor3 + rep * D.fac_stride : T.factor3, D.nfac);
    sync();
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      const rr_f2 dd = d < D.nv ? *(const rr_f2*)(s_qLD + 2 * (opaque(dofc1[s]) & 0xFFFF)) : rr_f2{1.0f, 1.0f};
      dinv[s] = d < D.nv ? 1.0f / dd.x : 0.0f;
      dinvB[s] = d < D.nv ? 1.0f / dd.y : 0.0f;
      if (d < D.nv) *(rr_f2*)(s_buf + 2 * d) = rr_f2{dinv[s], dinvB[s]};   // 1/D pairs per dof for the row scaling below (dead pose cells)
    }
    sync();
#pragma unroll
    for (int it = 0; it < NME; ++it) {
      const int ij = ment[it];
      if (ij >= 0) {
        const int i = ij & 255, j = ij >> 8;
        if (i != j) *(rr_f2*)(s_qLD + 2 * (lane + RR_LANES * it)) *= *(const rr_f2*)(s_buf + 2 * i);
      }
    }
    sync();
  }

  // W = I - L^-1 (strictly lower part) in place of L: L^-1 has the tree sparsity of L (non-zero only for j an ancestor of
  // i).  The triangular solves of mj_solveLD are chains of ~2*depth dependent LDS hand-offs each and the step makes 11
  // of them per factorisation; with W they become two independent sparse products (ldl_solve).  Gauss-Jordan, shallow dofs
  // first (k_linv, a row program like the factorisation's): for dof k every descendant row i does W_ia -= W_ik W_ka over the
  // strict ancestors a of k; W_ik must still hold L_ik (deeper dofs overwrite it later) and row k must be final.
  __device__ __forceinline__ void invert() { run_levels<false>(PAIR ? T.linv + rep * D.inv_stride : T.linv, D.ninv); }

  // x <- (L' D L)^-1 x = U D^-1 U' x  [MuJoCo mj_solveLD] with the explicit inverse factor U = I - W (see invert): no
  // dependent chain, no atomics.  U' b sums column j over its descendants i (a contiguous DFS range, entry (i, j) at
  // base[i] - depth[j]); U y sums row i over its ancestors (entry (i, p) at Madr[i] + p).  Both products are nM - nv
  // multiply-adds but the longest column / row is ~nv / ~depth long, so each is cut into balanced pieces of at most
  // D.lmax (<= 16) entries, one piece per lane and job slot (k_coljob / k_rowjob); the piece sums go through the (dead)
  // pose cells s_buf and the owner lane of the column / row adds up its pieces.  Lane d owns x_d.
  static constexpr int NJS = NVS >= 3 ? NVS + 1 : NVS;      // job slots per lane (ktables: nslot)
  // This lane's jobs (rodent_amd/ktables.py), reloaded per call from the L2-resident tables (held across the solver they
  // would be spilled).  PREDICATE-FREE: every job runs D.lmax steps; a column job lists the byte offsets of its matrix
  // entries (padding: the matrix array's ZERO cell) and walks the vector from i0, a row job lists its ancestor dof ids
  // (padding: vector cell nv, which always holds 0) and walks the matrix row from byte offset ra.
  struct Jobs { int cw[NJS][8], ci0[NJS], rb[NJS][4], ra[NJS], own[NVS]; };
  __device__ __forceinline__ void load_jobs(Jobs& j) {
    constexpr int WJ = NJS * RR_LANES;
    const int ol = opaque(lane);
#pragma unroll
    for (int s = 0; s < NJS; ++s) {
#pragma unroll
      for (int k = 0; k < 8; ++k) j.cw[s][k] = 2 * k < D.lmax ? g_int(T.coljob, k * WJ + s * RR_LANES + ol) : 0;
      j.ci0[s] = g_int(T.coljob, 8 * WJ + s * RR_LANES + ol);
#pragma unroll
      for (int k = 0; k < 4; ++k) j.rb[s][k] = 4 * k < D.lmax ? g_int(T.rowjob, k * WJ + s * RR_LANES + ol) : 0;
      j.ra[s] = g_int(T.rowjob, 4 * WJ + s * RR_LANES + ol);
    }
#pragma unroll
    for (int s = 0; s < NVS; ++s) j.own[s] = ol + RR_LANES * s < D.nv ? g_int(T.jobown, ol + RR_LANES * s) : 0;
  }
  // column piece: sum_t mat[cw_t] * vec[i0 + t];  row piece: sum_t mat[ra + t] * vec[rb_t]   (mat = one half of the pair array)
  __device__ __forceinline__ float col_piece(const Jobs& j, int s, const float* mat, const float* vec) const {
    const float* v = vec + j.ci0[s];
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t < D.lmax) acc += *(const float*)((const char*)mat + ((j.cw[s][t >> 1] >> (16 * (t & 1))) & 0xFFFF)) * v[t];
    return acc;
  }
  __device__ __forceinline__ float row_piece(const Jobs& j, int s, const float* mat, const float* vec) const {
    const float* m_ = (const float*)((const char*)mat + j.ra[s]);
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t < D.lmax) acc += m_[2 * t] * vec[(j.rb[s][t >> 2] >> (8 * (t & 3))) & 255];
    return acc;
  }
  // sum of the c <= cmax consecutive piece sums of this lane's column / row (fixed trip count, reads masked by select)
  static __device__ __forceinline__ float merge_pieces(const float* part, int t0, int c, int cmax) {
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < cmax) { const float v = part[t0 + r]; sum += r < c ? v : 0.0f; }
    return sum;
  }
  // DAMPED = true: the factor of M + dt*diag(damping) (second half of the pairs, dinvB)
  template <bool DAMPED = false>
  __device__ __forceinline__ void ldl_solve(float* x) { ldl_solve_on<DAMPED>(x, s_qLD + (DAMPED ? 1 : 0), DAMPED ? dinvB : dinv); }
  // mat: one half of a pair array holding an inverse factor; di: its 1/D
  template <bool DAMPED = false>
  __device__ __forceinline__ void ldl_solve_on(float* x, const float* mat, const float* di) {
    Jobs jb;
    load_jobs(jb);
#pragma unroll
    for (int s = 0; s < NVS; ++s) { const int d = lane + RR_LANES * s; if (d < D.nv) s_x[d] = x[s]; }
    sync();
#pragma unroll
    for (int s = 0; s < NJS; ++s) s_buf[s * RR_LANES + lane] = col_piece(jb, s, mat, s_x);
    sync();
    float y[NVS];
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      y[s] = (x[s] - merge_pieces(s_buf, jb.own[s] & 255, (jb.own[s] >> 8) & 255, D.cmax)) * di[s];
    }
    // every lane has taken its column pieces of x (hand-off above): the vector cells can take y
#pragma unroll
    for (int s = 0; s < NVS; ++s) { const int d = lane + RR_LANES * s; if (d < D.nv) s_x[d] = y[s]; }
    sync();
#pragma unroll
    for (int s = 0; s < NJS; ++s) s_buf[s * RR_LANES + lane] = row_piece(jb, s, mat, s_x);
    sync();
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      x[s] = y[s] - merge_pieces(s_buf, (jb.own[s] >> 16) & 255, (int)((unsigned)jb.own[s] >> 24), D.rmax);
    }
    sync();     // s_buf is reused by the next solve
  }

  // y = M * s_vec (s_vec must be visible).  Row i of the symmetric product is its ancestor part (entries of row i) plus
  // its descendant part (entries (k, i) of the rows below): the two sparse products of ldl_solve on the same vector, with
  // the same balanced jobs (k_coljob / k_rowjob), on the first half of the pairs while it still holds M.  No atomics.
  __device__ __forceinline__ void mul_m(float* y) { mul_m_on(y, s_qLD); }
  __device__ __forceinline__ void mul_m_on(float* y, const float* mp) {
    constexpr int WJ = NJS * RR_LANES;
    Jobs jb;
    load_jobs(jb);
#pragma unroll
    for (int s = 0; s < NJS; ++s) {
      s_buf[s * RR_LANES + lane] = col_piece(jb, s, mp, s_vec);
      s_buf[WJ + s * RR_LANES + lane] = row_piece(jb, s, mp, s_vec);
    }
    sync();
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      float sum = 0.0f;
      if (d < D.nv) {
        sum = mp[2 * (opaque(dofc1[s]) & 0xFFFF)] * s_vec[d];
        sum += merge_pieces(s_buf, jb.own[s] & 255, (jb.own[s] >> 8) & 255, D.cmax);
        sum += merge_pieces(s_buf + WJ, (jb.own[s] >> 16) & 255, (int)((unsigned)jb.own[s] >> 24), D.rmax);
      }
      y[s] = sum;
    }
    sync();
  }

  __device__ __forceinline__ void put_vec(const float* x) {
    sync();  // previous readers of s_vec are done
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) s_vec[d] = x[s];
    }
    sync();
  }

  // ---------------------------------------------------------------- passive + actuation + qfrc_smooth (per dof)
  __device__ __forceinline__ void smooth_forces(float* bias_out, float* passive_out) {
    if (DYN) {      // transmission with several joints per actuator (fixed tendons [REF models/rodent_cpu.xml:505-560]; UP mjx smooth.transmission):
                    // lane = actuator: length = sum coef qpos, velocity = sum coef qvel, force -> the (dead) pose cells, read per dof below
      if (lane < D.nu) {
        const int u = lane, e0 = T.act_i[2 * u], ne = T.act_i[2 * u + 1];
        float len = 0.0f, vel = 0.0f;
        for (int e = 0; e < ne; ++e) {
          const float c = T.act_m_f[e0 + e];
          len += c * s_qpos[T.act_m_i[2 * (e0 + e) + 1]];
          vel += c * s_qvel[T.act_m_i[2 * (e0 + e)]];
        }
        auto af = T.act_f + 8 * u;
        s_buf[u] = af[0] * s_act[u] + af[1] + af[2] * len + af[3] * vel;
      }
      sync();
    }
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      qfrc_smooth[s] = 0.0f;
      bias_out[s] = passive_out[s] = 0.0f;
      if (d < D.nv) {
        auto di = T.dof_i + RR_DOFI * d;
        auto df = T.dof_f + 16 * d;
        const float qv = s_qvel[d];
        float passive = -df[1] * qv;
        if (di[2] == 6) passive -= df[2] * (s_qpos[di[6]] - df[3]);
        const float bias = dot6(s_cdof + 6 * d, s_cfrc + 6 * ((opaque(dofc0[s]) >> 16) & 255));
        float actf = 0.0f;
        const int u = di[7];
        if (u >= 0) {
          if (DYN) {
            actf = df[14] * s_buf[u];          // moment coefficient of this dof x the actuator's force
          } else {
            auto af = T.act_f + 8 * u;
            const float a = s_act[u];
            actf = af[0] * a + af[1] + af[2] * s_qpos[di[6]] + af[3] * qv;
          }
        }
        s_qact[d] = actf;
        qfrc_smooth[s] = passive - bias + actf;
        bias_out[s] = bias; passive_out[s] = passive;
      }
    }
    if (DYN) sync();      // every lane has read the actuator forces before the pose cells are reused
  }

  // ---------------------------------------------------------------- A-4 collision: contact geometry -> registers
  // Runs right after com_pos while xpos / xquat are still live; the Jacobian is never materialised: a contact keeps
  // its offset from the tree COM and its frame, and J x / J' f are evaluated on the fly from cdof (J-free products).
  __device__ __forceinline__ void contact_geometry(float* dbg, float* o_dist = nullptr, float* o_pos = nullptr, float* o_frame = nullptr) {
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const int c = lane + RR_LANES * cs;
      con_act[cs] = false; con_mu[cs] = 0; con_D[cs] = 0; con_nanc[cs] = 0; con_kk[cs] = 0; con_b[cs] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { con_aref[cs][k] = 0; con_jar[cs][k] = 0; con_jv[cs][k] = 0; }
#pragma unroll
      for (int k = 0; k < 3; ++k) con_off[cs][k] = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) con_fr[cs][k] = 0;
      if (c < D.ncon) {
        auto ci = T.con_i + 8 * c;
        auto cf = T.con_f + 26 * c;
        const int kind = ci[0], b = ci[1], r = ci[2];
        float gq[4], bq[4], q[4], gm[9], xm[9];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bq[k] = s_xquat[4 * b + k]; gq[k] = cf[3 + k]; }
        quat_to_mat(xm, bq);
        v3 gp = ld3(s_xpos + 3 * b) + mat_vec(xm, ld3(cf));
        quat_mul(q, bq, gq);
        quat_to_mat(gm, q);
        const v3 size = ld3(cf + 7), n = ld3(cf + 10), pp = ld3(cf + 13);
        v3 fb, pos;
        float dist;
        v3 yb = (n.y > -0.5f && n.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);   // default tangent of make_frame(n)
        if (kind == 3) {  // plane - ellipsoid
          v3 sdir = mk3((gm[0] * n.x + gm[3] * n.y + gm[6] * n.z) * size.x, (gm[1] * n.x + gm[4] * n.y + gm[7] * n.z) * size.y,
                        (gm[2] * n.x + gm[5] * n.y + gm[8] * n.z) * size.z);
          const float nn = sqrtf(dot(sdir, sdir));
          const float inv = nn < RR_MINVAL ? 0.0f : -1.0f / nn;
          v3 sp = mk3(sdir.x * inv * size.x, sdir.y * inv * size.y, sdir.z * inv * size.z);
          v3 pt = mat_vec(gm, sp) + gp;
          dist = dot(n, pt - pp);
          pos = pt - n * (dist * 0.5f);
          v3 bb = yb - n * dot(n, yb);
          fb = bb * (1.0f / sqrtf(dot(bb, bb)));
        } else {
          v3 center = gp;
          const float radius = size.x;
          if (kind == 0) {
            v3 bb = yb - n * dot(n, yb);
            fb = bb * (1.0f / sqrtf(dot(bb, bb)));
          } else {
            const v3 axis = mk3(gm[2], gm[5], gm[8]);
            v3 bb = axis - n * dot(n, axis);
            const float bn = sqrtf(dot(bb, bb));
            fb = bn < 0.5f ? yb : bb * (1.0f / bn);
            center = center + axis * ((kind == 1 ? 1.0f : -1.0f) * size.y);
          }
          dist = dot(center - pp, n) - radius;
          pos = center - n * (radius + 0.5f * dist);
        }
        const v3 fc = cross(n, fb);
        if (dbg) {
          dbg[D.g_con_dist + c] = dist;
          st3(dbg + D.g_con_pos + 3 * c, pos);
          st3(dbg + D.g_con_frame + 9 * c, n); st3(dbg + D.g_con_frame + 9 * c + 3, fb); st3(dbg + D.g_con_frame + 9 * c + 6, fc);
        }
        if (o_dist) o_dist[c] = dist;
        if (o_pos) st3(o_pos + 3 * c, pos);
        if (o_frame) { st3(o_frame + 9 * c, n); st3(o_frame + 9 * c + 3, fb); st3(o_frame + 9 * c + 6, fc); }
        if (dist < 0) {
          con_act[cs] = true;
          float k, bcoef, imp;
          { const float si_[5] = {cf[20], cf[21], cf[22], cf[23], cf[24]}; kbi(D.dt, cf[18], cf[19], si_, dist, k, bcoef, imp); }
          const float rr = fmaxf(cf[17] * (1.0f - imp) / imp, RR_MINVAL);
          con_mu[cs] = cf[16]; con_D[cs] = 1.0f / rr; con_nanc[cs] = ci[4];
          con_kk[cs] = k * imp * dist; con_b[cs] = bcoef;
          const v3 off = pos - get_com(r);
          con_off[cs][0] = off.x; con_off[cs][1] = off.y; con_off[cs][2] = off.z;
          con_fr[cs][0] = n.x; con_fr[cs][1] = n.y; con_fr[cs][2] = n.z;
          con_fr[cs][3] = fb.x; con_fr[cs][4] = fb.y; con_fr[cs][5] = fb.z;
          con_fr[cs][6] = fc.x; con_fr[cs][7] = fc.y; con_fr[cs][8] = fc.z;
        }
        if (dbg) dbg[D.g_con_D + c] = con_D[cs];
      }
    }
  }

  // ---------------------------------------------------------------- DYN: candidate pairs of two moving geoms [UP mjx collision_primitive]
  // closest point of segment [a, b] to pt: t = (pt - a).ab / (ab.ab + 1e-6), clipped  [UP mjx math.closest_segment_point]
  static __device__ __forceinline__ v3 seg_point(v3 a, v3 b, v3 pt) {
    const v3 ab = b - a;
    float t = dot(pt - a, ab) / (dot(ab, ab) + 1e-6f);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return a + ab * t;
  }
  // closest points of two segments [UP mjx math.closest_segment_to_segment_points]: line-line solution from the mid-points (denominator
  // + 1e-6), clipped; then each clipped point re-projected on the other segment and the closer pair kept
  static __device__ __forceinline__ void seg_seg(v3 a0, v3 a1, v3 b0, v3 b1, v3& pa, v3& pb) {
    v3 da = a1 - a0, db = b1 - b0;
    const float la = sqrtf(dot(da, da)), lb = sqrtf(dot(db, db));
    da = da * (la > 0.0f ? 1.0f / la : 0.0f); db = db * (lb > 0.0f ? 1.0f / lb : 0.0f);
    const float ha = 0.5f * la, hb = 0.5f * lb;
    const v3 am = a0 + da * ha, bm = b0 + db * hb, tr = am - bm;
    const float dab = dot(da, db), dat = dot(da, tr), dbt = dot(db, tr);
    float ta = (-dat + dab * dbt) / (1.0f - dab * dab + 1e-6f);
    float tb = dbt + ta * dab;
    ta = fminf(fmaxf(ta, -ha), ha);
    tb = fminf(fmaxf(tb, -hb), hb);
    pa = am + da * ta; pb = bm + db * tb;
    const v3 na = seg_point(a0, a1, pb), nb = seg_point(b0, b1, pa);
    const v3 e1 = na - pb, e2 = nb - pa;
    if (dot(e1, e1) < dot(e2, e2)) pa = na; else pb = nb;
  }
  // pair p: signed distance, contact point, normal (geom1 -> geom2); kinds 4 sphere-sphere, 5 sphere-capsule, 6 capsule-capsule
  __device__ __forceinline__ float pair_geometry(int p, v3& pos, v3& n) const {
    auto ci = T.con_i + 8 * p;
    auto cf = T.con_f + 32 * p;
    const int kind = ci[0], b1 = ci[1], b2 = ci[2];
    v3 c[2], ax[2];
    float rad[2], hl[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int b = g ? b2 : b1;
      float bq[4], gq[4], q[4], xm[9], gm[9];
#pragma unroll
      for (int k = 0; k < 4; ++k) { bq[k] = s_xquat[4 * b + k]; gq[k] = cf[10 * g + 3 + k]; }
      quat_to_mat(xm, bq);
      c[g] = ld3(s_xpos + 3 * b) + mat_vec(xm, ld3(cf + 10 * g));
      quat_mul(q, bq, gq);
      quat_to_mat(gm, q);
      ax[g] = mk3(gm[2], gm[5], gm[8]);
      rad[g] = cf[10 * g + 7]; hl[g] = cf[10 * g + 8];
    }
    v3 pa = c[0], pb = c[1];
    if (kind == 5) pb = seg_point(c[1] - ax[1] * hl[1], c[1] + ax[1] * hl[1], c[0]);
    else if (kind == 6) seg_seg(c[0] - ax[0] * hl[0], c[0] + ax[0] * hl[0], c[1] - ax[1] * hl[1], c[1] + ax[1] * hl[1], pa, pb);
    const v3 d = pb - pa;
    const float len = sqrtf(dot(d, d));
    n = len == 0.0f ? mk3(1, 0, 0) : d * (1.0f / len);
    const float dist = len - (rad[0] + rad[1]);
    pos = pa + n * (rad[0] + 0.5f * dist);
    return dist;
  }
  __device__ __forceinline__ void contact_geometry_dyn() {
    int count = 0;
    for (int p0 = 0; p0 < D.ncon; p0 += RR_LANES) {          // scan: which candidate pairs are in penetration
      const int p = p0 + lane;
      bool pen = false;
      if (p < D.ncon) { v3 pos_, n_; pen = pair_geometry(p, pos_, n_) < 0.0f; }
      const unsigned long long mk = __ballot(pen);
      if (pen) {
        const int slot = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (slot < WC) s_jlist[slot] = p;
      }
      count += __popcll(mk);
    }
    dyn_overflow |= count > WC ? 1 : 0;
    if (count > WC) count = WC;
    sync();
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const int slot = lane + RR_LANES * cs;
      con_act[cs] = false; con_mu[cs] = 0; con_D[cs] = 0; con_nanc[cs] = 0; con_kk[cs] = 0; con_b[cs] = 0;
      con_pid[cs] = 0; con_leaf[cs] = 0; con_leaf1[cs] = -1; con_nrow[cs] = 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) { con_aref[cs][k] = 0; con_jar[cs][k] = 0; con_jv[cs][k] = 0; }
#pragma unroll
      for (int k = 0; k < 3; ++k) con_off[cs][k] = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) con_fr[cs][k] = 0;
      if (slot < count) {
        const int p = s_jlist[slot];
        auto ci = T.con_i + 8 * p;
        auto cf = T.con_f + 32 * p;
        v3 pos, n;
        const float dist = pair_geometry(p, pos, n);
        const v3 yb = (n.y > -0.5f && n.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);       // make_frame(n)
        v3 fb = yb - n * dot(n, yb);
        fb = fb * (1.0f / sqrtf(dot(fb, fb)));
        const v3 fc = cross(n, fb);
        con_act[cs] = true;           // the scan selected it: dist < 0 (same arithmetic)
        con_pid[cs] = p; con_leaf1[cs] = ci[5]; con_leaf[cs] = ci[6]; con_nrow[cs] = ci[7]; con_nanc[cs] = ci[4];
        float k, bcoef, imp;
        { const float si_[5] = {cf[24], cf[25], cf[26], cf[27], cf[28]}; kbi(D.dt, cf[22], cf[23], si_, dist, k, bcoef, imp); }
        const float rr = fmaxf(cf[21] * (1.0f - imp) / imp, RR_MINVAL);
        con_mu[cs] = con_nrow[cs] == 1 ? 0.0f : cf[20]; con_D[cs] = 1.0f / rr;
        con_kk[cs] = k * imp * dist; con_b[cs] = bcoef;
        const v3 off = pos - get_com(ci[3]);
        con_off[cs][0] = off.x; con_off[cs][1] = off.y; con_off[cs][2] = off.z;
        con_fr[cs][0] = n.x; con_fr[cs][1] = n.y; con_fr[cs][2] = n.z;
        con_fr[cs][3] = fb.x; con_fr[cs][4] = fb.y; con_fr[cs][5] = fb.z;
        con_fr[cs][6] = fc.x; con_fr[cs][7] = fc.y; con_fr[cs][8] = fc.z;
      }
    }
    sync();
  }

  // ---------------------------------------------------------------- A-5 constraint rows (limits; contact aref)
  __device__ __forceinline__ void constraint_rows(float* dbg) {
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      lim_act[s] = false; lim_sign[s] = 0; lim_D[s] = 0; lim_aref[s] = 0; lim_jar[s] = 0; lim_jv[s] = 0;
      if (d < D.nv) {
        auto di = T.dof_i + RR_DOFI * d;
        if (di[8]) {
          auto df = T.dof_f + 16 * d;
          const float q = s_qpos[di[6]];
          const float dmin_ = q - df[4], dmax_ = df[5] - q;
          const float pos = fminf(dmin_, dmax_);
          if (pos < 0) {
            float k, b, imp;
            { const float si_[5] = {df[8], df[9], df[10], df[11], df[12]}; kbi(D.dt, df[6], df[7], si_, pos, k, b, imp); }
            const float r = fmaxf(df[13] * (1.0f - imp) / imp, RR_MINVAL);
            lim_act[s] = true;
            lim_sign[s] = dmin_ < dmax_ ? 1.0f : -1.0f;
            lim_D[s] = 1.0f / r;
            lim_aref[s] = -b * (lim_sign[s] * s_qvel[d]) - k * imp * pos;
          }
          if (dbg) { dbg[D.g_lim + 3 * d] = pos; dbg[D.g_lim + 3 * d + 1] = lim_D[s]; dbg[D.g_lim + 3 * d + 2] = lim_aref[s]; }
        }
      }
    }
    float jq[NCS][4];
    jac_mul(jq, s_qvel);      // J * qvel, pyramid rows
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      if (con_act[cs]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) con_aref[cs][k] = -con_b[cs] * jq[cs][k] - con_kk[cs];
      }
      const int c = lane + RR_LANES * cs;
      if (dbg && c < D.ncon) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dbg[D.g_con_aref + 4 * c + k] = con_aref[cs][k];
      }
    }
  }

  // J * vec is needed only for the contacts in penetration (typically 5-15 of ncon), each a walk of up to 36 dofs along
  // the ancestor chain of its body -- with one lane per contact most lanes idle through 36 dependent LDS rounds.  Once
  // per substep the chains of the contacts in penetration are cut into pieces of 12 / 20 / 36 dofs, 4 / 2 / 1 lanes per
  // contact (as many as fit 64 * NCS lanes), the lanes of a contact adjacent so that a quad DPP add joins the pieces.
  __device__ __forceinline__ void contact_jobs() {
    int n_act = 0;
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const unsigned long long m = __ballot(con_act[cs]);
      con_rank[cs] = n_act + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      n_act += __popcll(m);
    }
    jnact = n_act;
    jP = 4 * n_act <= RR_LANES * NCS ? 4 : (2 * n_act <= RR_LANES * NCS ? 2 : 1);
    jLp = jP == 4 ? 12 : (jP == 2 ? 20 : (DYN ? 40 : 36));
    const int sh = jP == 4 ? 2 : (jP == 2 ? 1 : 0);
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const int c = lane + RR_LANES * cs;
      if (DYN) continue;           // DYN: the slots ARE the contacts in penetration (rank = slot) and s_jlist already holds their pair ids
      if (con_act[cs]) s_jlist[con_rank[cs]] = c;
      con_leaf[cs] = c < D.ncon ? (g_int(T.con_chain_rows, 9 * c) & 255) : 0;
    }
    sync();
#pragma unroll
    for (int js = 0; js < NCS; ++js) {
      const int J = lane + RR_LANES * js, r = J >> sh, p = J & (jP - 1);
      const int pad = D.nv * 0x01010101;      // dof id nv: zero motion vector, zero vector cell
#pragma unroll
      for (int k = 0; k < JW; ++k) jch[js][k] = pad;
      if (r < n_act) {
        const int c = s_jlist[r], start = p * jLp;
        const int left = g_int(T.con_i, 8 * c + 4) - start;
        const int n = left < 0 ? 0 : (left > jLp ? jLp : left);
#pragma unroll
        for (int k = 0; k < JW; ++k) {
          if (4 * k < jLp) {
            const int v = g_int(T.con_chain_rows, JW * c + (start >> 2) + k);   // the table has a slack row
            const int keep = n - 4 * k;                                          // ids of this int that belong to the piece
            const unsigned msk = keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : (1u << (8 * keep)) - 1u);
            jch[js][k] = (v & msk) | (pad & ~msk);
          }
        }
      }
    }
  }

  // pyramid rows of J * vec for this lane's contacts, J-free: the spatial velocity of the contact body induced by `vec`
  // (sum of cdof * vec over the ancestor chain, by the jobs above: pieces joined by a quad DPP add, handed to the
  // contact's lane through the dead pose cells), taken at the contact point and projected on the frame
  __device__ __forceinline__ void jac_mul(float (*out)[4], const float* vec) {
    const int sh = jP == 4 ? 2 : (jP == 2 ? 1 : 0);
#pragma unroll
    for (int js = 0; js < NCS; ++js) {
      float w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t0 = 0; t0 < 4 * JW; t0 += 4) {
        if (t0 < jLp) {     // wave-uniform; no per-lane predicate: ids beyond the piece are nv (zero vector)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int id = (opaque(jch[js][t0 >> 2]) >> (8 * u)) & 255;
            const int dd = DYN ? (id & 127) : id;
            float xv = vec[dd];
            if (DYN && (id & 128)) xv = -xv;           // a dof of body1's chain: J = jac(body2) - jac(body1)
            const float* cd = s_cdof + 6 * dd;
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] += cd[i] * xv;
          }
        }
      }
      if (jP >= 2) {
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = dpp_add(w[i], 0);
      }
      if (jP == 4) {
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = dpp_add(w[i], 1);
      }
      const int J = lane + RR_LANES * js;
      if ((J & (jP - 1)) == 0 && (J >> sh) < jnact) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s_buf[6 * (J >> sh) + i] = w[i];
      }
    }
    sync();
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      out[cs][0] = out[cs][1] = out[cs][2] = out[cs][3] = 0.0f;
      if (con_act[cs]) {
        const float* w = s_buf + 6 * con_rank[cs];
        const v3 pv = mk3(w[3], w[4], w[5]) + cross(mk3(w[0], w[1], w[2]), mk3(con_off[cs][0], con_off[cs][1], con_off[cs][2]));
        const float jn_ = dot(mk3(con_fr[cs][0], con_fr[cs][1], con_fr[cs][2]), pv);
        const float j1 = dot(mk3(con_fr[cs][3], con_fr[cs][4], con_fr[cs][5]), pv);
        const float j2 = dot(mk3(con_fr[cs][6], con_fr[cs][7], con_fr[cs][8]), pv);
        const float mu = con_mu[cs];
        out[cs][0] = jn_ + mu * j1; out[cs][1] = jn_ - mu * j1; out[cs][2] = jn_ + mu * j2; out[cs][3] = jn_ - mu * j2;
      }
    }
    sync();     // the pose cells are reused (solve partial sums, line-search staging)
  }

  // constraint state at the current Jaref: forces, qfrc_constraint, cost  [UP mjx solver._update_constraint]
  __device__ __forceinline__ void update_constraint() {
    float part[2] = {0.0f, 0.0f};  // [0] = sum D*Jaref^2 over active rows, [1] = gauss dot
    // J' f without a Jacobian and without atomics: a contact pushes with the spatial force (tau about the tree COM, F)
    // on every dof of its body's ancestor chain, qfrc_d += cdof_d . (tau, F).  The wave walks the contacts that carry
    // force (ballot), broadcasts each one's force and leaf dof (readlane), and every dof lane tests "am I on that chain"
    // by the DFS interval  d <= leaf <= last_desc(d).
    float qc[NVS], cd[NVS][6];
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      qc[s] = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) cd[s][k] = d < D.nv ? s_cdof[6 * d + k] : 0.0f;
    }
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      float W[6] = {0, 0, 0, 0, 0, 0};
      bool has = false;
      if (con_act[cs]) {
        float f[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float jr = con_jar[cs][k];
          if (jr < 0 && (!DYN || k < con_nrow[cs])) { f[k] = -con_D[cs] * jr; part[0] += con_D[cs] * jr * jr; }
        }
        const float mu = con_mu[cs];
        const float fn = f[0] + f[1] + f[2] + f[3], f1 = mu * (f[0] - f[1]), f2 = mu * (f[2] - f[3]);
        if (fn != 0.0f) {
          has = true;
          const v3 F = mk3(con_fr[cs][0], con_fr[cs][1], con_fr[cs][2]) * fn + mk3(con_fr[cs][3], con_fr[cs][4], con_fr[cs][5]) * f1 +
                       mk3(con_fr[cs][6], con_fr[cs][7], con_fr[cs][8]) * f2;
          const v3 tau = cross(mk3(con_off[cs][0], con_off[cs][1], con_off[cs][2]), F);
          W[0] = tau.x; W[1] = tau.y; W[2] = tau.z; W[3] = F.x; W[4] = F.y; W[5] = F.z;
        }
      }
      const int leaf = con_leaf[cs];
      unsigned long long mask = __ballot(has);
      while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        float w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) w[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(W[k]), l));
        const int ld = __builtin_amdgcn_readlane(leaf, l);
        const int ld1 = DYN ? __builtin_amdgcn_readlane(con_leaf1[cs], l) : -1;
#pragma unroll
        for (int s = 0; s < NVS; ++s) {
          if (s > 0 && ld < RR_LANES * s && ld1 < RR_LANES * s) continue;      // wave-uniform: the chains (dofs <= leaf) have no dof in this slot
          const int d = lane + RR_LANES * s;
          if (!DYN) {
            if (d <= ld && ld <= (opaque(dofc1[s]) >> 16))
              qc[s] += cd[s][0] * w[0] + cd[s][1] * w[1] + cd[s][2] * w[2] + cd[s][3] * w[3] + cd[s][4] * w[4] + cd[s][5] * w[5];
          } else {
            const int last = opaque(dofc1[s]) >> 16;
            const bool on2 = d <= ld && ld <= last;
            const bool on1 = ld1 >= 0 && d <= ld1 && ld1 <= last;          // body1's chain: the force enters with a minus sign
            if (on2 != on1) {       // a dof on both chains gets +t - t = 0
              const float t = cd[s][0] * w[0] + cd[s][1] * w[1] + cd[s][2] * w[2] + cd[s][3] * w[3] + cd[s][4] * w[4] + cd[s][5] * w[5];
              qc[s] += on2 ? t : -t;
            }
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      float qcs = qc[s];
      if (lim_act[s] && lim_jar[s] < 0) {
        const float f = -lim_D[s] * lim_jar[s];
        part[0] += lim_D[s] * lim_jar[s] * lim_jar[s];
        qcs += lim_sign[s] * f;
      }
      qfrc_con[s] = qcs;
      part[1] += (Ma[s] - qfrc_smooth[s]) * (qacc[s] - qacc_smooth[s]);
    }
    solver_sum_n<2>(part);
    gauss = 0.5f * part[1];
    prev_cost = cost;
    cost = 0.5f * part[0] + gauss;
  }

  // ---------------------------------------------------------------- Newton solver [UP mjx solver, SolverType.NEWTON; REF Rodent_Env_Brax.py:42-45]
  // H = M + J' diag(D * active) J.  Every constraint row of these models lives on ONE ancestor chain (a contact's body, a limited
  // dof), so J' D J couples a dof only with its own ancestors: H has exactly M's tree sparsity (entry Madr[i] + p <-> dof i and
  // its p-th ancestor) and is held as a second pair array, factorised and inverted by the SAME level schedules (address shift),
  // and solved by the same balanced jobs.  The contacts in penetration are walked like J' f (ballot + readlane broadcast of the
  // contact's frame); every dof lane on the chain forms its four pyramid-row entries, the rows go through the pose cells, and the
  // lane adds D_k r_k[i] r_k[anc_p(i)] along its own row of H.
  __device__ __forceinline__ void newton_hessian() {
    const int RS = (D.nv + 3) & ~3;                      // row stride in the pose cells: 4 rows of nv floats (4 nv <= 7 nbody + 4, host check)
    for (int e = lane; e < D.nM + 20; e += RR_LANES) *(rr_f2*)(s_H + 2 * e) = *(const rr_f2*)(s_Mp + 2 * e);     // H <- M (with the schedule cells)
    float cd[NVS][6];
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
#pragma unroll
      for (int k = 0; k < 6; ++k) cd[s][k] = d < D.nv ? s_cdof[6 * d + k] : 0.0f;
    }
    sync();
#pragma unroll
    for (int s = 0; s < NVS; ++s) {        // limit rows: J = +-e_d
      if (lim_act[s] && lim_jar[s] < 0) { rr_f2* h = (rr_f2*)(s_H + 2 * (opaque(dofc1[s]) & 0xFFFF)); *h += rr_f2{lim_D[s], lim_D[s]}; }
    }
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      unsigned long long mask = __ballot(con_act[cs]);
      while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        const float mu = bc(con_mu[cs]), Dc = bc(con_D[cs]);
        const v3 off = mk3(bc(con_off[cs][0]), bc(con_off[cs][1]), bc(con_off[cs][2]));
        const v3 fn = mk3(bc(con_fr[cs][0]), bc(con_fr[cs][1]), bc(con_fr[cs][2])), f1 = mk3(bc(con_fr[cs][3]), bc(con_fr[cs][4]), bc(con_fr[cs][5])),
                 f2 = mk3(bc(con_fr[cs][6]), bc(con_fr[cs][7]), bc(con_fr[cs][8]));
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = bc(con_jar[cs][k]) < 0 ? Dc : 0.0f;
        const int ld = __builtin_amdgcn_readlane(con_leaf[cs], l);
        float r[NVS][4];
        bool on[NVS];
#pragma unroll
        for (int s = 0; s < NVS; ++s) {
          const int d = lane + RR_LANES * s;
          on[s] = d < D.nv && d <= ld && ld <= (opaque(dofc1[s]) >> 16);
          const v3 jp = mk3(cd[s][3], cd[s][4], cd[s][5]) + cross(mk3(cd[s][0], cd[s][1], cd[s][2]), off);
          const float jn = dot(fn, jp), j1 = dot(f1, jp), j2 = dot(f2, jp);
          r[s][0] = on[s] ? jn + mu * j1 : 0.0f; r[s][1] = on[s] ? jn - mu * j1 : 0.0f;
          r[s][2] = on[s] ? jn + mu * j2 : 0.0f; r[s][3] = on[s] ? jn - mu * j2 : 0.0f;
          if (d < D.nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_buf[k * RS + d] = r[s][k];
          }
        }
        sync();
#pragma unroll
        for (int s = 0; s < NVS; ++s) {
          if (s > 0 && ld < RR_LANES * s) continue;            // wave-uniform: no chain dof in this slot
          if (on[s]) {
            const int madr = opaque(dofc1[s]) & 0xFFFF, dep = opaque(dofc0[s]) & 255;
            const float w0 = a[0] * r[s][0], w1 = a[1] * r[s][1], w2 = a[2] * r[s][2], w3 = a[3] * r[s][3];
            for (int p = 0; p <= dep; ++p) {
              const int j = s_anc[madr + p];
              const float v = w0 * s_buf[j] + w1 * s_buf[RS + j] + w2 * s_buf[2 * RS + j] + w3 * s_buf[3 * RS + j];
              rr_f2* h = (rr_f2*)(s_H + 2 * (madr + p));
              *h += rr_f2{v, v};
            }
          }
        }
        sync();       // the row cells are rewritten by the next contact
      }
    }
  }
  // factor + inverse factor of H in its pair array (both halves hold H), 1/D in dinvH
  __device__ __forceinline__ void newton_factor() {
    const int delta = (int)((D.o_H - D.o_qLD) * sizeof(float));
    int ment[NME];
    load_ment(ment);
    if (lane < 8) s_H[2 * D.nM + lane] = (lane >> 1) == 1 ? 1.0f : 0.0f;     // cells ZERO, ONE, TRASH (+ pad)
    sync();
    run_levels<true, true>(T.factor3, D.nfac, delta);
    sync();
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      dinvH[s] = d < D.nv ? 1.0f / s_H[2 * (opaque(dofc1[s]) & 0xFFFF)] : 0.0f;
      if (d < D.nv) *(rr_f2*)(s_buf + 2 * d) = rr_f2{dinvH[s], dinvH[s]};
    }
    sync();
#pragma unroll
    for (int it = 0; it < NME; ++it) {
      const int ij = ment[it];
      if (ij >= 0) {
        const int i = ij & 255, j = ij >> 8;
        if (i != j) *(rr_f2*)(s_H + 2 * (lane + RR_LANES * it)) *= *(const rr_f2*)(s_buf + 2 * i);
      }
    }
    sync();
    run_levels<false, true>(T.linv, D.ninv, delta);
    sync();
  }

  __device__ __forceinline__ void update_gradient() {
#pragma unroll
    for (int s = 0; s < NVS; ++s) { grad[s] = Ma[s] - qfrc_smooth[s] - qfrc_con[s]; Mgrad[s] = grad[s]; }
    for (int rep = 0; rep < RR_REP_SOLVE; ++rep) { float t_[NVS]; for (int s = 0; s < NVS; ++s) t_[s] = grad[s]; ldl_solve(t_); }
    if (NEWTON) {            // Mgrad = H^-1 grad
      newton_hessian();
      newton_factor();
      ldl_solve_on(Mgrad, s_H, dinvH);
    } else {
      ldl_solve(Mgrad);
    }
  }

  // [UP mjx solver._Context.create]: Jaref, Ma, constraint state (and gradient/search) at `qacc`
  // `at_smooth`: qacc == qacc_smooth, where Ma = M qacc_smooth = qfrc_smooth by construction (no product needed)
  __device__ __forceinline__ void ctx_create(bool at_smooth) {
    put_vec(qacc);
    jac_mul(con_jar, s_vec);
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
      for (int k = 0; k < 4; ++k) con_jar[cs][k] -= con_aref[cs][k];
#pragma unroll
    for (int s = 0; s < NVS; ++s) lim_jar[s] = lim_sign[s] * qacc[s] - lim_aref[s];
    if (at_smooth) {
#pragma unroll
      for (int s = 0; s < NVS; ++s) Ma[s] = qfrc_smooth[s];
    } else {
#pragma unroll
      for (int s = 0; s < NVS; ++s) Ma[s] = Ma_warm[s];     // M * qacc_warmstart (taken before the factorisations)
    }
    cost = INFINITY; prev_cost = 0.0f;
    update_constraint();
  }

  // line-search point(s): cost and derivatives of the piecewise-quadratic 1-D cost at alpha, over the COMPACTED active
  // rows (see linesearch): row r*64 + lane of the first R rows; padding rows are (0, 0, 0) and never active
  static constexpr int KR = 4 * NCS + NVS;     // row slots per lane if every contact / limit row were active
  // COST = false: derivatives only (the bracketing iterations never look at the cost; the costs of the two final points
  // are evaluated once after the loop, at the same alphas, by the same sums -- identical values, 3 fewer wave sums / point)
  template <int NP, bool COST>
  __device__ __forceinline__ void ls_eval(const float* alpha, const float* qg, LSPoint* out, int R, const float* rjr, const float* rjv,
                                          const float* rD) {
    constexpr int NQ = COST ? 3 : 2;
    float q[NQ * NP];
#pragma unroll
    for (int i = 0; i < NQ * NP; ++i) q[i] = 0.0f;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      if (r * RR_LANES < R) {
        const float jr = rjr[r], jv = rjv[r], Dv = rD[r];
        const float q0 = 0.5f * jr * jr * Dv, q1 = jv * jr * Dv, q2 = 0.5f * jv * jv * Dv;
#pragma unroll
        for (int i = 0; i < NP; ++i)
          if (jr + alpha[i] * jv < 0) { q[NQ * i] += q1; q[NQ * i + 1] += q2; if (COST) q[NQ * i + 2] += q0; }
      }
    }
    solver_sum_n<NQ * NP>(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float q1 = qg[1] + q[NQ * i], q2 = qg[2] + q[NQ * i + 1];
      const float a = alpha[i];
      out[i].alpha = a;
      out[i].cost = COST ? a * a * q2 + a * q1 + (qg[0] + q[NQ * i + (COST ? 2 : 0)]) : 0.0f;
      out[i].d0 = 2.0f * a * q2 + q1;
      out[i].d1 = 2.0f * q2 + (q2 == 0.0f ? RR_MINVAL : 0.0f);
    }
  }

  // [UP mjx solver._linesearch].  Only the ACTIVE constraint rows (contacts in penetration x 4 pyramid rows, violated
  // limits) enter the 1-D cost, typically a few dozen of the 4*ncon + nv candidates, scattered over the lanes.  Their
  // (Jaref, Jv, D) triples are compacted once per line search through LDS (positions by ballot / mbcnt; the staging cells
  // are the dead cinert / cvel / pose regions), so that every evaluation of the up to 2 + 3*ls_iterations points costs one
  // row per lane instead of 4*NCS + NVS.
  template <bool PROF>
  __device__ __forceinline__ void linesearch() {
    float red[4] = {0, 0, 0, 0};
    put_vec(search);   // CG: mv = M search is carried by the caller's recurrence; Newton: an explicit product with the copy of M
    if (NEWTON) mul_m_on(mv, s_Mp);
    for (int rep = 0; rep < RR_REP_JAC; ++rep) { float t_[NCS][4]; jac_mul(t_, s_vec); asm volatile("" :: "v"(t_[0][0]), "v"(t_[0][1]), "v"(t_[0][2]), "v"(t_[0][3]) : "memory"); }
    jac_mul(con_jv, s_vec);
    stamp<PROF>(16);
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      lim_jv[s] = lim_sign[s] * search[s];
      red[0] += search[s] * search[s];
      red[1] += search[s] * Ma[s];
      red[2] += search[s] * qfrc_smooth[s];
      red[3] += search[s] * mv[s];
    }
    int R = 0;
    {
      float* const st_jr = s_cinert; float* const st_jv = s_cvel; float* const st_D = s_buf;
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) {
        const unsigned long long m = __ballot(con_act[cs]);
        const int pos = R + 4 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        R += 4 * __popcll(m);
        if (con_act[cs]) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { st_jr[pos + k] = con_jar[cs][k]; st_jv[pos + k] = con_jv[cs][k]; st_D[pos + k] = (!DYN || k < con_nrow[cs]) ? con_D[cs] : 0.0f; }
        }
      }
#pragma unroll
      for (int s = 0; s < NVS; ++s) {
        const unsigned long long m = __ballot(lim_act[s]);
        const int pos = R + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        R += __popcll(m);
        if (lim_act[s]) { st_jr[pos] = lim_jar[s]; st_jv[pos] = lim_jv[s]; st_D[pos] = lim_D[s]; }
      }
    }
    sync();
    float rjr[KR], rjv[KR], rD[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      rjr[r] = rjv[r] = rD[r] = 0.0f;
      if (r * RR_LANES < R) {
        const int idx = r * RR_LANES + lane;
        if (idx < R) { rjr[r] = s_cinert[idx]; rjv[r] = s_cvel[idx]; rD[r] = s_buf[idx]; }
      }
    }
    solver_sum_n<4>(red);
    const float smag = sqrtf(red[0]) * D.meaninertia * (float)D.nv_scale;
    const float gtol = D.tolerance * D.ls_tolerance * smag;
    const float qg[3] = {gauss, red[1] - red[2], 0.5f * red[3]};
    stamp<PROF>(17);
    LSPoint p0, lo, hi, tmp3[3];
    float a1[1] = {0.0f};
    ls_eval<1, true>(a1, qg, &p0, R, rjr, rjv, rD);
    a1[0] = p0.alpha - div_nr(p0.d0, p0.d1);
    ls_eval<1, false>(a1, qg, &lo, R, rjr, rjv, rD);
    if (lo.d0 < p0.d0) { hi = p0; } else { hi = lo; lo = p0; }
    stamp<PROF>(18);
    bool swap = true;
    work += 4 * (1 + (R > RR_LANES ? 1 : 0));          // the four evaluations outside the bracketing loop
    for (int it = 0; it < D.ls_iterations; ++it) {
      work += 3 * (1 + (R > RR_LANES ? 1 : 0));        // the bracketing loop is what separates slow from fast environments
      bool done = !swap;
      done |= (lo.d0 < 0) && (lo.d0 > -gtol);
      done |= (hi.d0 > 0) && (hi.d0 < gtol);
      if (uni(done)) break;
      const float a3[3] = {lo.alpha - div_nr(lo.d0, lo.d1), hi.alpha - div_nr(hi.d0, hi.d1), 0.5f * (lo.alpha + hi.alpha)};
      ls_eval<3, false>(a3, qg, tmp3, R, rjr, rjv, rD);
      const LSPoint lo_next = tmp3[0], hi_next = tmp3[1], mid = tmp3[2];
      const bool swap_lo_next = (lo.d0 > 0) || (lo.d0 < lo_next.d0);
      if (swap_lo_next) lo = lo_next;
      const bool swap_lo_mid = (mid.d0 < 0) && (lo.d0 < mid.d0);
      if (swap_lo_mid) lo = mid;
      const bool swap_hi_next = (hi.d0 < 0) || (hi.d0 > hi_next.d0);
      if (swap_hi_next) hi = hi_next;
      const bool swap_hi_mid = (mid.d0 > 0) && (hi.d0 > mid.d0);
      if (swap_hi_mid) hi = mid;
      swap = swap_lo_next || swap_lo_mid || swap_hi_next || swap_hi_mid;
    }
    {   // costs of the bracket's end points
      const float a2[2] = {lo.alpha, hi.alpha};
      LSPoint f2[2];
      ls_eval<2, true>(a2, qg, f2, R, rjr, rjv, rD);
      lo.cost = f2[0].cost; hi.cost = f2[1].cost;
    }
    stamp<PROF>(19);
    const bool improved = uni((lo.cost < p0.cost) || (hi.cost < p0.cost));
    const float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
    if (improved) {
#pragma unroll
      for (int s = 0; s < NVS; ++s) { qacc[s] += search[s] * alpha; Ma[s] += mv[s] * alpha; lim_jar[s] += lim_jv[s] * alpha; }
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int k = 0; k < 4; ++k) con_jar[cs][k] += con_jv[cs][k] * alpha;
    }
  }

  // [UP mjx solver.solve] primal CG with warm start; returns the iteration count
  template <bool PROF>
  __device__ __forceinline__ int solve() {
    const float scale = 1.0f / (D.meaninertia * (float)D.nv_scale);
    // warm start [UP mjx solver.solve]: cost at qacc_smooth, cost at qacc_warmstart, then the full context at the cheaper
    // of the two.  One copy of the evaluation code, driven by a wave-uniform phase counter.
    float cost_smooth = 0.0f;
    bool use_smooth = false;
#pragma nounroll
    for (int ph = 0; ph < 3; ++ph) {
      if (ph == 2 && !use_smooth) break;        // the context already is the one at qacc_warmstart
#pragma unroll
      for (int s = 0; s < NVS; ++s) qacc[s] = ph != 1 ? qacc_smooth[s] : (lane + RR_LANES * s < D.nv ? s_warm[lane + RR_LANES * s] : 0.0f);
      ctx_create(ph != 1);
      if (ph == 0) cost_smooth = cost;
      if (ph == 1) use_smooth = uni(!(cost < cost_smooth));
    }
    update_gradient();
    // search = -Mgrad, and mv = M search = -grad because Mgrad = M^-1 grad: the product the reference recomputes every
    // iteration [UP mjx solver._linesearch: mv = M @ search] follows the search-direction recurrence exactly
#pragma unroll
    for (int s = 0; s < NVS; ++s) { search[s] = -Mgrad[s]; mv[s] = -grad[s]; }
    stamp<PROF>(8);
    int niter = 0;
    while (true) {
      const float improvement = (prev_cost - cost) * scale;
      float g2 = 0.0f;
#pragma unroll
      for (int s = 0; s < NVS; ++s) g2 += grad[s] * grad[s];
      const float gradient = sqrtf(solver_sum(g2)) * scale;
      bool done = niter >= D.iterations;
      done |= improvement < D.tolerance;
      done |= gradient < D.tolerance;
      if (uni(done)) break;
      stamp<PROF>(12);
      for (int rep = 0; rep < RR_REP_LS; ++rep) linesearch<false>();
      linesearch<PROF>();
      stamp<PROF>(9);
      float pm[NVS], gg = 0.0f;
#pragma unroll
      for (int s = 0; s < NVS; ++s) { gg += grad[s] * Mgrad[s]; pm[s] = Mgrad[s]; }
      for (int rep = 0; rep < RR_REP_UC; ++rep) { const float pc = prev_cost, c0 = cost; update_constraint(); prev_cost = pc; cost = c0; }
      update_constraint();
      stamp<PROF>(10);
      update_gradient();
      stamp<PROF>(11);
      float bt[2] = {0.0f, gg};
#pragma unroll
      for (int s = 0; s < NVS; ++s) bt[0] += grad[s] * (Mgrad[s] - pm[s]);
      solver_sum_n<2>(bt);
      const float beta = NEWTON ? 0.0f : fmaxf(0.0f, bt[0] / fmaxf(RR_MINVAL, bt[1]));      // Newton: search = -Mgrad
#pragma unroll
      for (int s = 0; s < NVS; ++s) { search[s] = -Mgrad[s] + beta * search[s]; mv[s] = -grad[s] + beta * mv[s]; }
      ++niter;
    }
#pragma unroll
    for (int s = 0; s < NVS; ++s) { const int d = lane + RR_LANES * s; if (d < D.nv) s_warm[d] = qacc[s]; }
    return niter;
  }

  // ---------------------------------------------------------------- A-8 euler (+eulerdamp) and position integration
  __device__ __forceinline__ void euler() {
    float qa[NVS];
#pragma unroll
    for (int s = 0; s < NVS; ++s) qa[s] = qfrc_smooth[s] + qfrc_con[s];
    ldl_solve<true>(qa);
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) {
        const int u = DYN ? -1 : T.dof_i[RR_DOFI * d + 7];      // DYN: an actuator may drive several dofs -- integrated per actuator below
        if (u >= 0) {   // filter activation dynamics: act_dot = (clamp(ctrl) - act) / tau
          auto af = T.act_f + 8 * u;
          const float c = fminf(fmaxf(s_ctrl[u], af[5]), af[6]);
          s_act[u] += D.dt * ((c - s_act[u]) / fmaxf(af[4], RR_MINVAL));
        }
        s_qvel[d] += D.dt * qa[s];
      }
    }
    if (DYN && lane < D.nu) {
      auto af = T.act_f + 8 * lane;
      const float c = fminf(fmaxf(s_ctrl[lane], af[5]), af[6]);
      s_act[lane] += D.dt * ((c - s_act[lane]) / fmaxf(af[4], RR_MINVAL));
    }
    sync();
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) {
        const int kind = T.dof_i[RR_DOFI * d + 2], qadr = T.dof_i[RR_DOFI * d + 6];
        if (kind == 6 || kind < 3) {
          s_qpos[qadr] += D.dt * s_qvel[d];
        } else if (kind == 3) {  // quaternion of the free joint: q <- normalize(q * exp(dt*w/2)), w in the body frame
          const v3 w = ld3(s_qvel + d);
          const float n = sqrtf(dot(w, w));
          v3 ax = mk3(0, 0, 0);
          if (n > RR_MINVAL) ax = w * (1.0f / n);
          const float ang = D.dt * n;
          float sn, cs;
          sincosf(ang * 0.5f, &sn, &cs);
          float qr[4] = {cs, ax.x * sn, ax.y * sn, ax.z * sn}, q0[4], qn[4];
          for (int k = 0; k < 4; ++k) q0[k] = s_qpos[qadr + k];
          quat_mul(qn, q0, qr);
          quat_normalize(qn);
          for (int k = 0; k < 4; ++k) s_qpos[qadr + k] = qn[k];
        }
      }
    }
    sync();
  }
};

// ------------------------------------------------------------------------------------------ kernel
// The kernel's explicit arguments as they lie in the kernarg segment.  The ~20 pointers of RRIO are needed only before the
// first and after the last substep; read through the (opaque) kernarg pointer where they are used, they do not occupy
// scalar registers -- or their spill lanes -- during the substeps.
// ASSUMPTION (AMDGPU kernel ABI): explicit by-value arguments are laid out in the kernarg segment in declaration order,
// each at its natural alignment -- i.e. like the members of this struct.  Guards: the static_asserts below pin the struct's
// own layout rules; tests/test_abi_and_oracle.py compares rr_kernarg_layout() with the argument offsets the device compiler
// recorded in the code object (tools/kernel_meta.py); and the debug-dump instance compares the block it re-reads with the
// real `io_kernarg` parameter on every launch (dump field `kernarg_ok`).
struct RRKArgs { RRDims D; RRTables T; RRIO io; int num_envs, n_frames; };
constexpr size_t rr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static_assert(alignof(RRDims) == 4 && alignof(RRTables) == 8 && alignof(RRIO) == 8, "argument alignments the kernarg layout relies on");
static_assert(offsetof(RRKArgs, T) == rr_align_up(sizeof(RRDims), alignof(RRTables)), "RRTables follows RRDims at its natural alignment");
static_assert(offsetof(RRKArgs, io) == rr_align_up(offsetof(RRKArgs, T) + sizeof(RRTables), alignof(RRIO)), "RRIO follows RRTables at its natural alignment");
static_assert(offsetof(RRKArgs, num_envs) == offsetof(RRKArgs, io) + sizeof(RRIO) && sizeof(RRIO) % 8 == 0, "scalars follow RRIO without padding");
static __device__ __forceinline__ RRIO load_io() {
#if defined(__HIP_DEVICE_COMPILE__)
  const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const RRIO __attribute__((address_space(4)))*)(p + offsetof(RRKArgs, io));
#else
  return RRIO{};
#endif
}

// The actor of a multi-step rollout, one wave = one env: policy MLP (obs -> 32 x a_nh -> 2A, SiLU) on the observation of step `ut`,
// tanh-normal sample with the given noise, action into a_actions (where the step reads its ctrl), raw action and log-prob into the
// trajectory buffers.  First layer: lane l holds the normalised observation entries l, l+64, ..; for each of the 32 units the 64
// partial dot products are summed over the wave (four units per DPP reduction).  Later layers: lane n = unit n, the activation
// vector handed around by shuffles, weights transposed so that the lanes read consecutive floats.
// trajectory addressing: step s of the launch is step t = s % L of segment u = s / L; rows [u][env][t]
struct RRTraj { int u, t; };
static __device__ __forceinline__ RRTraj rr_traj(const RRIO& io, int s) { RRTraj r; r.u = s / io.a_seg; r.t = s - r.u * io.a_seg; return r; }
static __device__ __forceinline__ size_t rr_traj_obs(const RRIO& io, int N, int env, int u, int t) { return ((size_t)u * N + env) * (io.a_seg + 1) + t; }   // row index
static __device__ __forceinline__ size_t rr_traj_at(const RRIO& io, int N, int env, int s) { const RRTraj r = rr_traj(io, s); return ((size_t)r.u * N + env) * io.a_seg + r.t; }
static __device__ __forceinline__ float rr_softplus_k(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
template <class DT>
__device__ __forceinline__ void rr_actor_step(const RRIO& io, const DT& D, int lane, int env, int ut, int num_envs) {
  constexpr int JM = 20;                      // observation entries per lane (host check: obs_dim <= 1280)
  const int K = D.obs_dim, A = D.nu;
  const RRTraj tr = rr_traj(io, ut);
  const float* ob = io.t_obs + rr_traj_obs(io, num_envs, env, tr.u, tr.t) * K;
  float x[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    const int k = lane + RR_LANES * j, kc = k < K ? k : K - 1;
    float v = ob[kc];
    if (io.a_mean) v = (v - io.a_mean[kc]) / io.a_std[kc];
    x[j] = k < K ? v : 0.0f;
  }
  // all 32 units at once: every observation entry meets 32 independent loads (one per unit, 256 contiguous bytes each across the
  // wave), so the memory system always has a full batch in flight; the 32 wave sums follow, four per DPP reduction
  float p[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) p[n] = 0.0f;
#pragma unroll 2
  for (int j = 0; j < JM; ++j) {
    const int k = lane + RR_LANES * j, kc = k < K ? k : K - 1;
    const float xj = x[j];
#pragma unroll
    for (int n = 0; n < 32; ++n) p[n] = fmaf(xj, io.a_W0[(size_t)n * K + kc], p[n]);
  }
  float z1 = 0.0f;
#pragma unroll
  for (int n0 = 0; n0 < 32; n0 += 4) {
    wave_sum_n<4>(p + n0);
#pragma unroll
    for (int u = 0; u < 4; ++u) z1 = lane == n0 + u ? p[n0 + u] : z1;
  }
  const int l32 = lane & 31;
  float h = z1 + io.a_b0[l32];
  h = h / (1.0f + expf(-h));
#pragma unroll
  for (int l = 1; l < 5; ++l) {             // constant indices into the kernel-argument arrays (they live in scalar registers)
    if (l < io.a_nh) {
      const float* wt = io.a_Wt[l - 1];
      float acc = io.a_b[l - 1][l32];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc = fmaf(__shfl(h, k, RR_LANES), wt[k * 32 + l32], acc);
      h = acc / (1.0f + expf(-acc));
    }
  }
  float o = io.a_bh[lane];
#pragma unroll
  for (int k = 0; k < 32; ++k) o = fmaf(__shfl(h, k, RR_LANES), io.a_Wth[k * 64 + lane], o);
  const float s_raw = __shfl(o, (A + lane) & 63, RR_LANES);      // lane a: logits[A + a]
  float lp = 0.0f;
  if (lane < A) {
    const float HALF_LOG_2PI = 0.91893853320467274178f, LOG2 = 0.69314718055994530942f;
    const float scale = rr_softplus_k(s_raw) + io.a_min_std;
    const float raw = o + scale * io.a_noise[((size_t)ut * num_envs + env) * A + lane];
    const float zz = (raw - o) / scale;
    lp = -0.5f * zz * zz - logf(scale) - HALF_LOG_2PI - 2.0f * (LOG2 - raw - rr_softplus_k(-2.0f * raw));
    io.a_actions[((size_t)ut * num_envs + env) * A + lane] = tanhf(raw);
    io.t_raw[rr_traj_at(io, num_envs, env, ut) * A + lane] = raw;
  }
  lp = wave_sum(lp);
  if (lane == 0) io.t_logp[rr_traj_at(io, num_envs, env, ut)] = lp;
}

// UNROLL: io.unroll_T env steps per launch.  The environments of a launch never wait for each other between steps (a synchronised
// step lasts as long as its slowest environment; over ten unsynchronised steps the slowest SUM is 6.5 % below ten slowest steps,
// tools/tail_probe.py), the state stays in LDS from step to step, and the Episode + AutoReset wrappers
// (brax.envs.wrappers.training; rr_wrap_episode_autoreset is their one-launch form) are applied in place.
// PAIR (see Wave): 128 threads = one wavefront per replica of a two-tree model; Dk / T describe ONE replica (nv_scale = the model's
// dof count), the state arrays are the model's ([N][2 nq] ...: replica r of environment e is row 2 e + r of an [2 N][nq] array).
// Physics only (pipeline_init / pipeline_step: no env epilogue, no optional outputs, no debug dump).
template <int NBS, int NVS, int NCS, bool PROF, bool DBG, class DT, bool NEWTON = false, bool UNROLL = false, bool ACTOR = false, bool PAIR = false, bool DYN = false>
__global__ __launch_bounds__((PAIR ? 2 : 1) * RR_LANES, (NVS >= 3 ? 1 : 2)) void rr_step_kernel(const RRDims Dk, const RRTables T, const RRIO io_kernarg, const int num_envs,
                                                           const int n_frames) {
  static_assert(!PAIR || (!PROF && !DBG && !NEWTON && !UNROLL && !ACTOR), "PAIR: production physics instance only");
  static_assert(!DYN || (!PROF && !DBG && !NEWTON && !UNROLL && !ACTOR && !PAIR), "DYN: production instance only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int env = blockIdx.x;
  if (env >= num_envs) return;
  // the level schedules address LDS by absolute byte address: the dynamic segment must start at 0, i.e. the kernel has no static
  // LDS -- checked on the HOST for every instance a batch may launch (rr_batch_create: hipFuncGetAttributes().sharedSizeBytes == 0)
  const DT D(Dk);
  const int wrep = PAIR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  Wave<NBS, NVS, NCS, DT, NEWTON, PAIR, DYN> w(D, T, PAIR ? lds + wrep * (D.lds_bytes_rep >> 2) : lds);
  int lane = threadIdx.x & (RR_LANES - 1);
  if (PAIR) { w.rep = wrep; w.s_xc = lds + 2 * (D.lds_bytes_rep >> 2); }
  RRIO io = load_io();
  if (io.env_map) {          // a permutation of 0 .. num_envs-1 (host-checked length); environments are independent, so the mapping
    env = __builtin_amdgcn_readfirstlane(io.env_map[env]);   // only decides which two of them share a SIMD
    if ((unsigned)env >= (unsigned)num_envs) return;
  }
  const int senv = PAIR ? 2 * env + wrep : env;      // row of this wave's replica in the state arrays
  if (DBG) {   // the re-read block must be the real parameter, word for word; on a mismatch say so in the dump and touch nothing else
    const RRIO ref_io = io_kernarg;
    bool same = true;
    for (unsigned i = 0; i < sizeof(RRIO) / sizeof(int); ++i) same &= ((const int*)&io)[i] == ((const int*)&ref_io)[i];
    if (ref_io.dbg && lane == 0) ref_io.dbg[(size_t)env * D.dbg_floats + D.g_kaok] = same ? 1.0f : 0.0f;
    if (!same) return;
  }
  const int mode = io.mode;
  // the debug dump (parity tests) is a separate instance: its paths keep dozens of values alive across the solver
  float* dbg = (DBG && io.dbg) ? io.dbg + (size_t)env * D.dbg_floats : nullptr;     // DBG instance without a dump buffer: contact outputs only

  // wrapper state of a multi-step rollout, wave-uniform
  const int nsteps = UNROLL ? io.unroll_T : 1;
  float u_steps = 0.0f, u_prev_done = 0.0f;
  int u_frame = 0;
  unsigned u_work = 0;
  if (UNROLL) {
    u_steps = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(io.steps_in[env])));
    u_prev_done = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(io.prev_done[env])));
    u_frame = __builtin_amdgcn_readfirstlane(io.cur_frame_in[env]);
  }
  int niter = 0;
  float xq1[4] = {1, 0, 0, 0};   // xquat of body 1 at the last forward pass (obs: xmat[1])
 for (int ut = 0; ut < nsteps; ++ut) {
  if (UNROLL) { lane = opaque(lane); w.lane = lane; asm volatile("" : "+s"(env)); io = load_io(); }
  const size_t ctrl_at = UNROLL ? ((size_t)ut * num_envs + env) * D.nu : (size_t)senv * D.nu;
  if (ACTOR) {
    if (ut == 0) {       // the observation the rollout starts from is row 0 of the env's trajectory
      for (int i = lane; i < D.obs_dim; i += RR_LANES) io.t_obs[rr_traj_obs(io, num_envs, env, 0, 0) * D.obs_dim + i] = io.a_obs_in[(size_t)env * D.obs_dim + i];
    }
    // ORDERING through global memory inside one wave: the observation row the actor reads was written by OTHER lanes of this wave (the
    // previous step's epilogue / the copy above), and the action it writes (lanes < A) is read back as ctrl by all lanes below.  Same-wave
    // vector memory operations complete in order, but the compiler must not move them across each other either: a wavefront-scope fence on
    // both sides states the dependency (guarded by tests/test_gpu_ppo.py::test_one_launch_unroll_with_the_actor_inside, bitwise).
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    rr_actor_step(io, D, lane, env, ut, num_envs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  // ---- load state (a multi-step rollout keeps it in LDS after its first step)
  if (!UNROLL || ut == 0) {
    for (int i = lane; i < D.nq; i += RR_LANES) w.s_qpos[i] = io.qpos_in[(size_t)senv * D.nq + i];
    for (int i = lane; i < D.nv; i += RR_LANES) w.s_qvel[i] = io.qvel_in[(size_t)senv * D.nv + i];
    for (int i = lane; i < D.nu; i += RR_LANES) w.s_act[i] = io.act_in[(size_t)senv * D.nu + i];
  }
  for (int i = lane; i < D.nu; i += RR_LANES) w.s_ctrl[i] = io.ctrl ? io.ctrl[ctrl_at + i] : 0.0f;
#pragma unroll
  for (int s = 0; s < NVS; ++s) {
    const int d = lane + RR_LANES * s;
    if ((!UNROLL || ut == 0) && d < D.nv) w.s_warm[d] = io.warm_in[(size_t)senv * D.nv + d];
    if (d < D.nv) {
      auto di = T.dof_i + RR_DOFI * d;
      w.dofc0[s] = (di[3] & 255) | ((di[2] & 15) << 8) | ((di[9] & 15) << 12) | ((di[0] & 255) << 16) | ((T.body_i[RR_BODYI * di[0]] & 255) << 24);
      w.dofc1[s] = (di[4] & 0xFFFF) | (di[10] << 16);
    } else {
      w.dofc0[s] = 255 | (6 << 8);
      w.dofc1[s] = 0;
    }
    w.qacc[s] = w.Ma[s] = w.grad[s] = w.Mgrad[s] = w.search[s] = w.mv[s] = w.qfrc_con[s] = 0.0f;
  }
  w.work = 0;
  for (int i = lane; i < D.nv; i += RR_LANES) w.s_arm[i] = T.dof_f[16 * i];
  if (NEWTON) for (int i = lane; i < (D.nM + 3) / 4; i += RR_LANES) ((int*)w.s_anc)[i] = T.anc4[i];
  if (lane < 6) w.s_cdof[6 * D.nv + lane] = 0.0f;
  if (lane == 0) w.s_qvel[D.nv] = 0.0f;
  if (lane < 40) w.s_qLD[2 * D.nM + lane] = 0.0f;        // cells ZERO .. pad, and the 16 zero cells behind them (pairs)
  if (lane < 16) { w.s_vec[D.nv + lane] = 0.0f; w.s_x[D.nv + lane] = 0.0f; }   // zero cells the job descriptors pad with / padded steps read
  w.sync();

  if (PROF) { for (int i = 0; i < RR_NPH; ++i) w.pt[i] = 0; w.pt_last = __builtin_readcyclecounter(); }
  const int frames = (mode & 1) ? n_frames : 1;
  for (int f = 0; f < frames; ++f) {
    // RR_FRAME_LOCAL: everything derived from the lane id / env id (per-lane table addresses, output offsets) is loop-invariant,
    // so the optimiser hoists it out of the substep loop and -- with 256 registers taken -- spills it to scratch at the loop head
    // (45 dwords per lane in round 1's build, reloaded one by one inside every substep).  Re-deriving the two ids through an
    // opaque copy per substep keeps those values local to their phase.
    lane = opaque(lane); w.lane = lane;
    asm volatile("" : "+s"(env));
    const bool last = f == frames - 1;
    if (last) io = load_io();
    float* dg = last ? dbg : nullptr;
    float bias[NVS], passive[NVS];
    w.template stamp<PROF>(15);
    // ---- per-substep (re)load of the model constants from the L2-resident tables.  Holding them in registers across
    // the solver made the allocator spill them to scratch (HBM-side write traffic ~80x the algorithmic bytes); a plain
    // reload costs the same read and no write.  `opaque` keeps the loads inside the substep loop.
    {
      const int ol = opaque(lane);
      w.bc0 = load_bodyc(T, ol, D.nbody);
      if (NBS > 1) w.bc1 = load_bodyc(T, ol + RR_LANES, D.nbody);
#pragma unroll
      for (int s = 0; s < NBS; ++s) {
        const int b = ol + RR_LANES * s;
        const bool ok = b >= 1 && b < D.nbody;
        w.banc[s][0] = ok ? T.body_anc[2 * b] : 0;
        w.banc[s][1] = ok ? T.body_anc[2 * b + 1] : 0;
        w.blast[s] = ok ? T.body_i[RR_BODYI * b + 10] : 0;
      }
    }
    if (lane == 0) {  // world body entries (their LDS cells are reused by later phases of every substep)
      for (int k = 0; k < 6; ++k) w.s_cvel[k] = 0.0f;
      for (int k = 0; k < 10; ++k) w.s_cinert[k] = 0.0f;
      for (int k = 0; k < 3; ++k) w.s_xpos[k] = 0.0f;
      w.s_xquat[0] = 1.0f; w.s_xquat[1] = w.s_xquat[2] = w.s_xquat[3] = 0.0f;
    }
    for (int rep = 0; rep < RR_REP_KIN; ++rep) w.kinematics();
    w.kinematics();
    w.template stamp<PROF>(0);
    w.com_pos();
    w.template stamp<PROF>(1);
    if (last) {   // pose outputs of the last forward pass, before the pose cells are recycled
#pragma unroll
      for (int k = 0; k < 4; ++k) xq1[k] = w.s_xquat[4 + k];
      if (io.o_xpos) for (int e = lane; e < 3 * D.nbody; e += RR_LANES) io.o_xpos[(size_t)env * 3 * D.nbody + e] = w.s_xpos[e];
      if (io.o_xmat || dg) {
        for (int b = lane; b < D.nbody; b += RR_LANES) {
          float q[4], mm[9];
          for (int k = 0; k < 4; ++k) q[k] = w.s_xquat[4 * b + k];
          quat_to_mat(mm, q);
          for (int k = 0; k < 9; ++k) {
            if (io.o_xmat) io.o_xmat[(size_t)env * 9 * D.nbody + 9 * b + k] = mm[k];
            if (dg) dg[D.g_xmat + 9 * b + k] = mm[k];
          }
        }
      }
      if (io.o_com && lane == 0) for (int k = 0; k < 3; ++k) io.o_com[(size_t)env * 3 + k] = w.com0[k];
      if (dg) {
        for (int e = lane; e < 3 * D.nbody; e += RR_LANES) dg[D.g_xpos + e] = w.s_xpos[e];
        for (int e = lane; e < 4 * D.nbody; e += RR_LANES) dg[D.g_xquat + e] = w.s_xquat[e];
        for (int e = lane; e < 10 * D.nbody; e += RR_LANES) dg[D.g_cinert + e] = w.s_cinert[e];
        for (int e = lane; e < 6 * D.nv; e += RR_LANES) dg[D.g_cdof + e] = w.s_cdof[e];
        if (lane == 0) { for (int k = 0; k < 3; ++k) { dg[D.g_com + k] = w.com0[k]; dg[D.g_com + 3 + k] = w.com1[k]; } }
      }
    }
    {   // contact geometry outputs (on request only) are served by the debug-dump instance: the production instances carry no code for them
      float *od = nullptr, *op = nullptr, *of = nullptr;
      if (DBG && last) {
        od = io.o_cdist ? io.o_cdist + (size_t)env * D.ncon : nullptr;
        op = io.o_cpos ? io.o_cpos + (size_t)env * 3 * D.ncon : nullptr;
        of = io.o_cframe ? io.o_cframe + (size_t)env * 9 * D.ncon : nullptr;
      }
      if (DYN) w.contact_geometry_dyn();
      else w.contact_geometry(dg, od, op, of);
    }
    w.velocity_sweep();
    w.template stamp<PROF>(2);
    if (last) {   // cinert / cvel of the last forward pass go out now: cinert's cells become the composite inertia next
      if (io.o_cinert) for (int e = lane; e < 10 * D.nbody; e += RR_LANES) io.o_cinert[(size_t)env * 10 * D.nbody + e] = w.s_cinert[e];
      if (io.o_cvel) for (int e = lane; e < 6 * D.nbody; e += RR_LANES) io.o_cvel[(size_t)env * 6 * D.nbody + e] = w.s_cvel[e];
      if (io.obs) {
        float* ob = (ACTOR ? io.t_obs + rr_traj_obs(io, num_envs, env, rr_traj(io, ut).u, rr_traj(io, ut).t + 1) * D.obs_dim : io.obs + (size_t)env * D.obs_dim) + D.nq + D.nv;
        for (int i = lane; i < 10 * (D.nbody - 1); i += RR_LANES) ob[i] = w.s_cinert[10 + i];
        ob += 10 * (D.nbody - 1);
        for (int i = lane; i < 6 * (D.nbody - 1); i += RR_LANES) ob[i] = w.s_cvel[6 + i];
      }
      if (dg) for (int e = lane; e < 6 * D.nbody; e += RR_LANES) dg[D.g_cvel + e] = w.s_cvel[e];
    }
    w.backward_sweep();
    w.template stamp<PROF>(3);
    w.smooth_forces(bias, passive);     // needs cfrc, whose cells the factorisation overwrites
    if (dg) {
      for (int e = lane; e < 10 * D.nbody; e += RR_LANES) dg[D.g_crb + e] = w.s_crb[e];
      for (int e = lane; e < 6 * D.nbody; e += RR_LANES) dg[D.g_cfrc + e] = w.s_cfrc[e];
    }
    w.contact_jobs();     // J*x jobs of the contacts in penetration: needed from here to the end of the substep
    // WAVE PRIORITY.  2048 environments are exactly one resident round, so a launch lasts as long as its slowest environment,
    // and an environment is slow when many contacts carry force (more J'f terms, more line-search rows).  The heavier of the two
    // waves that share a SIMD issues first; the lighter one has slack.  Four graded levels (0 / 2+ / 6+ / 12+ contacts in
    // penetration): -4.6 % launch time, bit-identical results (tools/variant_bench.py; a two-level split gave -3.1 %).
    w.env_prio();
    w.sync();
    for (int rep = 0; rep < RR_REP_MM; ++rep) w.mass_matrix();
    w.mass_matrix();
    w.template stamp<PROF>(4);
    if (dg) for (int e = lane; e < D.nM; e += RR_LANES) dg[D.g_qM + e] = w.s_qLD[2 * e];
    if (NEWTON) {      // M itself is needed all through the Newton iterations (M * search, H = M + ...): keep a copy of the pair array
      for (int e = lane; e < D.nM + 20; e += RR_LANES) *(rr_f2*)(w.s_Mp + 2 * e) = *(const rr_f2*)(w.s_qLD + 2 * e);
      w.sync();
    }
    {   // the substep's only product with M itself: M * qacc_warmstart, for the solver's warm-start context
      float wv[NVS];
#pragma unroll
      for (int s = 0; s < NVS; ++s) { const int d = lane + RR_LANES * s; wv[s] = d < D.nv ? w.s_warm[d] : 0.0f; }
      w.put_vec(wv);
      w.mul_m(w.Ma_warm);
    }
    // ... and by phase: the two level schedules are one long dependent chain of LDS round trips that issues little; at top priority
    // its instructions go out the moment they are ready (-1.2 ... -1.6 % launch time; the same for the solves, the line-search
    // iterations or the tree sweeps measured +0.3 ... +0.6 % each and +3 % together)
#if RR_FACTOR_PRIO
    if ((w.lag_mode & 8) && w.lag_prio == 0) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#endif
    w.factor();
    if (dg) for (int e = lane; e < D.nM; e += RR_LANES) dg[D.g_qLD + e] = w.s_qLD[2 * e];
    w.invert();
    w.env_prio();
    w.template stamp<PROF>(5);
#pragma unroll
    for (int s = 0; s < NVS; ++s) w.qacc_smooth[s] = w.qfrc_smooth[s];
    w.ldl_solve(w.qacc_smooth);
    w.template stamp<PROF>(6);
    if (dg) {
#pragma unroll
      for (int s = 0; s < NVS; ++s) {
        const int d = lane + RR_LANES * s;
        if (d < D.nv) {
          dg[D.g_dinv + d] = w.dinv[s]; dg[D.g_bias + d] = bias[s]; dg[D.g_passive + d] = passive[s];
          dg[D.g_actuator + d] = w.s_qact[d]; dg[D.g_smooth + d] = w.qfrc_smooth[s];
          dg[D.g_qacc_smooth + d] = w.qacc_smooth[s];
        }
      }
    }
    w.constraint_rows(dg);
    w.template stamp<PROF>(7);
    niter = w.template solve<PROF>();
    w.template stamp<PROF>(12);
    if (dg) {
#pragma unroll
      for (int s = 0; s < NVS; ++s) {
        const int d = lane + RR_LANES * s;
        if (d < D.nv) { dg[D.g_qacc + d] = w.qacc[s]; dg[D.g_qfrc_constraint + d] = w.qfrc_con[s]; }
      }
      if (lane == 0) { dg[D.g_misc] = (float)niter; dg[D.g_misc + 1] = w.cost; }
    }
    if (mode & 1) w.euler();
    w.template stamp<PROF>(13);
    if (UNROLL && !last) {       // pacing at substep granularity (RRIO::pace_mode bit 2)
      const RRIO iop = load_io();
      if (iop.progress && (iop.pace_mode & 4)) {
        unsigned seen = 0;
        if (lane == 0) seen = __hip_atomic_fetch_add(iop.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        const float behind = ((float)seen / (float)num_envs - (float)(ut * frames + f + 1)) / (float)frames;      // in env steps
        w.lag_prio = behind > iop.pace_t3 ? 3 : (behind > iop.pace_t2 ? 2 : (behind > iop.pace_t1 ? 1 : 0));
        w.lag_mode = iop.pace_mode;
      }
    }
  }

  w.template stamp<PROF>(14);
  lane = opaque(lane); w.lane = lane;
  asm volatile("" : "+s"(env));
  io = load_io();
  if (PROF && io.prof && lane == 0) for (int i = 0; i < RR_NPH; ++i) io.prof[(size_t)env * RR_NPH + i] = w.pt[i];
  if (UNROLL) u_work += (unsigned)w.work;        // a multi-step launch reports the work of all its steps
  if (io.cost && lane == 0 && wrep == 0) io.cost[env] = (UNROLL ? u_work : (unsigned)w.work) | (DYN && w.dyn_overflow ? 0x80000000u : 0u);
  if (DYN && w.dyn_overflow && io.dyn_overflow && lane == 0) __hip_atomic_fetch_add(io.dyn_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- write back state (a multi-step rollout writes it once, after the wrappers of its last step: see below)
  if (!UNROLL) {
    for (int i = lane; i < D.nq; i += RR_LANES) io.qpos[(size_t)senv * D.nq + i] = w.s_qpos[i];
    for (int i = lane; i < D.nv; i += RR_LANES) io.qvel[(size_t)senv * D.nv + i] = w.s_qvel[i];
    for (int i = lane; i < D.nu; i += RR_LANES) io.act[(size_t)senv * D.nu + i] = w.s_act[i];
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) {
        io.warm[(size_t)senv * D.nv + d] = w.s_warm[d];
        if (io.o_qfrc_actuator) io.o_qfrc_actuator[(size_t)senv * D.nv + d] = w.s_qact[d];
      }
    }
  }

  // ---- reference env epilogue [REF Rodent_Env_Brax.py:103-158]
  if (io.obs) {
    const bool is_reset = (mode & 2) != 0;
    const int old_frame = UNROLL ? u_frame : io.cur_frame_in[env];
    const int new_frame = is_reset ? old_frame : old_frame + 1;
    float* ob = ACTOR ? io.t_obs + rr_traj_obs(io, num_envs, env, rr_traj(io, ut).u, rr_traj(io, ut).t + 1) * D.obs_dim : io.obs + (size_t)env * D.obs_dim;
    int o = 0;
    for (int i = lane; i < D.nq; i += RR_LANES) ob[o + i] = w.s_qpos[i];
    o += D.nq;
    for (int i = lane; i < D.nv; i += RR_LANES) ob[o + i] = w.s_qvel[i];
    o += D.nv;
    o += 16 * (D.nbody - 1);   // cinert[1:], cvel[1:] were written right after the last forward pass
#pragma unroll
    for (int s = 0; s < NVS; ++s) {
      const int d = lane + RR_LANES * s;
      if (d < D.nv) ob[o + d] = w.s_qact[d];
    }
    o += D.nv;
    if (lane < 3) {  // xmat[1] @ (track_pos[frame + 1] - qpos[:3]); JAX clamps the gather index
      int fi = new_frame + 1;
      fi = fi < 0 ? 0 : (fi > io.track_len - 1 ? io.track_len - 1 : fi);
      const v3 v = ld3(io.track_pos + 3 * fi) - ld3(w.s_qpos);
      float m1[9];
      quat_to_mat(m1, xq1);
      // row `lane` of xmat[1] by selects (indexing a register array by the lane id would put it into scratch memory)
      const float r0 = m1[0] * v.x + m1[1] * v.y + m1[2] * v.z, r1 = m1[3] * v.x + m1[4] * v.y + m1[5] * v.z, r2 = m1[6] * v.x + m1[7] * v.y + m1[8] * v.z;
      ob[o + lane] = lane == 0 ? r0 : (lane == 1 ? r1 : r2);
    }
    if (!is_reset) {
      float a2 = 0.0f;
      for (int i = lane; i < D.nu; i += RR_LANES) { const float a = io.ctrl[ctrl_at + i]; a2 += a * a; }
      a2 = wave_sum(a2);
      if (lane == 0) {
        int fi = old_frame < 0 ? 0 : (old_frame > io.track_len - 1 ? io.track_len - 1 : old_frame);
        const v3 dx = ld3(w.s_qpos) - ld3(io.track_pos + 3 * fi);
        // explicit roundings (no fused multiply-add left to the optimiser), so that the instances of the kernel form the reward from the
        // same operations: single-step and multi-step instances agree bit for bit (tests/test_gpu_env.py); the actor-inside instance to
        // ONE ulp -- `expf` below is expanded inline per instance and its expansion there rounds differently
        // (tests/test_gpu_ppo.py::test_one_launch_unroll_with_the_actor_inside allows exactly that)
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx.x, dx.x), __fmul_rn(dx.y, dx.y)), __fmul_rn(dx.z, dx.z));
        const float pos_reward = expf(__fmul_rn(-100.0f, sqrtf(d2)));
        const float z = w.s_qpos[2];
        float healthy = z < io.z_min ? 0.0f : 1.0f;
        if (z > io.z_max) healthy = 0.0f;
        const float hr = io.terminate_when_unhealthy ? io.healthy_reward : __fmul_rn(io.healthy_reward, healthy);
        const float cc = __fmul_rn(io.ctrl_cost_weight, a2);     // explicit roundings: every instance of the kernel forms the reward identically
        const float rew = __fsub_rn(__fadd_rn(pos_reward, hr), cc);   // (left to the optimiser, one instance fused the product into the sum: 1 ulp)
        io.reward[env] = rew;
        if (ACTOR) io.t_reward[rr_traj_at(io, num_envs, env, ut)] = rew;
        io.done[env] = io.terminate_when_unhealthy ? 1.0f - healthy : 0.0f;
        io.metrics[3 * env] = pos_reward; io.metrics[3 * env + 1] = -cc; io.metrics[3 * env + 2] = hr;
        io.cur_frame[env] = new_frame;
      }
    }
    if (UNROLL) {
      // EpisodeWrapper + AutoResetWrapper on the step just made (action_repeat 1), as rr_wrap_kernel applies them:
      // steps' = (prev_done ? 0 : steps) + 1; over = steps' >= episode_length; done <- over ? 1 : done; truncation = over ? 1 - done_env : 0;
      // where done, the stored first state and first observation come back (info -- cur_frame, steps -- is not restored)
      const float z = w.s_qpos[2];
      const float healthy = (z < io.z_min || z > io.z_max) ? 0.0f : 1.0f;
      const float done_env = io.terminate_when_unhealthy ? 1.0f - healthy : 0.0f;
      u_steps = (u_prev_done != 0.0f ? 0.0f : u_steps) + 1.0f;
      const bool over = u_steps >= io.episode_length;
      const float done2 = over ? 1.0f : done_env, trunc = over ? 1.0f - done_env : 0.0f;
      u_prev_done = done2;
      u_frame = new_frame;
      if (ACTOR && lane == 0) { const size_t at = rr_traj_at(io, num_envs, env, ut); io.t_discount[at] = 1.0f - done2; io.t_trunc[at] = trunc; }
      if (__builtin_amdgcn_readfirstlane(__float_as_int(done2)) != 0) {
        w.sync();
        for (int i = lane; i < D.nq; i += RR_LANES) w.s_qpos[i] = io.first_qpos[(size_t)env * D.nq + i];
        for (int i = lane; i < D.nv; i += RR_LANES) { w.s_qvel[i] = io.first_qvel[(size_t)env * D.nv + i]; w.s_warm[i] = io.first_warm[(size_t)env * D.nv + i]; }
        for (int i = lane; i < D.nu; i += RR_LANES) w.s_act[i] = io.first_act[(size_t)env * D.nu + i];
        for (int i = lane; i < D.obs_dim; i += RR_LANES) ob[i] = io.first_obs[(size_t)env * D.obs_dim + i];
        w.sync();
      }
      if (ACTOR) {      // the last observation of a segment is also the first of the next one (the learner's trajectories overlap by one row)
        const RRTraj tr = rr_traj(io, ut);
        if (tr.t + 1 == io.a_seg && ut + 1 < nsteps) {
          float* nx = io.t_obs + rr_traj_obs(io, num_envs, env, tr.u + 1, 0) * D.obs_dim;
          for (int i = lane; i < D.obs_dim; i += RR_LANES) nx[i] = ob[i];
        }
      }
      if (io.progress && ut + 1 < nsteps) {
        unsigned seen = 0;
        if (lane == 0) seen = __hip_atomic_fetch_add(io.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        const float per_step = (io.pace_mode & 4) ? (float)frames : 1.0f;          // the counter's units per env step
        const float behind = ((float)seen / (float)num_envs) / per_step - (float)(ut + 1);      // env steps behind the average environment
        w.lag_prio = behind > io.pace_t3 ? 3 : (behind > io.pace_t2 ? 2 : (behind > io.pace_t1 ? 1 : 0));
        w.lag_mode = io.pace_mode;
      }
      if (ut == nsteps - 1) {
        if (lane == 0) { io.done[env] = done2; io.steps_out[env] = u_steps; io.trunc_out[env] = trunc; }
        for (int i = lane; i < D.nq; i += RR_LANES) io.qpos[(size_t)env * D.nq + i] = w.s_qpos[i];
        for (int i = lane; i < D.nv; i += RR_LANES) { io.qvel[(size_t)env * D.nv + i] = w.s_qvel[i]; io.warm[(size_t)env * D.nv + i] = w.s_warm[i]; }
        for (int i = lane; i < D.nu; i += RR_LANES) io.act[(size_t)env * D.nu + i] = w.s_act[i];
      }
    }
  }
 }     // ut
}
