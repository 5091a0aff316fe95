// rr_mlp.h -- fused actor/critic MLP forward on the f32 matrix cores of gfx950 (MI355X).
//
// Path replaced: `acting.actor_step` -> `ppo.networks.make_inference_fn` (normalise obs, policy MLP) and the forward half of
// `ppo.losses.compute_ppo_loss` (policy + value MLP on the same observations) [UP brax.training; SURVEY.md a22 / a25 /
// Appendix E; driven by REF brax_rodent_run_ppo.py:97-114].  Shapes of `make_ppo_networks` defaults: policy
// obs -> 32 x4 -> 2*action_size, value obs -> 256 x5 -> 1, swish (SiLU) on hidden layers.
//
// One workgroup (4 wavefronts) owns 32 observation rows through ALL layers of BOTH networks:
//   * the observation chunk [32 x 16] is read from HBM once, normalised ((x - mean) / std) while it is staged into LDS,
//     and feeds the first layer of both nets in the same k-loop (policy 1263 -> 32, value 1263 -> 256);
//   * hidden activations never leave LDS ([32 x 256] value, [32 x 32] policy); weights stream through L2
//     (2.5 MB per workgroup, shared by all workgroups);
//   * arithmetic: v_mfma_f32_32x32x2_f32 for the 256-wide value layers (each wave owns two 32x32 output tiles: 4 waves x 64
//     columns), v_mfma_f32_16x16x4_f32 for the 32-wide policy layers (each wave one 16x16 tile of the 32x32 output) -- f32 in,
//     f32 accumulate, bit-for-bit a k-ordered fmaf chain (the reference's precision; no bf16 down-cast anywhere);
//   * global loads of chunk c+1 are issued into registers before the MFMAs of chunk c (register double buffer);
//   * LDS row strides 18 / 258 / 34 floats make every fragment read (lane -> row, lane>>5|4 -> k) bank-conflict free.
// Optional outputs: the hidden PRE-activations of every layer (for a hand-written backward pass), row-major [M][width].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define RR_MLP_BM 32        // observation rows per workgroup
#define RR_MLP_KC 16        // k-chunk staged through LDS
#define RR_MLP_VH 256       // value hidden width
#define RR_MLP_PH 32        // policy hidden width
#define RR_MLP_MAXL 8

struct RRMlpNet {
  const float* W[RR_MLP_MAXL];   // [out][in] row-major (torch.nn.Linear.weight)
  const float* b[RR_MLP_MAXL];   // [out]
  int nlayers;                   // hidden layers + 1
  int out_dim;                   // width of the last layer (policy: 2*action_size <= 64; value: 1)
};
struct RRMlpArgs {
  const float* obs; int M, K;
  const int64_t* rows;                       // nullable: sample m reads row rows[m] of `obs` (a minibatch addressed in place)
  const float* mean; const float* std_;      // nullable: no normalisation
  RRMlpNet pol, val;                         // nlayers == 0: that network is skipped
  float* pol_out;                            // [M][pol.out_dim]
  float* val_out;                            // [M]
  float* pol_act;                            // nullable: [pol.nlayers-1][M][32]   hidden PRE-activations z = h W' + b (for a backward pass)
  float* val_act;                            // nullable: [val.nlayers-1][M][256]
  unsigned long long* prof;                  // nullable, diagnostic (RR_MLP_PROF): [workgroups][16] shader-clock stamps per phase
};

typedef float rr_f4 __attribute__((ext_vector_type(4)));
typedef float rr_f16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rr_silu(float v) { return v / (1.0f + expf(-v)); }
// sigmoid for the VALUE network's 256-wide epilogues (forward and backward: 32 values per thread and layer, where libm's expf + the IEEE
// quotient were ~25 instructions per value = the bulk of the "store" phases): v_exp_f32 on x log2(e) and a reciprocal refined by one Newton
// step, each <= 1 ulp.  The policy network keeps rr_silu: its forward must agree bit for bit with the rollout's actor (rr_kernel.h, rr_ppo.h).
#ifndef RR_MLP_FAST_SIGMOID
#define RR_MLP_FAST_SIGMOID 1
#endif
__device__ __forceinline__ float rr_sigmoid_val(float v) {
#if RR_MLP_FAST_SIGMOID
  const float e = __builtin_amdgcn_exp2f(fminf(-1.4426950408889634f * v, 126.0f));      // exp(-v); clamped: 1 + 2^126 stays finite
  const float d = 1.0f + e, r = __builtin_amdgcn_rcpf(d);
  return r * (2.0f - d * r);
#else
  return 1.0f / (1.0f + expf(-v));
#endif
}

constexpr int RR_SX = RR_MLP_KC + 2;        // stage stride (18): 18 n mod 64 is a bijection of n = 0..31 onto the even banks
constexpr int RR_SV = RR_MLP_VH + 2;        // value activation stride (258)
constexpr int RR_SP = RR_MLP_PH + 2;        // policy activation stride (34)
// LDS of the forward kernel, 51.5 KB so that THREE workgroups share a CU (the 704 workgroups of the launcher's minibatch are then one
// resident round; at two per CU the second round was 37 % full).  Two regions, reused by phase:
//   A [32][258]  layer 1: observation chunk [32][18] + weight chunk [288][18] (value rows 0..255, policy 256..287);
//                afterwards the value activations (written by layer 1's epilogue, when every wave is done with the stage)
//   B [256][18]  policy layers 2..head (run right after layer 1): their whole weight matrix [2 x 64][18] at the bottom, the policy
//                activations [32][34] above it; then, the policy being finished, the weight chunks of the value hidden layers
constexpr int RR_MLP_LDS_A = RR_MLP_BM * RR_SV, RR_MLP_LDS_B = RR_MLP_VH * RR_SX;
constexpr int RR_MLP_ACTP_AT = 3072;        // actP inside B (above the 2304 floats of the policy weights)
constexpr int RR_MLP_LDS_FLOATS = RR_MLP_LDS_A + RR_MLP_LDS_B;
static_assert(RR_MLP_BM * RR_SX + (RR_MLP_VH + RR_MLP_PH) * RR_SX <= RR_MLP_LDS_A, "layer-1 stage fits region A");
static_assert(2 * 64 * RR_SX <= RR_MLP_ACTP_AT && RR_MLP_ACTP_AT + RR_MLP_BM * RR_SP <= RR_MLP_LDS_B, "policy weights and activations fit region B");

// Stage rows [row0, row0+nrows) x k [k0, k0+KC) of a row-major matrix (leading dimension ld, valid k < K, valid rows < R) in
// two steps: `fetch` issues the global loads into registers, `commit` writes them to LDS (so loads fly during the MFMAs).
// The tile moves in 16-byte pieces (global_load_dwordx4; rows may start at any 4-byte boundary, K = 1263): element-wise staging
// cost as many issue cycles per chunk as the chunk's matrix-core work.  The LDS rows are 72 bytes apart: two 8-byte stores.
typedef float rr_f2v __attribute__((ext_vector_type(2)));
typedef float rr_f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NROWS>
struct RRStage {
  static constexpr int NV = NROWS * RR_MLP_KC / 4;      // float4 pieces of the tile; piece v = row v / 4, floats 4 (v % 4) ..
  static constexpr int PER = (NV + 255) / 256;
  rr_f4 r[PER];
  // No branches around the loads (a conditional load cannot be speculated: the compiler turns it into divergent control flow
  // with a full s_waitcnt inside, which serialises the prefetch): rows are CLAMPED into the matrix -- the outputs of the
  // duplicated rows are never stored -- and only the last, partial k-chunk (FULL = false) masks, element-wise, by select.
  template <bool FULL>
  static __device__ __forceinline__ rr_f4 load4(const float* row, int k, int K) {
    if (FULL) { const rr_f4u u = *(const rr_f4u*)(row + k); return rr_f4{u[0], u[1], u[2], u[3]}; }
    rr_f4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float x = row[min(k + j, K - 1)]; t[j] = k + j < K ? x : 0.0f; }
    return t;
  }
  template <bool FULL>
  __device__ __forceinline__ void fetch(const float* src, int ld, int row0, int R, int k0, int K) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int v = min((int)threadIdx.x + 256 * i, NV - 1), n = min(row0 + (v >> 2), R - 1);
      r[i] = load4<FULL>(src + (size_t)n * ld, k0 + 4 * (v & 3), K);
    }
  }
  // same, piece i of this thread reading from the row that starts at src + off[i]; the offsets are the same for every chunk
  // (a thread keeps its tile rows), so the caller looks the row indices up once
  template <bool FULL>
  __device__ __forceinline__ void fetch_at(const float* src, const long long (&off)[PER], int k0, int K) {
#pragma unroll
    for (int i = 0; i < PER; ++i) r[i] = load4<FULL>(src + off[i], k0 + 4 * ((threadIdx.x + 256 * i) & 3), K);
  }
  __device__ __forceinline__ void commit(float* dst /* [NROWS][RR_SX] */) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int v = threadIdx.x + 256 * i;
      if (v < NV) {
        float* d = dst + (v >> 2) * RR_SX + 4 * (v & 3);
        *(rr_f2v*)d = rr_f2v{r[i][0], r[i][1]};
        *(rr_f2v*)(d + 2) = rr_f2v{r[i][2], r[i][3]};
      }
    }
  }
};

// k-loop over one staged chunk.  A rows come from `xa` (stride sa, k offset ka); weights from the stage `w`.
// Value: wave `wv` owns output columns [64 wv, 64 wv + 64) as two 32x32 tiles.  Policy: wave owns the 16x16 tile (mt, nt).
template <bool VAL, bool POL>
__device__ __forceinline__ void rr_mlp_chunk(const float* xa, int sa, int ka, const float* w, int prow0, rr_f16& a0, rr_f16& a1, rr_f4& ap,
                                             int lane, int wv) {
  if (VAL) {
    const float* xr = xa + (lane & 31) * sa + ka + (lane >> 5);
    const float* w0 = w + (64 * wv + (lane & 31)) * RR_SX + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < RR_MLP_KC; kk += 2) {
      const float a = xr[kk];
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w0[kk], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w0[32 * RR_SX + kk], a1, 0, 0, 0);
    }
  }
  if (POL) {
    const int mt = wv >> 1, nt = wv & 1;
    const float* xr = xa + (16 * mt + (lane & 15)) * sa + ka + (lane >> 4);
    const float* w0 = w + (prow0 + 16 * nt + (lane & 15)) * RR_SX + (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < RR_MLP_KC; kk += 4) ap = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[kk], w0[kk], ap, 0, 0, 0);
  }
}

// one 256 -> 256 product of the tile in `actV` with W ([256 out][256 in] row-major), the weight chunks through two register stages
// (chunk c+2 is in flight while chunk c is multiplied).  Ends with a barrier: every wave has read actV when it returns.
__device__ __forceinline__ void rr_mlp_hidden_layer(const float* W, const float* actV, float* sW, rr_f16& a0, rr_f16& a1, rr_f4& ap, int lane, int wv) {
  constexpr int nchunk = RR_MLP_VH / RR_MLP_KC;
  RRStage<RR_MLP_VH> g0, g1;
  g0.fetch<true>(W, RR_MLP_VH, 0, RR_MLP_VH, 0, RR_MLP_VH);
  g1.fetch<true>(W, RR_MLP_VH, 0, RR_MLP_VH, RR_MLP_KC, RR_MLP_VH);
#pragma unroll 1
  for (int c = 0; c < nchunk; c += 2) {
    g0.commit(sW);
    __syncthreads();
    if (c + 2 < nchunk) g0.fetch<true>(W, RR_MLP_VH, 0, RR_MLP_VH, (c + 2) * RR_MLP_KC, RR_MLP_VH);
    rr_mlp_chunk<true, false>(actV, RR_SV, c * RR_MLP_KC, sW, 0, a0, a1, ap, lane, wv);
    __syncthreads();
    g1.commit(sW);
    __syncthreads();
    if (c + 3 < nchunk) g1.fetch<true>(W, RR_MLP_VH, 0, RR_MLP_VH, (c + 3) * RR_MLP_KC, RR_MLP_VH);
    rr_mlp_chunk<true, false>(actV, RR_SV, (c + 1) * RR_MLP_KC, sW, 0, a0, a1, ap, lane, wv);
    __syncthreads();
  }
}

// epilogue of a 256-wide value layer: act[m][n] = silu(acc + b[n]); C/D map of the 32x32 tile: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__device__ __forceinline__ void rr_mlp_store_val(float* actV, const rr_f16& a0, const rr_f16& a1, const float* bias, int lane, int wv,
                                                 float* dump, int row0, int M) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = 64 * wv + 32 * t + (lane & 31);
    const float bn = bias[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float z = (t ? a1[r] : a0[r]) + bn;
      actV[m * RR_SV + n] = z * rr_sigmoid_val(z);
      if (dump && row0 + m < M) dump[(size_t)(row0 + m) * RR_MLP_VH + n] = z;
    }
  }
}
// 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + reg
__device__ __forceinline__ void rr_mlp_store_pol(float* actP, const rr_f4& ap, const float* bias, int lane, int wv, float* dump, int row0, int M) {
  const int mt = wv >> 1, nt = wv & 1;
  const int n = 16 * nt + (lane & 15);
  const float bn = bias[n];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 16 * mt + 4 * (lane >> 4) + r;
    const float z = ap[r] + bn;
    actP[m * RR_SP + n] = rr_silu(z);
    if (dump && row0 + m < M) dump[(size_t)(row0 + m) * RR_MLP_PH + n] = z;
  }
}

#ifndef RR_MLP_FWD_WGS
#define RR_MLP_FWD_WGS 3
#endif
#ifndef RR_MLP_RCP
#define RR_MLP_RCP 0         // 1: multiply by a refined reciprocal instead of dividing -- measured 0.443 vs 0.446-0.450 ms, within noise, so the IEEE division (what the reference computes) stays
#endif
#ifndef RR_MLP_NO_NORM
#define RR_MLP_NO_NORM 0      // 1 (experiment): the normaliser compiled out of the forward (pre-normalised observations)
#endif
#ifndef RR_MLP_STAGES
#define RR_MLP_STAGES 1      // register stages of layer 1's chunks (2: chunk c+2 in flight during chunk c; needs more than 168 VGPRs)
#endif
// HV / HP: which networks the launch carries, as template constants.  As run-time flags they left a three-way branch inside layer 1's chunk
// loop; the compiler kept the 36 accumulator registers in different places on its arms and joined them with 32 v_mov per chunk behind an
// s_nop 15 (the MFMA-result hazard), i.e. every chunk waited for its matrix work to drain: layer 1 ran 6300 ticks per chunk against 2200
// for the identically shaped chunks of the hidden layers (RR_MLP_PROF stamps, tools/pmc_forward_rows.py).
template <bool HV, bool HP>
__global__ __launch_bounds__(256, RR_MLP_FWD_WGS) void rr_mlp_forward_kernel(const RRMlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* actV = lds;                                 // region A: [32][258] value activations ...
  float* sX = lds;                                   //   ... during layer 1: [32][18] normalised observation chunk
  float* sW = sX + RR_MLP_BM * RR_SX;                //   ... and [288][18] weight chunk: rows 0..255 value, 256..287 policy
  float* sB = lds + RR_MLP_LDS_A;                    // region B: [256][18] weight chunks of the hidden layers / policy weights
  float* actP = sB + RR_MLP_ACTP_AT;                 //   ... [32][34] policy activations
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row0 = blockIdx.x * RR_MLP_BM;
  constexpr bool has_val = HV, has_pol = HP;
  const int M = A.M, K = A.K;
  auto stamp = [&](int i) { if (A.prof && threadIdx.x == 0) A.prof[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);

  // ------------------------------------------------------------------ layer 1 of both nets: one pass over the observation
  {
    rr_f16 a0 = {0}, a1 = {0};
    rr_f4 ap = {0, 0, 0, 0};
    // TWO register stages: the loads of chunk c+2 are issued while chunk c is multiplied (one chunk is ~1150 matrix-core cycles
    // per wave, an L2 round trip under load is longer: with a single stage every chunk waited for its loads)
    struct Stage { RRStage<RR_MLP_BM> gx; RRStage<RR_MLP_VH> gv; RRStage<RR_MLP_PH> gp; rr_f4 mu[RRStage<RR_MLP_BM>::PER], sd[RRStage<RR_MLP_BM>::PER]; };
#if RR_MLP_STAGES == 2
    Stage S0, S1;
#else
    Stage S0;
#endif
    const int nchunk = (K + RR_MLP_KC - 1) / RR_MLP_KC;
    long long xoff[RRStage<RR_MLP_BM>::PER];       // start of this thread's observation rows (minibatch addressed in place)
#pragma unroll
    for (int i = 0; i < RRStage<RR_MLP_BM>::PER; ++i) {
      const int v = min((int)threadIdx.x + 256 * i, RRStage<RR_MLP_BM>::NV - 1), m = min(row0 + (v >> 2), M - 1);   // clamped: see RRStage
      xoff[i] = (long long)(A.rows ? A.rows[m] : m) * K;
    }
    auto fetch_t = [&](Stage& S, int c, auto full) {
      constexpr bool FULL = decltype(full)::value;
      const int k0 = c * RR_MLP_KC;
      S.gx.template fetch_at<FULL>(A.obs, xoff, k0, K);
      if (!RR_MLP_NO_NORM && A.mean) {      // the normaliser's chunk rides along; it is APPLIED at commit time (arithmetic on the loaded values here
#pragma unroll       // would wait for them -- and for every older load -- inside the fetch, i.e. no load would ever fly during the MFMAs)
        for (int i = 0; i < RRStage<RR_MLP_BM>::PER; ++i) {
          const int k = k0 + 4 * ((threadIdx.x + 256 * i) & 3);
          S.mu[i] = RRStage<RR_MLP_BM>::template load4<FULL>(A.mean, k, K);
          S.sd[i] = RRStage<RR_MLP_BM>::template load4<FULL>(A.std_, k, K);
          if (!FULL) {     // columns past K: x = mean = 0 there, keep the quotient finite (it meets zero weights)
#pragma unroll
            for (int j = 0; j < 4; ++j) S.sd[i][j] = k + j < K ? S.sd[i][j] : 1.0f;
          }
        }
      }
      if (has_val) S.gv.template fetch<FULL>(A.val.W[0], K, 0, RR_MLP_VH, k0, K);
      if (has_pol) S.gp.template fetch<FULL>(A.pol.W[0], K, 0, RR_MLP_PH, k0, K);
    };
    auto fetch = [&](Stage& S, int c) {            // uniform branch: only the last chunk can be partial
      if ((c + 1) * RR_MLP_KC <= K) fetch_t(S, c, std::true_type{});
      else fetch_t(S, c, std::false_type{});
    };
    auto step = [&](Stage& S, int c) {
      if (!RR_MLP_NO_NORM && A.mean) {      // normalise in registers: (x - mean) / std  (running_statistics.normalize); columns past K meet zero weights
#pragma unroll
        for (int i = 0; i < RRStage<RR_MLP_BM>::PER; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#if RR_MLP_RCP
            // (x - mean) * (1 / std), the reciprocal by v_rcp_f32 + one Newton step (<= 1 ulp from the quotient): the IEEE division was 11 of
            // the ~22 vector instructions per staged element-quad of this loop, on the critical path between two barriers
            const float r0 = __builtin_amdgcn_rcpf(S.sd[i][j]);
            S.gx.r[i][j] = (S.gx.r[i][j] - S.mu[i][j]) * (r0 * (2.0f - S.sd[i][j] * r0));
#else
            S.gx.r[i][j] = (S.gx.r[i][j] - S.mu[i][j]) / S.sd[i][j];
#endif
          }
      }
      S.gx.commit(sX);
      if (has_val) S.gv.commit(sW);
      if (has_pol) S.gp.commit(sW + RR_MLP_VH * RR_SX);
      __syncthreads();
      if (c + RR_MLP_STAGES < nchunk) fetch(S, c + RR_MLP_STAGES);
      rr_mlp_chunk<HV, HP>(sX, RR_SX, 0, sW, RR_MLP_VH, a0, a1, ap, lane, wv);
      __syncthreads();
    };
#if RR_MLP_STAGES == 2
    fetch(S0, 0);
    if (nchunk > 1) fetch(S1, 1);
    for (int c = 0; c < nchunk; c += 2) {
      step(S0, c);
      if (c + 1 < nchunk) step(S1, c + 1);
    }
#else
    fetch(S0, 0);
    for (int c = 0; c < nchunk; ++c) step(S0, c);
#endif
    stamp(1);
    if (has_val) rr_mlp_store_val(actV, a0, a1, A.val.b[0], lane, wv, A.val_act, row0, M);
    if (has_pol) rr_mlp_store_pol(actP, ap, A.pol.b[0], lane, wv, A.pol_act, row0, M);
    __syncthreads();
    stamp(2);
  }

  // ------------------------------------------------------------------ policy hidden layers 32 -> 32 and the head 32 -> out_dim (region B)
  for (int l = 1; has_pol && l < A.pol.nlayers; ++l) {
    const bool head = l == A.pol.nlayers - 1;
    const int nout = head ? A.pol.out_dim : RR_MLP_PH;
    __syncthreads();
    // the whole weight matrix [nout <= 64][32] as two k-chunks side by side: chunk c of row n at sW[(64 c + n) * 18 ..]
    for (int e = threadIdx.x; e < 64 * RR_MLP_PH; e += 256) {
      const int n = e / RR_MLP_PH, k = e % RR_MLP_PH;
      sB[(64 * (k / RR_MLP_KC) + n) * RR_SX + (k % RR_MLP_KC)] = n < nout ? A.pol.W[l][n * RR_MLP_PH + k] : 0.0f;
    }
    __syncthreads();
    if (!head) {
      rr_f16 d0 = {0}, d1 = {0};
      rr_f4 ap = {0, 0, 0, 0};
      rr_mlp_chunk<false, true>(actP, RR_SP, 0, sB, 0, d0, d1, ap, lane, wv);
      rr_mlp_chunk<false, true>(actP, RR_SP, RR_MLP_KC, sB, 64, d0, d1, ap, lane, wv);
      __syncthreads();
      rr_mlp_store_pol(actP, ap, A.pol.b[l], lane, wv, A.pol_act ? A.pol_act + (size_t)l * M * RR_MLP_PH : nullptr, row0, M);
    } else {
      // [32 x 64] output = 2 x 4 tiles of 16x16: wave wv takes n-tile wv for both m-tiles
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        rr_f4 ap = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float* xr = actP + (16 * mt + (lane & 15)) * RR_SP + c * RR_MLP_KC + (lane >> 4);
          const float* w0 = sB + (64 * c + 16 * wv + (lane & 15)) * RR_SX + (lane >> 4);
#pragma unroll
          for (int kk = 0; kk < RR_MLP_KC; kk += 4) ap = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[kk], w0[kk], ap, 0, 0, 0);
        }
        const int n = 16 * wv + (lane & 15);
        if (n < nout) {
          const float bn = A.pol.b[l][n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * mt + 4 * (lane >> 4) + r;
            if (row0 + m < M) A.pol_out[(size_t)(row0 + m) * nout + n] = ap[r] + bn;
          }
        }
      }
    }
  }
  __syncthreads();        // the policy is done with region B
  stamp(3);
  // ------------------------------------------------------------------ value hidden layers 256 -> 256 (activations stay in LDS; weight chunks through region B)
  for (int l = 1; has_val && l < A.val.nlayers - 1; ++l) {
    rr_f16 a0 = {0}, a1 = {0};
    rr_f4 ap = {0, 0, 0, 0};
    rr_mlp_hidden_layer(A.val.W[l], actV, sB, a0, a1, ap, lane, wv);
    stamp(2 + 2 * l);
    // every wave has read the whole input before anyone overwrites it (the barrier closing the last chunk)
    rr_mlp_store_val(actV, a0, a1, A.val.b[l], lane, wv, A.val_act ? A.val_act + (size_t)l * M * RR_MLP_VH : nullptr, row0, M);
    __syncthreads();
    stamp(3 + 2 * l);
  }
  // value head 256 -> 1: eight lanes per row
  if (has_val) {
    const int l = A.val.nlayers - 1;
    const int m = threadIdx.x >> 3, part = threadIdx.x & 7;
    const float* w = A.val.W[l];
    float s = 0.0f;
#pragma unroll 8
    for (int k = part; k < RR_MLP_VH; k += 8) s = fmaf(actV[m * RR_SV + k], w[k], s);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (part == 0 && row0 + m < M) A.val_out[row0 + m] = s + A.val.b[l][0];
  }
  stamp(12);
}

// ------------------------------------------------------------------------------------------ value network, backward: the delta chain
// Backward pass of the value MLP's hidden stack on the matrix cores, the mirror image of the forward kernel: one workgroup
// carries 32 rows from the head back to the first hidden layer,
//   delta_{nh-1} = g w_head * silu'(z_{nh-1}),     delta_{j-1} = (delta_j W_j) * silu'(z_{j-1}),  j = nh-1 .. 1,
// with the delta tile resident in LDS between layers (the products delta_j W_j never go to HBM) and the epilogue doing what
// were separate passes: silu' from the forward's pre-activation dump, h_j = silu(z_j) written over the dump (operand of
// dW_{j+1} = delta_{j+1}' h_j, left to the caller's matrix products together with dW_0 = delta_0' x), delta_j written out, and
// the column sums of delta_j (= db_j) as per-workgroup partial sums reduced in fixed order by rr_mlp_colsum_kernel.
// The B operand of delta_j W_j is W_j read "down the columns"; the caller passes W_j TRANSPOSED ([in][out] row-major), so the
// weight staging and the k-loop are the forward's (rr_mlp_chunk).
struct RRMlpBwdArgs {
  const float* g;                    // [M]  d loss / d value
  const float* w_head;               // [256]
  const float* Wt[RR_MLP_MAXL];      // Wt[j], j = 1 .. nh-1: W_j transposed, [256 in][256 out]
  float* z;                          // [nh][M][256]  pre-activations in, silu(z) out
  float* delta;                      // [nh][M][256]  out
  float* part;                       // [nh][gridDim.x][256]
  float* bgrad[RR_MLP_MAXL];         // db_j [256], j = 0 .. nh-1 (written by rr_mlp_colsum_kernel)
  int M, nh, nblk;
};

// acc (or, at the head, g w_head) -> delta: multiply by silu'(z), store delta / silu(z), keep the tile in LDS, column sums
template <bool HEAD>
__device__ __forceinline__ void rr_mlp_bwd_epilogue(const RRMlpBwdArgs& A, int j, float* actV, const rr_f16& a0, const rr_f16& a1,
                                                    const float* gm /* [16] rows of this lane, HEAD only */, int lane, int wv, int row0) {
  float* zj = A.z + (size_t)j * A.M * RR_MLP_VH;
  float* dj = A.delta + (size_t)j * A.M * RR_MLP_VH;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = 64 * wv + 32 * t + (lane & 31);
    const float wn = HEAD ? A.w_head[n] : 0.0f;
    float cs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float d = 0.0f;
      if (row0 + m < A.M) {
        const size_t i = (size_t)(row0 + m) * RR_MLP_VH + n;
        const float zz = zj[i], s = rr_sigmoid_val(zz);
        d = (HEAD ? gm[r] * wn : (t ? a1[r] : a0[r])) * (s * (1.0f + zz * (1.0f - s)));
        dj[i] = d;
        zj[i] = zz * s;
      }
      actV[m * RR_SV + n] = d;
      cs += d;
    }
    cs += __shfl_xor(cs, 32);
    if (lane < 32) A.part[((size_t)j * A.nblk + blockIdx.x) * RR_MLP_VH + n] = cs;
  }
}

// three workgroups per CU (51.5 KB of LDS each, <= 168 VGPRs): the 704 workgroups of the launcher's minibatch are resident at once;
// at two per CU (512 slots) they ran as two rounds, the second 37 % full: 0.277 -> 0.215 ms
#ifndef RR_MLP_BWD_WGS
#define RR_MLP_BWD_WGS 3
#endif
__global__ __launch_bounds__(256, RR_MLP_BWD_WGS) void rr_mlp_value_backward_kernel(const RRMlpBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sW = lds;                                   // [256][18]  chunk of W_j transposed
  float* actV = sW + RR_MLP_VH * RR_SX;              // [32][258]  delta tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row0 = blockIdx.x * RR_MLP_BM;
  {
    float gm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      gm[r] = row0 + m < A.M ? A.g[row0 + m] : 0.0f;
    }
    rr_f16 z0 = {0}, z1 = {0};
    rr_mlp_bwd_epilogue<true>(A, A.nh - 1, actV, z0, z1, gm, lane, wv, row0);
    __syncthreads();
  }
  for (int j = A.nh - 1; j >= 1; --j) {
    rr_f16 a0 = {0}, a1 = {0};
    rr_f4 ap = {0, 0, 0, 0};
    rr_mlp_hidden_layer(A.Wt[j], actV, sW, a0, a1, ap, lane, wv);
    rr_mlp_bwd_epilogue<false>(A, j - 1, actV, a0, a1, nullptr, lane, wv, row0);
    __syncthreads();
  }
}
constexpr int RR_MLP_BWD_LDS_FLOATS = RR_MLP_VH * RR_SX + RR_MLP_BM * RR_SV;

// db_j[n] = sum over workgroups of part[j][b][n]; grid (16 column groups, nh layers)
__global__ __launch_bounds__(256) void rr_mlp_colsum_kernel(const RRMlpBwdArgs A) {
  __shared__ float sh[256];
  const int j = blockIdx.y, c = threadIdx.x & 15, rg = threadIdx.x >> 4, n = blockIdx.x * 16 + c;
  const float* part = A.part + (size_t)j * A.nblk * RR_MLP_VH;
  float t = 0.0f;
  for (int b = rg; b < A.nblk; b += 16) t += part[(size_t)b * RR_MLP_VH + n];
  sh[threadIdx.x] = t;
  __syncthreads();
  if (rg == 0) {
    float u = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) u += sh[r * 16 + c];
    A.bgrad[j][n] = u;
  }
}

// ------------------------------------------------------------------------------------------ weight gradients: dW = delta' h, split over the rows
// C[o][i] = sum_m a[m][o] b[row(m)][i] with a = delta [M][O], b = h [M][I] or the raw observations (then b is normalised while
// it is staged, so dW_0 needs no correction term, and `rows` addresses the minibatch inside the unroll buffer, so no gathered
// copy of the observations is read here).  The output is small (<= 256 x 1263) and the reduction long (M ~ 2e4): a library
// product fills 1 .. 40 workgroups of the 1024 the chip wants.  Here the row range is cut into `nslice` slices; workgroup
// (tile, slice) accumulates its [TO x TI] tile over its rows on v_mfma_f32_32x32x2_f32 (both operands are k-major in memory,
// so a fragment read is 32 consecutive floats of one staged row: lanes 0..31 row k, lanes 32..63 row k+1, strides = 32 mod 64
// floats keep the two halves on disjoint banks) and writes a partial tile; rr_mlp_dw_reduce_kernel adds the slices in order.
struct RRDwArgs {
  const float* a; const float* b;
  const int64_t* rows;                       // nullable: b's row of sample m is rows[m]
  const float* mean; const float* std_;      // nullable: the product is taken with (b - mean[i]) / std[i]: applied to the SUM,
  const float* bsum;                         //   out = (sum_m a b - bsum[o] mean[i]) / std[i], bsum[o] = sum_m a[m][o]
  int M, O, I, rows_per_slice, nslice;
  float* part;                               // [nslice][O][I]
  float* out;                                // [O][I]
};

// several products in one launch: item blockIdx.z (workgroups past an item's own grid leave at once)
#define RR_DW_MAXB 12
struct RRDwBatch { RRDwArgs it[RR_DW_MAXB]; int n; };

template <int GO, int GI, int WO, int WI, int KC>
__global__ __launch_bounds__(256, 2) void rr_mlp_dw_kernel(const RRDwBatch B) {
  static_assert(GO * GI == 4, "four wavefronts");
  const RRDwArgs& A = B.it[blockIdx.z];
  {
    constexpr int TO_ = GO * WO * 32, TI_ = GI * WI * 32;
    if ((int)blockIdx.y >= A.nslice || (int)blockIdx.x >= ((A.O + TO_ - 1) / TO_) * ((A.I + TI_ - 1) / TI_)) return;
  }
  constexpr int TO = GO * WO * 32, TI = GI * WI * 32;
  constexpr int SA = TO + (TO % 64 == 0 ? 32 : 0), SB = TI + (TI % 64 == 0 ? 32 : 0);
  constexpr int PA = KC * TO / 256, PB = KC * TI / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sA = lds;                  // [KC][SA]
  float* sB = lds + KC * SA;        // [KC][SB]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_i = (A.I + TI - 1) / TI;
  const int o0 = (blockIdx.x / tiles_i) * TO, i0 = (blockIdx.x % tiles_i) * TI;
  const int m0 = blockIdx.y * A.rows_per_slice, m1 = min(A.M, m0 + A.rows_per_slice);
  const int woff = (wv / GI) * WO * 32, ioff = (wv % GI) * WI * 32;
  rr_f16 acc[WO][WI];
#pragma unroll
  for (int x = 0; x < WO; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = rr_f16{0};
  // staging in 16-byte pieces: a thread moves PA/4 + PB/4 float4 per chunk (global_load_dwordx4 -> ds_write_b128); four times fewer
  // memory instructions and address computations than element-wise staging, which cost as many issue cycles as the matrix-core
  // work of a chunk.  Rows of b may start at any 4-byte boundary (I = 1263): the vector type carries 4-byte alignment.
  // No branches around the loads (see RRStage): rows past the slice are CLAMPED to its last row and the a-piece is zeroed by
  // select (0 * finite = 0); workgroups whose tile overhangs the matrix (last tile column / row, or O not a multiple of 4) take
  // the element-wise path -- a workgroup-uniform choice.
  constexpr int VA = PA / 4, VB = PB / 4;
  static_assert(PA % 4 == 0 && PB % 4 == 0, "whole float4 per thread");
  // Three workgroup-uniform cases.  `whole`: the tile lies inside the matrix and a's rows are 16-byte aligned: aligned vector loads,
  // ds_write_b128.  Overhanging tile (last tile column / row): a piece that would cross the edge is loaded from the last four columns
  // instead (start clamped to width - 4) and written to the LDS columns it really holds -- it overlaps its neighbour with identical
  // values; LDS columns past the edge keep stale data and only feed outputs that are never stored.  Width < 4: element-wise.
  const bool whole = (A.O & 3) == 0 && o0 + TO <= A.O && i0 + TI <= A.I;
  const bool vec_a = A.O - o0 >= 4, vec_b = A.I - i0 >= 4;        // at least one whole piece inside the matrix
  int acol[VA], bcol[VB];           // first tile column of this thread's pieces (loop invariant)
#pragma unroll
  for (int q = 0; q < VA; ++q) { const int c = 4 * ((threadIdx.x + 256 * q) % (TO / 4)); acol[q] = vec_a ? min(o0 + c, A.O - 4) - o0 : c; }
#pragma unroll
  for (int q = 0; q < VB; ++q) { const int c = 4 * ((threadIdx.x + 256 * q) % (TI / 4)); bcol[q] = vec_b ? min(i0 + c, A.I - 4) - i0 : c; }
  rr_f4 ra[VA], rb[VB];
  long long boff[VB];               // start of b's row of this thread's pieces in the NEXT chunk to fetch
  auto lookup = [&](int mc) {       // one chunk ahead of the loads that use it, so the index load is never on their critical path
#pragma unroll
    for (int q = 0; q < VB; ++q) {
      const int m = min(mc + (int)(threadIdx.x + 256 * q) / (TI / 4), m1 - 1);
      boff[q] = (long long)(A.rows ? A.rows[m] : m) * A.I;
    }
  };
  auto load4 = [&](const float* p, bool vec, int col, int width) {     // p = row start + tile start
    rr_f4 t;
    if (vec) { const rr_f4u tu = *(const rr_f4u*)(p + col); t = rr_f4{tu[0], tu[1], tu[2], tu[3]}; }
    else {
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = p[min(col + u, width - 1)];
    }
    return t;
  };
  auto fetch = [&](int mc) {        // no arithmetic on the loaded values here: it would wait for the loads
#pragma unroll
    for (int q = 0; q < VA; ++q) {
      const int m = min(mc + (int)(threadIdx.x + 256 * q) / (TO / 4), m1 - 1);
      const float* p = A.a + (size_t)m * A.O + o0;
      if (whole) ra[q] = *(const rr_f4*)(p + acol[q]);
      else ra[q] = load4(p, vec_a, acol[q], A.O - o0);
    }
#pragma unroll
    for (int q = 0; q < VB; ++q) rb[q] = load4(A.b + boff[q] + i0, whole || vec_b, bcol[q], A.I - i0);
  };
  lookup(m0);
  fetch(m0);
  lookup(m0 + KC);
  for (int mc = m0; mc < m1; mc += KC) {
#pragma unroll
    for (int q = 0; q < VA; ++q) {
      const int kk = (threadIdx.x + 256 * q) / (TO / 4);
      const rr_f4 t = mc + kk < m1 ? ra[q] : rr_f4{0.0f, 0.0f, 0.0f, 0.0f};       // rows past the slice contribute nothing
      float* d = sA + kk * SA + acol[q];
      if (whole) *(rr_f4*)d = t;
      else { d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3]; }
    }
#pragma unroll
    for (int q = 0; q < VB; ++q) {
      float* d = sB + ((threadIdx.x + 256 * q) / (TI / 4)) * SB + bcol[q];
      if (whole) *(rr_f4*)d = rb[q];
      else { d[0] = rb[q][0]; d[1] = rb[q][1]; d[2] = rb[q][2]; d[3] = rb[q][3]; }
    }
    __syncthreads();
    if (mc + KC < m1) { fetch(mc + KC); lookup(mc + 2 * KC); }
    const float* pa = sA + (lane >> 5) * SA + woff + (lane & 31);
    const float* pb = sB + (lane >> 5) * SB + ioff + (lane & 31);
#pragma unroll 8
    for (int k2 = 0; k2 < KC; k2 += 2) {
      float af[WO], bf[WI];
#pragma unroll
      for (int x = 0; x < WO; ++x) af[x] = pa[k2 * SA + 32 * x];
#pragma unroll
      for (int y = 0; y < WI; ++y) bf[y] = pb[k2 * SB + 32 * y];
#pragma unroll
      for (int x = 0; x < WO; ++x)
#pragma unroll
        for (int y = 0; y < WI; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[x], bf[y], acc[x][y], 0, 0, 0);
    }
    __syncthreads();
  }
  float* dst = A.part + (size_t)blockIdx.y * A.O * A.I;
#pragma unroll
  for (int x = 0; x < WO; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) {
      const int i = i0 + ioff + 32 * y + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + woff + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (o < A.O && i < A.I) dst[(size_t)o * A.I + i] = acc[x][y][r];
      }
    }
}
// out[e] = sum_s part[s][e] (then the normaliser, see RRDwArgs), slices added in order s = 0, 1, .. (fixed order: graph replay == eager,
// bit for bit).  A thread owns FOUR consecutive outputs and walks the slices with 16-byte loads, eight in flight: a wave reads 1 KB of
// one slice per instruction.  (Round 2's form -- a block per 16 outputs, 16 thread groups each adding every 16th slice in 64-byte
// pieces -- took 101 us per minibatch for ~170 MB of partial tiles; the planner now also cuts fewer slices, see dw_plan_batch.)
__global__ __launch_bounds__(256) void rr_mlp_dw_reduce_kernel(const RRDwBatch B) {
  const RRDwArgs& A = B.it[blockIdx.y];
  const size_t n = (size_t)A.O * A.I;
  const size_t e0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= n) return;
  float u[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int ns = A.nslice;
  if ((n & 3) == 0) {                       // every slice starts 16-byte aligned
    const float* p = A.part + e0;
    int s = 0;
    for (; s + 8 <= ns; s += 8) {
      rr_f4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const rr_f4*)(p + (size_t)(s + j) * n);
#pragma unroll
      for (int j = 0; j < 8; ++j) { u[0] += v[j][0]; u[1] += v[j][1]; u[2] += v[j][2]; u[3] += v[j][3]; }
    }
    for (; s < ns; ++s) { const rr_f4 v = *(const rr_f4*)(p + (size_t)s * n); u[0] += v[0]; u[1] += v[1]; u[2] += v[2]; u[3] += v[3]; }
  } else {
    for (int s = 0; s < ns; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) if (e0 + j < n) u[j] += A.part[(size_t)s * n + e0 + j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t e = e0 + j;
    if (e < n) {
      float r = u[j];
      if (A.mean) {
        const int o = (int)(e / A.I), i = (int)(e % A.I);
        r = (r - A.bsum[o] * A.mean[i]) / A.std_[i];
      }
      A.out[e] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------ the rollout's actor: policy net + head, two launches
// The actor step between two env steps is [N x 1263] -> 32 x4 -> 2A at N = 2048: 0.17 GFLOP and 10 MB -- microseconds of work -- but as a
// row-tiled kernel it is 64 workgroups walking 79 k-chunks one after the other (~70 us).  Here the first layer is split over k as well:
// rr_policy_l1_kernel, grid (N / 32, KS), each workgroup 32 rows x one k-slice on v_mfma_f32_16x16x4_f32, partial sums [KS][N][32];
// rr_policy_tail_kernel (csrc/rr_ppo.h side: 32 lanes per row, weights in LDS) adds the slices in order, applies the remaining layers
// and the tanh-normal head (sample, squash, log-prob) or its mode.
struct RRPolL1Args {
  const float* obs; const int64_t* rows; const float* mean; const float* std_; const float* W;    // W [32][K]
  int M, K, chunks_per_slice;
  float* part;                     // [gridDim.y][M][32]
};
__global__ __launch_bounds__(256) void rr_policy_l1_kernel(const RRPolL1Args A) {
  __shared__ __attribute__((aligned(16))) float sX[RR_MLP_BM * RR_SX], sW[RR_MLP_PH * RR_SX];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row0 = blockIdx.x * RR_MLP_BM, M = A.M, K = A.K;
  const int nchunk = (K + RR_MLP_KC - 1) / RR_MLP_KC;
  const int c0 = blockIdx.y * A.chunks_per_slice, c1 = min(nchunk, c0 + A.chunks_per_slice);
  typedef RRStage<RR_MLP_BM> St;     // 32 rows x 16: 128 pieces, threads 0..127
  St gx, gw;
  rr_f4 mu[St::PER], sd[St::PER];
  long long xoff[St::PER];
#pragma unroll
  for (int i = 0; i < St::PER; ++i) {
    const int v = min((int)threadIdx.x + 256 * i, St::NV - 1), m = min(row0 + (v >> 2), M - 1);
    xoff[i] = (long long)(A.rows ? A.rows[m] : m) * K;
  }
  auto fetch_t = [&](int c, auto full) {
    constexpr bool FULL = decltype(full)::value;
    const int k0 = c * RR_MLP_KC;
    gx.template fetch_at<FULL>(A.obs, xoff, k0, K);
    gw.template fetch<FULL>(A.W, K, 0, RR_MLP_PH, k0, K);
    if (A.mean) {
#pragma unroll
      for (int i = 0; i < St::PER; ++i) {
        const int k = k0 + 4 * ((threadIdx.x + 256 * i) & 3);
        mu[i] = St::template load4<FULL>(A.mean, k, K);
        sd[i] = St::template load4<FULL>(A.std_, k, K);
        if (!FULL) {
#pragma unroll
          for (int j = 0; j < 4; ++j) sd[i][j] = k + j < K ? sd[i][j] : 1.0f;
        }
      }
    }
  };
  auto fetch = [&](int c) {
    if ((c + 1) * RR_MLP_KC <= K) fetch_t(c, std::true_type{});
    else fetch_t(c, std::false_type{});
  };
  rr_f16 d0 = {0}, d1 = {0};
  rr_f4 ap = {0, 0, 0, 0};
  if (c0 < c1) fetch(c0);
  for (int c = c0; c < c1; ++c) {
    if (A.mean) {
#pragma unroll
      for (int i = 0; i < St::PER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) gx.r[i][j] = (gx.r[i][j] - mu[i][j]) / sd[i][j];
    }
    gx.commit(sX);
    gw.commit(sW);
    __syncthreads();
    if (c + 1 < c1) fetch(c + 1);
    rr_mlp_chunk<false, true>(sX, RR_SX, 0, sW, 0, d0, d1, ap, lane, wv);
    __syncthreads();
  }
  const int mt = wv >> 1, nt = wv & 1, n = 16 * nt + (lane & 15);
  float* dst = A.part + (size_t)blockIdx.y * M * RR_MLP_PH;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = row0 + 16 * mt + 4 * (lane >> 4) + r;
    if (m < M) dst[(size_t)m * RR_MLP_PH + n] = ap[r];
  }
}
