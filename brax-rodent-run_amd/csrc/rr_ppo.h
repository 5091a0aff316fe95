// The loss half of one PPO minibatch update and its gradient with respect to the network outputs, three launches.
//
// What `brax.training.agents.ppo.losses.compute_ppo_loss` [UP; REF brax_rodent_run_ppo.py:97-114]
// computes after the two network forwards -- truncation-aware GAE, device-local advantage normalisation, tanh-normal
// log-probabilities, the clipped surrogate, the value loss, the sampled entropy -- and what reverse-mode differentiation of it
// hands back to the networks (d total_loss / d policy_logits, d total_loss / d values).  Composed from tensor ops this is ~45
// launches per minibatch (forward) plus as many in the backward; here:
//   K1 rr_ppo_gae_kernel    one thread per trajectory: reverse scan, vs and advantages, block sums of adv and adv^2 (double)
//   K2 rr_ppo_loss_kernel   32 lanes per sample (one action dimension per lane, coalesced rows): log-prob, entropy, the three
//                           loss terms (block sums, double) and the gradient rows
//   K3 rr_ppo_metrics_kernel  the four scalars of the metrics dict
// No atomics: every reduction is a fixed tree over per-block partial sums, so a replayed HIP graph reproduces the eager result
// bit for bit.  The minibatch is addressed THROUGH the permutation (`idx`), so the five gathered copies of the batch leaves
// (raw_action, log_prob, reward, discount, truncation) are never made.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RRPpoArgs {
  // network outputs on the gathered minibatch, time-major: logits [(T+1)*B][2A] (row t*B + b; the last B rows are the bootstrap
  // observation's and carry no gradient), values [(T+1)*B]
  const float* logits; const float* values;
  // batch leaves, batch-major as the unroll buffer holds them: raw_action [R][T][A], log_prob/reward/discount/truncation [R][T];
  // idx [B] selects the minibatch rows (NULL = rows 0..B-1)
  const float* raw_action; const float* log_prob; const float* reward; const float* discount; const float* truncation;
  const int64_t* idx;
  const float* noise;            // [T*B][A] standard normal draws of the entropy estimate (time-major, like the logits)
  int T, B, A;
  float entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon, min_std;
  int normalize_advantage;
  float* grad_logits;            // [(T+1)*B][2A]
  float* grad_values;            // [(T+1)*B]
  float* metrics;                // total_loss, policy_loss, v_loss, entropy_loss
  float* vs; float* adv;         // workspace [T*B] each
  double* part_adv;              // [2 * nblk1]
  double* part_loss;             // [3 * nblk2]
  int nblk1, nblk2;
};

template <int N>
static __device__ __forceinline__ void rr_block_sum(double (&v)[N], double* sh /* [N * 4] */, double* out, int stride) {
  // 256 threads = 4 waves: wave sums by shuffles, then the four wave sums through LDS
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < N; ++k) sh[k * 4 + w] = v[k];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < N; ++k) out[k * stride] = (sh[k * 4] + sh[k * 4 + 1]) + (sh[k * 4 + 2] + sh[k * 4 + 3]);
}

__global__ __launch_bounds__(256) void rr_ppo_gae_kernel(const RRPpoArgs P) {
  __shared__ double sh[8];
  const int b = blockIdx.x * 256 + threadIdx.x;
  const int T = P.T, B = P.B;
  double s[2] = {0.0, 0.0};
  if (b < B) {
    const size_t row = (size_t)(P.idx ? P.idx[b] : b) * T;
    const float vb = P.values[(size_t)T * B + b];
    float acc = 0.0f;
    for (int t = T - 1; t >= 0; --t) {     // vs_t - v_t = delta_t + gamma (1-term_t)(1-trunc_t) lambda (vs_{t+1} - v_{t+1})
      const size_t i = (size_t)t * B + b;
      const float tr = P.truncation[row + t], mask = 1.0f - tr, nt = 1.0f - (1.0f - P.discount[row + t]) * mask;
      const float r = P.reward[row + t] * P.reward_scaling, v = P.values[i];
      const float vnext = t == T - 1 ? vb : P.values[i + B];
      const float delta = (r + P.discounting * nt * vnext - v) * mask;
      acc = delta + P.discounting * nt * mask * P.gae_lambda * acc;
      P.vs[i] = acc + v;
    }
    for (int t = 0; t < T; ++t) {          // advantages use vs_{t+1} (bootstrap at the end)
      const size_t i = (size_t)t * B + b;
      const float tr = P.truncation[row + t], mask = 1.0f - tr, nt = 1.0f - (1.0f - P.discount[row + t]) * mask;
      const float vsn = t == T - 1 ? vb : P.vs[i + B];
      const float a = (P.reward[row + t] * P.reward_scaling + P.discounting * nt * vsn - P.values[i]) * mask;
      P.adv[i] = a;
      s[0] += (double)a; s[1] += (double)a * (double)a;
    }
  }
  rr_block_sum<2>(s, sh, P.part_adv + blockIdx.x, P.nblk1);
}

static __device__ __forceinline__ float rr_softplus(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void rr_ppo_loss_kernel(const RRPpoArgs P) {
  __shared__ double sh[12];
  __shared__ float s_stat[2];
  const int T = P.T, B = P.B, A = P.A;
  const int n = T * B;
  if (threadIdx.x < 64) {                  // advantage mean / std (population) from K1's block sums
    double s0 = 0.0, s1 = 0.0;
    for (int k = threadIdx.x; k < P.nblk1; k += 64) { s0 += P.part_adv[k]; s1 += P.part_adv[P.nblk1 + k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    if (threadIdx.x == 0) {
      const double mean = s0 / n, var = fmax(s1 / n - mean * mean, 0.0);
      s_stat[0] = P.normalize_advantage ? (float)mean : 0.0f;
      s_stat[1] = P.normalize_advantage ? 1.0f / ((float)sqrt(var) + 1e-8f) : 1.0f;
    }
  }
  __syncthreads();
  const float amean = s_stat[0], ainv = s_stat[1];
  const int lane = threadIdx.x & 31;
  const int P2 = 2 * A;
  const float invn = 1.0f / (float)n;
  const float HALF_LOG_2PI = 0.91893853320467274178f, LOG2 = 0.69314718055994530942f;
  double acc[3] = {0.0, 0.0, 0.0};
  // bootstrap rows: no gradient
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * P2; i += gridDim.x * 256) P.grad_logits[(size_t)n * P2 + i] = 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) P.grad_values[n + i] = 0.0f;
  const int per_block = 256 / 32;
  for (int smp = blockIdx.x * per_block + (threadIdx.x >> 5); smp < n; smp += gridDim.x * per_block) {
    const int t = smp / B, b = smp - t * B;
    const size_t row = (size_t)(P.idx ? P.idx[b] : b) * T + t;
    const float* lg = P.logits + (size_t)smp * P2;
    float lp = 0.0f, ent = 0.0f;
    // pass 1: log-prob of the behaviour action under the current policy, entropy estimate (sums over the action dimensions)
    for (int a = lane; a < A; a += 32) {
      const float loc = lg[a], sr = lg[A + a], scale = rr_softplus(sr) + P.min_std;
      const float raw = P.raw_action[row * A + a], z = (raw - loc) / scale, ls = logf(scale);
      lp += -0.5f * z * z - ls - HALF_LOG_2PI - 2.0f * (LOG2 - raw - rr_softplus(-2.0f * raw));
      const float x = loc + scale * P.noise[(size_t)smp * A + a];
      ent += 0.5f + HALF_LOG_2PI + ls + 2.0f * (LOG2 - x - rr_softplus(-2.0f * x));
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { lp += __shfl_xor(lp, o, 32); ent += __shfl_xor(ent, o, 32); }
    const float adv = (P.adv[smp] - amean) * ainv;
    const float rho = expf(lp - P.log_prob[row]);
    const float lo = 1.0f - P.clipping_epsilon, hi = 1.0f + P.clipping_epsilon;
    const float s1 = rho * adv, s2 = fminf(fmaxf(rho, lo), hi) * adv;
    const float inr = (rho >= lo && rho <= hi) ? 1.0f : 0.0f;
    // d min(s1, s2) / d rho: ties split evenly (torch.minimum / jnp.minimum), the clamp passes the gradient inside its range
    const float w = s1 < s2 ? 1.0f : (s1 > s2 ? inr : 0.5f + 0.5f * inr);
    const float g_lp = -invn * adv * w * rho;              // d policy_loss / d log_prob
    const float g_h = -P.entropy_cost * invn;              // d entropy_loss / d entropy
    const float v = P.values[smp], ve = P.vs[smp] - v;
    if (lane == 0) {
      acc[0] += (double)fminf(s1, s2); acc[1] += (double)ve * (double)ve; acc[2] += (double)ent;
      P.grad_values[smp] = -0.5f * invn * ve;              // d (0.25 mean(ve^2)) / d v
    }
    // pass 2: gradient rows
    for (int a = lane; a < A; a += 32) {
      const float loc = lg[a], sr = lg[A + a], scale = rr_softplus(sr) + P.min_std, is = 1.0f / scale;
      const float raw = P.raw_action[row * A + a], z = (raw - loc) * is;
      const float eps = P.noise[(size_t)smp * A + a], th = tanhf(loc + scale * eps);
      const float dloc = g_lp * z * is + g_h * (-2.0f * th);
      const float dscale = g_lp * (z * z - 1.0f) * is + g_h * (is - 2.0f * th * eps);
      const float sig = 1.0f / (1.0f + expf(-sr));         // softplus'
      P.grad_logits[(size_t)smp * P2 + a] = dloc;
      P.grad_logits[(size_t)smp * P2 + A + a] = dscale * sig;
    }
  }
  rr_block_sum<3>(acc, sh, P.part_loss + blockIdx.x, P.nblk2);
}

__global__ __launch_bounds__(64) void rr_ppo_metrics_kernel(const RRPpoArgs P) {
  double s[3] = {0.0, 0.0, 0.0};
  for (int k = threadIdx.x; k < P.nblk2; k += 64)
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] += P.part_loss[j * P.nblk2 + k];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o, 64);
  if (threadIdx.x == 0) {
    const double n = (double)P.T * P.B;
    const float pl = (float)(-s[0] / n), vl = (float)(0.25 * s[1] / n), el = (float)(-(double)P.entropy_cost * s[2] / n);
    P.metrics[0] = pl + vl + el; P.metrics[1] = pl; P.metrics[2] = vl; P.metrics[3] = el;
  }
}

// ------------------------------------------------------------------------------------------ backward of a hidden SiLU layer, elementwise part
// Given G = delta_l W_l (the matrix product, [M][H]) and the layer's pre-activations z [M][H]:
//   delta_{l-1} = G * silu'(z),  silu'(z) = s (1 + z (1 - s)),  s = sigmoid(z)      (may overwrite G)
//   h_{l-1}     = silu(z) = z s   (the operand of dW_l = delta_l' h_{l-1}; may overwrite z)
//   db_{l-1}[n] = sum_m delta_{l-1}[m][n]   (per-block partial sums, then rr_colsum_kernel: fixed order, no atomics)
// One launch instead of the seven (silu, the five pieces of the composite silu_backward, the bias sum) it replaces.
// H divides 256; a block owns `rows_per_block` consecutive rows; thread -> column tid % H, row group tid / H.
__global__ __launch_bounds__(256) void rr_silu_bwd_kernel(const float* __restrict__ G, const float* z, int M, int H, int rows_per_block,
                                                          float* delta, float* h, float* __restrict__ part /* [gridDim.x][H] */) {
  __shared__ float sh[256];
  const int n = threadIdx.x % H, rg = threadIdx.x / H, nrg = 256 / H;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc = 0.0f;
  for (int m = m0 + rg; m < m1; m += nrg) {
    const size_t i = (size_t)m * H + n;
    const float zz = z[i], s = 1.0f / (1.0f + expf(-zz));
    const float d = G[i] * (s * (1.0f + zz * (1.0f - s)));
    delta[i] = d;
    h[i] = zz * s;
    acc += d;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (rg == 0) {
    float t = acc;
    for (int r = 1; r < nrg; ++r) t += sh[r * H + n];
    part[(size_t)blockIdx.x * H + n] = t;
  }
}
// out[n] = sum_b part[b][n]: a block owns 16 columns, 16 row groups of threads each sum every 16th row (64-byte coalesced
// reads), then the 16 group sums go through LDS in a fixed order
__global__ __launch_bounds__(256) void rr_colsum_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ out) {
  __shared__ float sh[256];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4, n = blockIdx.x * 16 + c;
  float t = 0.0f;
  if (n < H)
    for (int b = rg; b < nblk; b += 16) t += part[(size_t)b * H + n];
  sh[threadIdx.x] = t;
  __syncthreads();
  if (rg == 0 && n < H) {
    float u = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) u += sh[r * 16 + c];
    out[n] = u;
  }
}

// ------------------------------------------------------------------------------------------ the actor's head: sample, squash, log-prob
// `NormalTanhDistribution` on the rollout path [UP brax.training.distribution; acting.actor_step]: from the policy logits
// (loc | pre-softplus scale) and one standard-normal draw per action dimension,
//   raw = loc + (softplus(s) + min_std) * eps,  action = tanh(raw),
//   log_prob = sum_i [ -0.5 eps_i^2 - log scale_i - 0.5 log 2 pi - 2 (log 2 - raw_i - softplus(-2 raw_i)) ]
// in ONE launch (32 lanes per env, one action dimension per lane) instead of the ~20 elementwise / reduction launches of the
// composed tensor expression, which cost ~0.07 ms between two 1.5 ms env steps.  (z = (raw - loc) / scale is eps up to rounding;
// the composed expression recomputes it, so the two log-probs differ by float32 rounding, as tested.)
__global__ __launch_bounds__(256) void rr_policy_sample_kernel(const float* __restrict__ logits, const float* __restrict__ noise, int N, int A,
                                                               float min_std, float* __restrict__ action, float* __restrict__ raw_out,
                                                               float* __restrict__ logp) {
  const int lane = threadIdx.x & 31;
  const int n = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (n >= N) return;
  const float HALF_LOG_2PI = 0.91893853320467274178f, LOG2 = 0.69314718055994530942f;
  const float* lg = logits + (size_t)n * 2 * A;
  float lp = 0.0f;
  for (int a = lane; a < A; a += 32) {
    const float loc = lg[a], scale = rr_softplus(lg[A + a]) + min_std;
    const float raw = loc + scale * noise[(size_t)n * A + a];
    const float z = (raw - loc) / scale;
    lp += -0.5f * z * z - logf(scale) - HALF_LOG_2PI - 2.0f * (LOG2 - raw - rr_softplus(-2.0f * raw));
    raw_out[(size_t)n * A + a] = raw;
    action[(size_t)n * A + a] = tanhf(raw);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 32);
  if (lane == 0) logp[n] = lp;
}

// ------------------------------------------------------------------------------------------ policy network, backward: the delta chain
// The 32-wide policy stack is 0.1 GFLOP per minibatch -- nothing for the matrix cores, but as library calls its backward was four
// small products, four elementwise passes and four column sums per update (~110 us of launches).  Here one launch: 32 lanes per
// row (lane = hidden unit), all weights in LDS, the row's delta handed around the 32 lanes by shuffles:
//   delta_{nh-1} = (g W_head) * silu'(z_{nh-1}),   delta_{j-1} = (delta_j W_j) * silu'(z_{j-1}),
// writing delta_j (operand of dW_{j+1} = delta_{j+1}' h_j), h_j = silu(z_j) over the forward's dump, and per-block column sums of
// delta_j (= db_j; fixed-order reduction by rr_policy_colsum_kernel).
#define RR_POL_MAXL 8
struct RRPolBwdArgs {
  const float* g;                 // [M][P]  d loss / d logits
  const float* w_head;            // [P][32]
  const float* W[RR_POL_MAXL];    // W[j], j = 1 .. nh-1: [32 out][32 in] (torch layout)
  float* z;                       // [nh][zrows][32]  pre-activations in, silu(z) out (rows 0 .. M-1 of each layer)
  float* delta;                   // [nh][M][32]  out
  float* part;                    // [nh][gridDim.x][32]
  float* bgrad[RR_POL_MAXL];      // db_j [32]
  int M, P, nh, nblk, zrows;      // zrows: rows per layer of z (>= M: the minibatch may carry bootstrap rows behind the M used ones)
};
__global__ __launch_bounds__(256) void rr_policy_backward_kernel(const RRPolBwdArgs A) {
  extern __shared__ float sw[];                 // [P][32] head, then (nh - 1) x [32][32]
  __shared__ float sh[RR_POL_MAXL][8][32];
  const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
  for (int e = threadIdx.x; e < A.P * 32; e += 256) sw[e] = A.w_head[e];
  for (int j = 1; j < A.nh; ++j)
    for (int e = threadIdx.x; e < 1024; e += 256) sw[A.P * 32 + (j - 1) * 1024 + e] = A.W[j][e];
  __syncthreads();
  float cs[RR_POL_MAXL];
#pragma unroll
  for (int j = 0; j < RR_POL_MAXL; ++j) cs[j] = 0.0f;
  for (int row = blockIdx.x * 8 + rg; row < A.M; row += gridDim.x * 8) {
    const float* gr = A.g + (size_t)row * A.P;
    const float g0 = lane < A.P ? gr[lane] : 0.0f, g1 = lane + 32 < A.P ? gr[lane + 32] : 0.0f;
    float acc = 0.0f;
    for (int o = 0; o < A.P; ++o) acc = fmaf(o < 32 ? __shfl(g0, o, 32) : __shfl(g1, o - 32, 32), sw[o * 32 + lane], acc);
#pragma unroll
    for (int jj = 0; jj < RR_POL_MAXL; ++jj) {
      const int j = A.nh - 1 - jj;             // nh-1 .. 0
      if (j < 0) break;
      const size_t i = ((size_t)j * A.M + row) * 32 + lane, iz = ((size_t)j * A.zrows + row) * 32 + lane;
      const float zz = A.z[iz], s = 1.0f / (1.0f + expf(-zz));
      const float d = acc * (s * (1.0f + zz * (1.0f - s)));
      A.delta[i] = d;
      A.z[iz] = zz * s;
      cs[jj] += d;
      if (j > 0) {
        const float* w = sw + A.P * 32 + (j - 1) * 1024;
        acc = 0.0f;
#pragma unroll
        for (int o = 0; o < 32; ++o) acc = fmaf(__shfl(d, o, 32), w[o * 32 + lane], acc);
      }
    }
  }
#pragma unroll
  for (int jj = 0; jj < RR_POL_MAXL; ++jj) sh[jj][rg][lane] = cs[jj];
  __syncthreads();
  if (rg == 0) {
    for (int jj = 0; jj < A.nh; ++jj) {
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += sh[jj][r][lane];
      A.part[((size_t)(A.nh - 1 - jj) * A.nblk + blockIdx.x) * 32 + lane] = t;
    }
  }
}
// db_j[n] = sum over blocks of part[j][b][n]: block j, 32 groups of 32 lanes each add every 32nd partial, fixed order
__global__ __launch_bounds__(1024) void rr_policy_colsum_kernel(const RRPolBwdArgs A) {
  __shared__ float sh[32][32];
  const int j = blockIdx.x, lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const float* part = A.part + (size_t)j * A.nblk * 32;
  float t = 0.0f;
  for (int b = rg; b < A.nblk; b += 32) t += part[(size_t)b * 32 + lane];
  sh[rg][lane] = t;
  __syncthreads();
  if (rg == 0) {
    float u = 0.0f;
#pragma unroll
    for (int r = 0; r < 32; ++r) u += sh[r][lane];
    A.bgrad[j][lane] = u;
  }
}

// ------------------------------------------------------------------------------------------ the rollout's actor, second launch
// Rows of the policy's first-layer partial sums (rr_policy_l1_kernel, csrc/rr_mlp.h) -> action.  32 lanes per row (lane = hidden unit),
// weights TRANSPOSED in LDS (lane n reads w[k][n]: consecutive banks), the activation vector handed around by shuffles:
//   h = silu(sum_slices + b_0); h = silu(W_l h + b_l), l = 1 .. nh-1; logits = W_head h + b_head;
//   noise given: raw = loc + (softplus(s) + min_std) eps, action = tanh(raw), log_prob as rr_policy_sample_kernel; noise NULL: tanh(loc).
struct RRPolTailArgs {
  const float* part; int nslice;
  const float* W[RR_POL_MAXL];    // W[l], l = 1 .. nh-1: [32][32];  W[nh]: head [P][32]
  const float* b[RR_POL_MAXL + 1];// b[0 .. nh-1]: [32]; b[nh]: [P]
  int M, P, A, nh;
  const float* noise; float min_std;
  float* action; float* raw; float* logp; float* logits;     // raw / logp / logits nullable
};
__global__ __launch_bounds__(256) void rr_policy_tail_kernel(const RRPolTailArgs T) {
  extern __shared__ float sw[];                      // (nh - 1) x [32 k][32 n], then head [32 k][64 n], then biases nh x 32 + 64
  const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
  float* swh = sw + (T.nh - 1) * 1024;
  float* sb = swh + 2048;
  for (int l = 1; l < T.nh; ++l)
    for (int e = threadIdx.x; e < 1024; e += 256) sw[(l - 1) * 1024 + (e & 31) * 32 + (e >> 5)] = T.W[l][e];      // [n][k] -> [k][n]
  for (int e = threadIdx.x; e < 2048; e += 256) { const int n = e >> 5, k = e & 31; swh[k * 64 + n] = n < T.P ? T.W[T.nh][n * 32 + k] : 0.0f; }
  for (int e = threadIdx.x; e < T.nh * 32; e += 256) sb[e] = T.b[e >> 5][e & 31];
  for (int e = threadIdx.x; e < 64; e += 256) sb[T.nh * 32 + e] = e < T.P ? T.b[T.nh][e] : 0.0f;
  __syncthreads();
  const float HALF_LOG_2PI = 0.91893853320467274178f, LOG2 = 0.69314718055994530942f;
  for (int row = blockIdx.x * 8 + rg; row < T.M; row += gridDim.x * 8) {
    float z = sb[lane];
    for (int s = 0; s < T.nslice; ++s) z += T.part[((size_t)s * T.M + row) * 32 + lane];
    float h = z / (1.0f + expf(-z));
    for (int l = 1; l < T.nh; ++l) {
      const float* w = sw + (l - 1) * 1024;
      float acc = sb[l * 32 + lane];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc = fmaf(__shfl(h, k, 32), w[k * 32 + lane], acc);
      h = acc / (1.0f + expf(-acc));
    }
    float o0 = sb[T.nh * 32 + lane], o1 = sb[T.nh * 32 + 32 + lane];
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float hk = __shfl(h, k, 32); o0 = fmaf(hk, swh[k * 64 + lane], o0); o1 = fmaf(hk, swh[k * 64 + 32 + lane], o1); }
    if (T.logits) {
      if (lane < T.P) T.logits[(size_t)row * T.P + lane] = o0;
      if (lane + 32 < T.P) T.logits[(size_t)row * T.P + 32 + lane] = o1;
    }
    // lane a needs logits[a] (its own o0) and logits[A + a]: o0 of lane A + a, or o1 of lane A + a - 32
    const float s_lo = __shfl(o0, (T.A + lane) & 31, 32), s_hi = __shfl(o1, (T.A + lane - 32) & 31, 32);
    const float sraw = T.A + lane < 32 ? s_lo : s_hi;
    float lp = 0.0f;
    if (lane < T.A) {
      const float loc = o0;
      if (T.noise) {
        const float scale = rr_softplus(sraw) + T.min_std;
        const float eps = T.noise[(size_t)row * T.A + lane], raw = loc + scale * eps;
        const float zz = (raw - loc) / scale;
        lp = -0.5f * zz * zz - logf(scale) - HALF_LOG_2PI - 2.0f * (LOG2 - raw - rr_softplus(-2.0f * raw));
        if (T.raw) T.raw[(size_t)row * T.A + lane] = raw;
        T.action[(size_t)row * T.A + lane] = tanhf(raw);
      } else {
        T.action[(size_t)row * T.A + lane] = tanhf(loc);
      }
    }
    if (T.noise && T.logp) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 32);
      if (lane == 0) T.logp[row] = lp;
    }
  }
}

// ------------------------------------------------------------------------------------------ observation normaliser: the update's sums
// brax.training.acme.running_statistics.update [UP; SURVEY.md a24, App. E] needs, per observation column, S1 = sum_x (x - mean_old)
// and sum_x (x - mean_old)(x - mean_new) over all transitions of a training step (1.3 M rows x 1263 columns = 6.6 GB at the launcher's
// sizes).  With d = mean_new - mean_old the second sum is S2 - d S1, S2 = sum_x (x - mean_old)^2, so ONE pass over the observations
// yields both (the tensor-expression form made two 6.6 GB temporaries and ~10 elementwise / reduction passes per training step).
// The rows are addressed inside the unroll buffer [nseq][Tp1][K]: the first T of every Tp1 rows (the bootstrap row is not a transition).
// Block b sums its rows for all K columns in double (thread = column, coalesced 4-byte reads across the columns); a second launch adds the
// block partials in order (fixed order: deterministic, no atomics).
struct RRMomArgs {
  const float* obs; const float* mean;
  long long nrows;          // nseq * T transitions
  int Tp1, T, K, rows_per_block, nblk;
  double* part;             // [nblk][2][K]
  double* out;              // [2][K]: S1 | S2
};
__global__ __launch_bounds__(256) void rr_obs_moments_kernel(const RRMomArgs A) {
  const long long r0 = (long long)blockIdx.x * A.rows_per_block;
  const long long r1 = r0 + A.rows_per_block < A.nrows ? r0 + A.rows_per_block : A.nrows;
  for (int k = threadIdx.x; k < A.K; k += 256) {
    const double m = (double)A.mean[k];
    double s1 = 0.0, s2 = 0.0;
    long long seq = r0 / A.T;
    int t = (int)(r0 - seq * A.T);
    const float* p = A.obs + ((size_t)seq * A.Tp1 + t) * A.K + k;
    for (long long r = r0; r < r1; ++r) {
      const double d = (double)*p - m;      // the difference itself in double: the sums are exact to double round-off
      s1 += d;
      s2 += d * d;
      if (++t == A.T) { t = 0; p += (size_t)(A.Tp1 - A.T + 1) * A.K; } else p += A.K;
    }
    A.part[((size_t)blockIdx.x * 2) * A.K + k] = s1;
    A.part[((size_t)blockIdx.x * 2 + 1) * A.K + k] = s2;
  }
}
__global__ __launch_bounds__(256) void rr_obs_moments_reduce_kernel(const RRMomArgs A) {
  const int e = blockIdx.x * 256 + threadIdx.x;        // 0 .. 2K-1
  if (e >= 2 * A.K) return;
  double s = 0.0;
  for (int b = 0; b < A.nblk; ++b) s += A.part[(size_t)b * 2 * A.K + e];
  A.out[e] = s;
}
