// train.hip -- the optimiser step on the device (second half; first version).
//
//   batch_updates!(tr, n)                      src/learning.jl:131-141
//   Network.train!(callback, nn, opt, ...)     src/networks/flux.jl:68-95   (Flux.withgradient + Flux.update!)
//   losses                                      src/learning.jl:67-90
//   ResNet in TRAIN mode                        src/networks/architectures/resnet.jl:53-92 (BatchNorm with batch statistics)
//
// Structure of this version: the F -> F tower convolutions run on hand-written MFMA kernels in all three passes --
// forward and data gradient on k_conv16_layer (the tower's implicit GEMM as a stand-alone layer), the weight gradient
// on k_wgrad16 (resnet16.h); the stem (K = 27), the 1x1 head convolutions and the dense layers are small plain fp32
// GEMMs on a hand-written MFMA kernel (gemm.h; im2col by a hand-written gather where needed);
// everything that is not a GEMM is hand-written here too: batch-norm statistics / apply / backward with
// deterministic two-stage column reductions, ReLU and skip wiring, the loss with its analytic gradient (softmax,
// mask normalisation, KL, invalid-mass penalty, tanh / MSE), parameter layout maps between Flux's arrays and the
// GEMM matrices, Adam / Nesterov and the running statistics.  No library call is left in the step (round 1 used a
// dlopen'ed rocBLAS for the small GEMMs).  The fused inference tower is untouched: after training the new
// parameters go back through az_net_set_params.
//
#include "engine.h"

// ------------------------------------------------------------------------------------------ small GEMMs
#include "gemm.h"
struct az_trainer;
static int tr_gemm(az_trainer* t, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                   float beta, float* C, int ldc);

// ------------------------------------------------------------------------------------------ kernels
// work[j] = blob[map[j]]  /  gblob[map[j]] = gwork[j]   (parameter layout maps, built on the host once)
__global__ void k_tr_gather(const float* __restrict__ blob, const int* __restrict__ map, long long n, float* __restrict__ work) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) work[j] = blob[map[j]];
}
// gblob[map[j]] = gwork[j] for the working entries j = list[i] that carry a gradient (one launch for every layer)
__global__ void k_tr_scatter(const float* __restrict__ gwork, const int* __restrict__ map, const int* __restrict__ list, long long n, float* __restrict__ gblob) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int j = list[i]; gblob[map[j]] = gwork[j]; }
}
// the batch of a step: rows idx[0..B) of the data set's tensors
__global__ void k_tr_batch(const int* __restrict__ idx, int B, int xs, int A, const float* __restrict__ W, const float* __restrict__ X,
                           const float* __restrict__ Am, const float* __restrict__ P, const float* __restrict__ V,
                           float* bW, float* bX, float* bA, float* bP, float* bV) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const size_t s = (size_t)idx[b];
  for (int i = threadIdx.x; i < xs; i += blockDim.x) bX[(size_t)b * xs + i] = X[s * xs + i];
  for (int i = threadIdx.x; i < A; i += blockDim.x) { bA[(size_t)b * A + i] = Am[s * A + i]; bP[(size_t)b * A + i] = P[s * A + i]; }
  if (threadIdx.x == 0) { bW[b] = W[s]; bV[b] = V[s]; }
}
template <bool PLANES>
__global__ void __launch_bounds__(256) k_tr_im2col(const float* __restrict__ in, long long R, int C, int Wd, int Hd, float* __restrict__ col) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = R * 9 * C;
  if (t >= total) return;
  const int c = (int)(t % C);
  const int tap = (int)((t / C) % 9);
  const long long r = t / (9LL * C);
  const int P = Wd * Hd, q = (int)(r % P), x = q % Wd, y = q / Wd;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  float v = 0.0f;
  if (y + dy >= 0 && y + dy < Hd && x + dx >= 0 && x + dx < Wd) {
    const long long rr = r + dy * Wd + dx;
    v = PLANES ? in[((rr / P) * C + c) * P + (rr % P)] : in[rr * C + c];
  }
  col[t] = v;
}
// two-stage column sums over R rows of up to two derived quantities.  Stage 1: a block = 64 channels x 4 row lanes takes
// a chunk of 64 rows (thread: 16 rows, then the 4 lanes are added in lane order); stage 2: one block per channel adds
// the chunk partials with a fixed-shape tree.  Deterministic, double accumulators.
//   MODE 0: (x, x*x)                       batch-norm statistics of the GEMM output
//   MODE 1: (dy, dy * xhat)                batch-norm backward, dy = da * (out > 0), xhat = (g - mean) * invstd
//   MODE 2: (x, 0)                         bias gradients of the dense layers
constexpr int TR_CHUNK = 64;
template <int MODE>
__global__ void __launch_bounds__(256) k_tr_colsum(const float* __restrict__ x, const float* __restrict__ out_act, const float* __restrict__ g,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd, long long R, int C,
                                                   double* __restrict__ part /* [nchunks][2][C] */, TrFinal fin) {
  __shared__ double sh[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const long long r0 = (long long)blockIdx.x * TR_CHUNK, r1 = r0 + TR_CHUNK < R ? r0 + TR_CHUNK : R;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    float mu = 0.f, is = 0.f;
    if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
    for (long long r = r0 + rl; r < r1; r += 4) {
      const size_t i = (size_t)r * C + c;
      if (MODE == 0) { const float v = x[i]; s0 += (double)v; s1 += (double)v * (double)v; }
      else if (MODE == 1) {
        const float dy = out_act[i] > 0.0f ? x[i] : 0.0f;
        const float xh = (g[i] - mu) * is;
        s0 += (double)dy; s1 += (double)dy * (double)xh;
      } else s0 += (double)x[i];
    }
  }
  sh[0][rl][cl] = s0; sh[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    part_store(part + ((size_t)blockIdx.x * 2) * C + c, ((sh[0][0][cl] + sh[0][1][cl]) + sh[0][2][cl]) + sh[0][3][cl]);
    part_store(part + ((size_t)blockIdx.x * 2 + 1) * C + c, ((sh[1][0][cl] + sh[1][1][cl]) + sh[1][2][cl]) + sh[1][3][cl]);
  }
  __syncthreads();
  if (C <= 128) tr_finish(part, C, fin, &sh[0][0][0]);               // 512 doubles of scratch: 256 + 2 C (host: wider layers keep the separate launch)
}
// MODE 1 with four channels per thread (C % 4 == 0, 256 % (C/4) == 0): a block = C/4 channel quads x 256/(C/4) row lanes
// over the same 64-row chunk, 16-byte loads; the row lanes are added in lane order through 8 KB of LDS (it has to fit
// beside two workgroups of k_wgrad16, and the first form of that kernel left 10 KB).  Same partial layout as k_tr_colsum.
__global__ void __launch_bounds__(256) k_tr_colsum1v(const float* __restrict__ x, const float* __restrict__ out_act, const float* __restrict__ g,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd, long long R, int C,
                                                     double* __restrict__ part /* [nchunks][2][C] */, TrFinal fin) {
  __builtin_amdgcn_s_setprio(3);   // HBM-bound and short: its few instructions go first when it shares a SIMD with k_wgrad16's wavefronts
  __shared__ double sh[256][4];
  const int CQ = C >> 2, RL = 256 / CQ;
  const int q = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const long long r0 = (long long)blockIdx.x * TR_CHUNK, r1 = r0 + TR_CHUNK < R ? r0 + TR_CHUNK : R;
  const float4 mu = *(const float4*)(mean + 4 * q), is = *(const float4*)(invstd + 4 * q);
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  auto add = [&](const float4& vx, const float4& va, const float4& vg) {
    const float dx[4] = {vx.x, vx.y, vx.z, vx.w}, av[4] = {va.x, va.y, va.z, va.w}, gv[4] = {vg.x, vg.y, vg.z, vg.w};
    const float m[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dy = av[k] > 0.0f ? dx[k] : 0.0f;
      const float xh = (gv[k] - m[k]) * iv[k];
      s0[k] += (double)dy; s1[k] += (double)dy * (double)xh;
    }
  };
  long long r = r0 + rl;
  for (; r + 3 * RL < r1; r += 4 * RL) {                            // four rows in flight: beside k_wgrad16 this kernel gets one wavefront per SIMD
    const size_t i0 = (size_t)r * C + 4 * q, i1 = i0 + (size_t)RL * C, i2 = i1 + (size_t)RL * C, i3 = i2 + (size_t)RL * C;
    const float4 x0 = *(const float4*)(x + i0), a0 = *(const float4*)(out_act + i0), g0 = *(const float4*)(g + i0);
    const float4 x1 = *(const float4*)(x + i1), a1 = *(const float4*)(out_act + i1), g1 = *(const float4*)(g + i1);
    const float4 x2 = *(const float4*)(x + i2), a2 = *(const float4*)(out_act + i2), g2 = *(const float4*)(g + i2);
    const float4 x3 = *(const float4*)(x + i3), a3 = *(const float4*)(out_act + i3), g3 = *(const float4*)(g + i3);
    add(x0, a0, g0); add(x1, a1, g1); add(x2, a2, g2); add(x3, a3, g3);
  }
  for (; r + RL < r1; r += 2 * RL) {                                // two rows in flight
    const size_t i = (size_t)r * C + 4 * q, j = i + (size_t)RL * C;
    const float4 x0 = *(const float4*)(x + i), a0 = *(const float4*)(out_act + i), g0 = *(const float4*)(g + i);
    const float4 x1 = *(const float4*)(x + j), a1 = *(const float4*)(out_act + j), g1 = *(const float4*)(g + j);
    add(x0, a0, g0); add(x1, a1, g1);
  }
  for (; r < r1; r += RL) {
    const size_t i = (size_t)r * C + 4 * q;
    add(*(const float4*)(x + i), *(const float4*)(out_act + i), *(const float4*)(g + i));
  }
#pragma unroll
  for (int w = 0; w < 2; ++w) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = w == 0 ? s0[k] : s1[k];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double v = sh[q][k];
        for (int l = 1; l < RL; ++l) v += sh[q + l * CQ][k];
        part_store(part + ((size_t)blockIdx.x * 2 + w) * C + 4 * q + k, v);
      }
    }
    __syncthreads();
  }
  tr_finish(part, C, fin, &sh[0][0]);                                // 1024 doubles of scratch >= 256 + 2 C (C <= 256 here... the host checks)
}
// the second stage of a column sum: one workgroup per channel adds the producers' partials in a fixed order and does what the
// mode says (TrFinal, resnet16.h).  A launch of its own after every producer: riding in the producer's LAST workgroup (tr_finish)
// was built and measured in round 4 -- one workgroup reading 256 ... 672 x 2 C partials through cache-bypassing loads takes
// longer (3.4 ms per step at 5x64 instead of 1.9) than the ~6 us a dependent launch costs.
__global__ void __launch_bounds__(64) k_tr_colsum_final(const double* __restrict__ part, int nchunks, int C, double* __restrict__ sums /* [2][C] */, TrFinal f) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ double sh[2][64];
  const int c = blockIdx.x, t = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int k = t; k < nchunks; k += 64) { s0 += part[((size_t)k * 2) * C + c]; s1 += part[((size_t)k * 2 + 1) * C + c]; }
  sh[0][t] = s0; sh[1][t] = s1;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (t < w) { sh[0][t] += sh[0][t + w]; sh[1][t] += sh[1][t + w]; }
    __syncthreads();
  }
  if (t == 0) {
    const double s0 = sh[0][0], s1 = sh[1][0];
    sums[c] = s0; sums[C + c] = s1;
    if (f.mode == 1) {
      // batch statistics -> mean, 1/sqrt(var + eps) (Flux BatchNorm: biased variance, eps 1e-5) and the running statistics
      // mu <- (1-m) mu + m (mean + bias), var <- (1-m) var + m * var * R/(R-1)
      const double m = s0 / (double)f.R;
      double var = s1 / (double)f.R - m * m;
      if (var < 0.0) var = 0.0;
      f.mean[c] = (float)m;
      f.invstd[c] = (float)(1.0 / __builtin_sqrt(var + 1e-5));
      const float b = f.bias ? f.bias[c] : 0.0f;
      f.run_mean[c] = (1.0f - f.momentum) * f.run_mean[c] + f.momentum * ((float)m + b);
      f.run_var[c] = (1.0f - f.momentum) * f.run_var[c] + f.momentum * (float)(var * ((double)f.R / (double)(f.R - 1)));
    } else if (f.mode == 2) { f.dbeta[c] = (float)s0; f.dgamma[c] = (float)s1; f.mf[c] = (float)(s0 / (double)f.R); f.mf[C + c] = (float)(s1 / (double)f.R); }
    else if (f.mode == 3) f.out0[c] = (float)s0;
  }
}
// a = relu(gamma * xhat + beta (+ res))
__global__ void k_tr_bn_apply(const float* __restrict__ g, const float* __restrict__ mean, const float* __restrict__ invstd,
                              const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res,
                              long long n, int C, float* __restrict__ a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  float v = gamma[c] * ((g[i] - mean[c]) * invstd[c]) + beta[c];
  if (res) v += res[i];
  a[i] = v > 0.0f ? v : 0.0f;
}
// batch-norm backward: dy = da * (a > 0); dg = gamma * invstd * (dy - sum(dy)/R - xhat * sum(dy*xhat)/R); dgamma = sum(dy*xhat),
// dbeta = sum(dy).  `dy_out` (optional) receives dy: the gradient that also flows into the skip connection.  mf = the two
// means as floats (k_tr_colsum_final, mode 2).  V = 4: four channels per thread (C % 4 == 0), the form that still moves
// bytes when it shares the chip with k_wgrad16 and gets one wavefront per SIMD.
template <int V>
__global__ void __launch_bounds__(256) k_tr_bn_bwd(const float* __restrict__ da, const float* __restrict__ a, const float* __restrict__ g,
                            const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                            const float* __restrict__ mf, long long n, int C, float* __restrict__ dg, float* __restrict__ dy_out) {
  __builtin_amdgcn_s_setprio(3);
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (i >= n) return;
  const int c = (int)(i % C);
  float vda[V], va[V], vg[V], o[V], dyv[V];
  if constexpr (V == 4) {
    const float4 x = *(const float4*)(da + i), y = *(const float4*)(a + i), z = *(const float4*)(g + i);
    vda[0] = x.x; vda[1] = x.y; vda[2] = x.z; vda[3] = x.w; va[0] = y.x; va[1] = y.y; va[2] = y.z; va[3] = y.w; vg[0] = z.x; vg[1] = z.y; vg[2] = z.z; vg[3] = z.w;
  } else { vda[0] = da[i]; va[0] = a[i]; vg[0] = g[i]; }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const float dy = va[k] > 0.0f ? vda[k] : 0.0f;
    const float xh = (vg[k] - mean[c + k]) * invstd[c + k];
    o[k] = gamma[c + k] * invstd[c + k] * (dy - mf[c + k] - xh * mf[C + c + k]);
    dyv[k] = dy;
  }
  if constexpr (V == 4) {
    *(float4*)(dg + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (dy_out) *(float4*)(dy_out + i) = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
  } else { dg[i] = o[0]; if (dy_out) dy_out[i] = dyv[0]; }
}
// dense epilogues: y = [relu](x + bias[c]);  backward mask: dx = dy * (y > 0)
__global__ void k_tr_bias_act(float* __restrict__ x, const float* __restrict__ bias, long long n, int C, int relu) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i] + bias[(int)(i % C)];
  x[i] = relu ? (v > 0.0f ? v : 0.0f) : v;
}
__global__ void k_tr_relu_bwd(float* __restrict__ dy, const float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(y[i] > 0.0f)) dy[i] = 0.0f;
}
// `losses` (learning.jl:67-90) for one batch, forward terms and the analytic gradient w.r.t. the policy logits and the
// value pre-activation.  scale = (mean(W_batch) / Wmean) / sum(W_batch) is folded into the gradients.
//   p = softmax(logits); u = p .* A; S = sum(u); q = u / (S + eps)            (network.jl:264-271)
//   Lp_i = -sum_a P_a log(q_a + eps) w;  Linv_i = (1 - S) w;  Lv_i = ((tanh(t) - V) / rho)^2 w
__global__ void k_tr_loss(const float* __restrict__ logits, const float* __restrict__ tpre, const float* __restrict__ Am,
                          const float* __restrict__ P, const float* __restrict__ V, const float* __restrict__ W, int B, int A,
                          float cinv, float rho, float gscale, float* __restrict__ dlogits, float* __restrict__ dt,
                          double* __restrict__ t_w, double* __restrict__ t_kl, double* __restrict__ t_mse, double* __restrict__ t_inv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float eps = 1.1920929e-07f;
  float p[AZ_MAX_ACTIONS], u[AZ_MAX_ACTIONS], gq[AZ_MAX_ACTIONS];
  float mx = logits[(size_t)i * A];
  for (int a = 1; a < A; ++a) mx = logits[(size_t)i * A + a] > mx ? logits[(size_t)i * A + a] : mx;
  float se = 0.0f;
  for (int a = 0; a < A; ++a) { p[a] = az_expf(logits[(size_t)i * A + a] - mx); se += p[a]; }
  float S = 0.0f;
  for (int a = 0; a < A; ++a) { p[a] = p[a] / se; u[a] = p[a] * Am[(size_t)i * A + a]; S += u[a]; }
  const float w = W[i];
  const float den = S + eps;
  double kl = 0.0;
  float gu_sum = 0.0f;                                             // sum_b gq_b * u_b / den^2
  for (int a = 0; a < A; ++a) {
    const float q = u[a] / den;
    const float Pa = P[(size_t)i * A + a];
    kl += (double)(Pa * az_logf(q + eps) * w);
    gq[a] = -Pa / (q + eps);                                       // dLp_i/dq_a (per unit weight)
    gu_sum += gq[a] * u[a];
  }
  gu_sum = gu_sum / (den * den);
  // dL/dp_a = A_a * (gq_a / den - gu_sum - cinv) * w * gscale ;  softmax backward
  float dp[AZ_MAX_ACTIONS], dot = 0.0f;
  for (int a = 0; a < A; ++a) {
    dp[a] = Am[(size_t)i * A + a] * (gq[a] / den - gu_sum - cinv) * w * gscale;
    dot += p[a] * dp[a];
  }
  for (int a = 0; a < A; ++a) dlogits[(size_t)i * A + a] = p[a] * (dp[a] - dot);
  const float vh = az_tanhf(tpre[i]);
  const float d = vh / rho - V[i] / rho;
  dt[i] = 2.0f * d / rho * (1.0f - vh * vh) * w * gscale;
  t_w[i] = (double)w; t_kl[i] = kl; t_mse[i] = (double)(d * d * w); t_inv[i] = (double)((1.0f - S) * w);
}
// the five sums of a step in one launch: out = (sum w, sum kl, sum mse, sum inv, sum of squares of the trainables)
__global__ void __launch_bounds__(256) k_tr_step_sums(const double* __restrict__ tw, const double* __restrict__ tkl, const double* __restrict__ tmse,
                                                      const double* __restrict__ tinv, int B, const double* __restrict__ ssq_part, int nssq,
                                                      double* __restrict__ out) {
  __shared__ double sh[5][256];
  double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < B; i += 256) { a[0] += tw[i]; a[1] += tkl[i]; a[2] += tmse[i]; a[3] += tinv[i]; }
  for (int i = threadIdx.x; i < nssq; i += 256) a[4] += ssq_part[i];
#pragma unroll
  for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
#pragma unroll
      for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x < 5) out[threadIdx.x] = sh[threadIdx.x][0];
}
// Adam (Optimisers.Adam: beta (0.9, 0.999), eps 1e-8) / Nesterov (Optimisers.Nesterov) on the trainable entries of
// the blob; the L2 term of the loss, scale * creg * sum(w^2), is added to the gradient here: + scale * 2 creg w.
__global__ void k_tr_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                          const unsigned char* __restrict__ trainable, long long n, float lr, float b1t, float b2t,
                          const double* __restrict__ sums, float reg_c /* 2 creg / (B Wmean) */) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !trainable[i]) return;
  const float reg2 = (float)sums[0] * reg_c;                       // scale * 2 creg, scale = mean(W_batch) / Wmean
  const float gi = g[i] + reg2 * w[i];
  const float mt = 0.9f * m[i] + (1.0f - 0.9f) * gi;
  const float vt = 0.999f * v[i] + (1.0f - 0.999f) * gi * gi;
  m[i] = mt; v[i] = vt;
  w[i] -= mt / (1.0f - b1t) / (__builtin_sqrtf(vt / (1.0f - b2t)) + 1e-8f) * lr;
}
__global__ void k_tr_nesterov(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ vel,
                              const unsigned char* __restrict__ trainable, long long n, float lr, float rho,
                              const double* __restrict__ sums, float reg_c) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !trainable[i]) return;
  const float reg2 = (float)sums[0] * reg_c;
  const float gi = g[i] + reg2 * w[i];
  const float vo = vel[i];
  const float d = rho * rho * vo - (1.0f + rho) * lr * gi;         // Optimisers.Nesterov: newdx = -d
  vel[i] = rho * vo - lr * gi;
  w[i] += d;
}
__global__ void k_tr_sumsq(const float* __restrict__ w, const unsigned char* __restrict__ trainable, long long n, double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) if (trainable[i]) s += (double)w[i] * (double)w[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// ------------------------------------------------------------------------------------------ host
struct TrConv {                 // one convolution + batch norm (3x3 of the tower, or a 1x1 head convolution)
  int cin, cout, taps;
  size_t off_w, off_b, off_bn;  // blob offsets: W, bias, (gamma, beta, mean, var)
  size_t wk_wm;                 // offset in the working-parameter array: GEMM matrix [taps*cin][cout]
  size_t wk_ffwd, wk_fdg;       // tower convolutions F -> F: MFMA fragments of k_conv16_layer for forward / data gradient
  bool mfma;                    // forward and data gradient on k_conv16_layer instead of im2col + GEMM
  float *col, *g, *a;           // im2col input [R][taps*cin] (taps == 1: alias of the input), GEMM output, activation
  float *mean, *invstd;
};
struct az_trainer {
  az_engine* e;
  az_dataset* d;
  az_train_cfg cfg;
  int game, device; GameInfo gi;
  hipStream_t stream;
  bool one_stream, wg_late;
  hipStream_t side;                                                          // the weight gradients of the tower run here, beside the batch-norm backward passes of the next layer
  std::vector<hipEvent_t> ev_dg, ev_wg;                                     // per tower layer: output gradient ready / weight gradient done
  float* gemm_ws; size_t gemm_ws_floats;                                     // split-reduction workspace of gemm_f32
  int B, nblocks, F, npf, nvf, nA;
  long long R;
  size_t nparams;
  std::vector<TrConv> convs;    // stem, block convs..., then policy-head conv, value-head conv
  size_t off_pd_w, off_pd_b, off_v1_w, off_v1_b, off_v2_w, off_v2_b;      // dense layers in the blob
  size_t wk_pd, wk_v1, wk_v2, nwork;
  float *blob, *gblob, *opt_m, *opt_v; unsigned char* trainable;           // [nparams]
  int* scat; long long nscat;                                                // working entries that carry a gradient (the primary ranges)
  float *work, *gwork; int* map;                                             // working parameters and their blob map
  // batch + head buffers
  int* d_idx; float *bW, *bX, *bA, *bP, *bV;
  float *logits, *v1, *tpre, *dlogits, *dt, *dv1;
  float *dact, *dact2, *dact3, *dcol;                                        // [R][F] gradients (dcol / dact / dact3 take turns as the running gradient, dact2 = the skip share)
  float* wg_part; int wg_splits, wg_bpw;                       // k_wgrad16: partial dW per row split, boards per workgroup
  double *part, *sums, *terms, *bsums;
  float* bn_mf;                                                              // [2][C] the batch-norm backward means of the layer in flight
  bool conv_nt6;                                                             // AZHIP_TRAIN_NT6=0 switches the 6-tile layer kernel off (A/B)
  bool fin_inside;                                                           // AZHIP_TRAIN_FINISH_INSIDE=1: second stage of the column sums in the producer's last workgroup (measured slower: off)
  int* fin_counter;                                                          // tr_finish: counter of the workgroups that have left their partial sums
  std::vector<void*> allocs;
  std::vector<int> perm; int64_t perm_pos, epoch; int64_t step;
  float b1t, b2t;
};

template <class T> static int tr_alloc(az_trainer* t, T** p, size_t n, bool zero = false) {
  AZCHK(mem_alloc(&t->allocs, p, n));
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, std::max<size_t>(n * sizeof(T), 16), t->stream));
  return AZ_OK;
}
static inline unsigned tr_grid(long long n) { return (unsigned)((n + 255) / 256); }
static int tr_gemm(az_trainer* t, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                   float beta, float* C, int ldc) {
  gemm_f32(t->stream, t->gemm_ws, t->gemm_ws_floats, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  return AZ_OK;
}

extern "C" int az_train_cfg_init(az_train_cfg* c) {
  if (!c) return fail(AZ_ERR_BAD_ARG, "cfg is NULL");
  memset(c, 0, sizeof *c);
  c->struct_size = (int32_t)sizeof *c;
  c->optimiser = AZ_OPT_ADAM; c->lr = 2e-3f;                       // games/connect-four/params.jl:46-58
  c->lr_base = 1e-3f; c->lr_high = 1e-2f; c->lr_low = 1e-3f; c->momentum_low = 0.8f; c->momentum_high = 0.9f;
  c->l2_regularization = 1e-4; c->nonvalidity_penalty = 1.0; c->rewards_renormalization = 1.0;
  c->batch_size = 1024; c->batch_norm_momentum = 0.1f; c->seed = 1;
  return AZ_OK;
}

extern "C" int az_trainer_destroy(az_trainer* t) {
  if (!t) return AZ_OK;
  (void)hipSetDevice(t->device);
  for (void* p : t->allocs) (void)hipFree(p);
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
  for (hipEvent_t ev : t->ev_dg) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : t->ev_wg) (void)hipEventDestroy(ev);
  if (t->stream) (void)hipStreamDestroy(t->stream);
  delete t;
  return AZ_OK;
}

// builds layer tables, parameter maps and buffers
static int trainer_build(az_trainer* t) {
  const GameInfo& gi = t->gi;
  const int F = t->F, C = gi.C, P = gi.P, A = gi.A, npf = t->npf, nvf = t->nvf;
  const long long R = t->R;
  size_t off = 0, wk = 0;
  auto add_conv = [&](int cin, int cout, int taps) {
    TrConv c{};
    c.cin = cin; c.cout = cout; c.taps = taps;
    c.off_w = off; off += (size_t)taps * cin * cout;
    c.off_b = off; off += cout;
    c.off_bn = off; off += 4 * (size_t)cout;
    c.wk_wm = wk; wk += (size_t)taps * cin * cout;
    c.mfma = taps == 9 && cin == cout;                               // the tower's F -> F convolutions (F = 64 or 128: az_engine_create): MFMA layer kernels
    c.wk_ffwd = c.wk_fdg = 0;
    if (c.mfma) { c.wk_ffwd = wk; wk += (size_t)taps * cin * cout; c.wk_fdg = wk; wk += (size_t)taps * cin * cout; }
    t->convs.push_back(c);
  };
  add_conv(C, F, 9);
  for (int l = 0; l < 2 * t->nblocks; ++l) add_conv(F, F, 9);
  // tr_forward_backward takes the input of every MFMA layer from the layer below (bn_of(l - 1)) and the stem's from the batch's planes:
  // that holds only while the stem is the one non-MFMA tower layer.  A game with as many planes as filters would make it one.
  t->convs[0].mfma = false; t->convs[0].wk_ffwd = t->convs[0].wk_fdg = 0;
  for (int l = 1; l <= 2 * t->nblocks; ++l)
    if (!t->convs[l].mfma) return fail(AZ_ERR_STATE, "trainer: tower layer %d is not an MFMA layer", l);
  add_conv(F, npf, 1);                                              // policy head: conv, (bn), then dense
  t->off_pd_w = off; off += (size_t)A * P * npf; t->off_pd_b = off; off += A;
  add_conv(F, nvf, 1);                                              // value head
  t->off_v1_w = off; off += (size_t)F * P * nvf; t->off_v1_b = off; off += F;
  t->off_v2_w = off; off += F; t->off_v2_b = off; off += 1;
  if (off != t->nparams) return fail(AZ_ERR_STATE, "trainer layout (%zu) does not match the network (%zu parameters)", off, t->nparams);
  t->wk_pd = wk; wk += (size_t)P * npf * A;
  t->wk_v1 = wk; wk += (size_t)P * nvf * F;
  t->wk_v2 = wk; wk += F;
  t->nwork = wk;
  // parameter maps: work[j] = blob[map[j]]
  std::vector<int> map(wk);
  std::vector<unsigned char> trainable(t->nparams, 1);
  for (const TrConv& c : t->convs) {
    for (int tap = 0; tap < c.taps; ++tap) {
      const int dy = c.taps == 9 ? tap / 3 - 1 : 0, dx = c.taps == 9 ? tap % 3 - 1 : 0;
      const int wi = c.taps == 9 ? 1 - dx : 0, wj = c.taps == 9 ? 1 - dy : 0, ks = c.taps == 9 ? 3 : 1;
      for (int ci = 0; ci < c.cin; ++ci) for (int co = 0; co < c.cout; ++co) {
        const int b = (int)(c.off_w + (size_t)wi + (size_t)ks * (wj + (size_t)ks * (ci + (size_t)c.cin * co)));   // Flux W(kw, kh, ci, co), true convolution
        map[c.wk_wm + ((size_t)tap * c.cin + ci) * c.cout + co] = b;
      }
    }
    if (c.mfma) {
      // fragment order of conv16 (see az_net_set_params): lane ln of float4 sq, component q, column tile ct supplies input
      // channel ci = (g & 1) F/2 + 2 (4 sq + q) + (g >> 1), g = ln >> 4, for output channel co = 16 ct + (ln & 15)
      const int Fc = c.cin, CT = Fc / 16;
      auto blob_of = [&](int tap, int ci, int co) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, wi = 1 - dx, wj = 1 - dy;
        return (int)(c.off_w + (size_t)wi + 3 * (wj + 3 * (ci + (size_t)Fc * co)));
      };
      for (int tap = 0; tap < 9; ++tap) for (int ct = 0; ct < CT; ++ct) for (int sq = 0; sq < CT; ++sq) for (int ln = 0; ln < 64; ++ln) for (int q = 0; q < 4; ++q) {
        const int sidx = 4 * sq + q, g = ln >> 4, kin = (g & 1) * (Fc / 2) + 2 * sidx + (g >> 1), nout = ct * 16 + (ln & 15);
        const size_t j = ((((size_t)tap * CT + ct) * CT + sq) * 64 + ln) * 4 + q;
        map[c.wk_ffwd + j] = blob_of(tap, kin, nout);                  // forward: in = ci, out = co
        map[c.wk_fdg + j] = blob_of(8 - tap, nout, kin);               // data gradient: in = co, out = ci, taps mirrored
      }
    }
    for (int i = 0; i < 2 * c.cout; ++i) trainable[c.off_bn + 2 * c.cout + i] = 0;                              // running mean / variance
  }
  for (int p = 0; p < P; ++p) for (int f = 0; f < npf; ++f) for (int a = 0; a < A; ++a)
    map[t->wk_pd + ((size_t)p * npf + f) * A + a] = (int)(t->off_pd_w + a + (size_t)A * (p + (size_t)P * f));  // Dense W(out, in), in = p + P f (Flux.flatten of WHC)
  for (int p = 0; p < P; ++p) for (int f = 0; f < nvf; ++f) for (int o = 0; o < F; ++o)
    map[t->wk_v1 + ((size_t)p * nvf + f) * F + o] = (int)(t->off_v1_w + o + (size_t)F * (p + (size_t)P * f));
  for (int o = 0; o < F; ++o) map[t->wk_v2 + o] = (int)(t->off_v2_w + o);
  std::vector<int> scat;
  for (const TrConv& c : t->convs) for (size_t j = 0; j < (size_t)c.taps * c.cin * c.cout; ++j) scat.push_back((int)(c.wk_wm + j));
  for (size_t j = t->wk_pd; j < wk; ++j) scat.push_back((int)j);
  t->nscat = (long long)scat.size();
  AZCHK(tr_alloc(t, &t->scat, scat.size()));
  HIPCHK(hipMemcpyAsync(t->scat, scat.data(), sizeof(int) * scat.size(), hipMemcpyHostToDevice, t->stream));
  AZCHK(tr_alloc(t, &t->map, wk)); AZCHK(tr_alloc(t, &t->work, wk)); AZCHK(tr_alloc(t, &t->gwork, wk, true));
  AZCHK(tr_alloc(t, &t->blob, t->nparams)); AZCHK(tr_alloc(t, &t->gblob, t->nparams, true));
  AZCHK(tr_alloc(t, &t->opt_m, t->nparams, true)); AZCHK(tr_alloc(t, &t->opt_v, t->nparams, true));
  AZCHK(tr_alloc(t, &t->trainable, t->nparams));
  HIPCHK(hipMemcpyAsync(t->map, map.data(), sizeof(int) * wk, hipMemcpyHostToDevice, t->stream));
  HIPCHK(hipMemcpyAsync(t->trainable, trainable.data(), t->nparams, hipMemcpyHostToDevice, t->stream));
  HIPCHK(hipMemcpyAsync(t->blob, t->e->blob.data(), sizeof(float) * t->nparams, hipMemcpyHostToDevice, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));                         // host vectors go out of scope
  // activations
  const int ntower = 1 + 2 * t->nblocks;
  for (size_t l = 0; l < t->convs.size(); ++l) {
    TrConv& c = t->convs[l];
    if (c.taps == 9 && !c.mfma) AZCHK(tr_alloc(t, &c.col, (size_t)R * 9 * c.cin)); else c.col = nullptr;   // the stem's im2col (K = 9 C)
    AZCHK(tr_alloc(t, &c.g, (size_t)R * c.cout)); AZCHK(tr_alloc(t, &c.a, (size_t)R * c.cout));
    AZCHK(tr_alloc(t, &c.mean, c.cout)); AZCHK(tr_alloc(t, &c.invstd, c.cout));
  }
  for (int l = 0; l < ntower; ++l) {
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming)); t->ev_dg.push_back(a);
    HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming)); t->ev_wg.push_back(b);
  }
  const int B = t->B;
  AZCHK(tr_alloc(t, &t->d_idx, B)); AZCHK(tr_alloc(t, &t->bW, B)); AZCHK(tr_alloc(t, &t->bV, B));
  AZCHK(tr_alloc(t, &t->bX, (size_t)B * C * P)); AZCHK(tr_alloc(t, &t->bA, (size_t)B * A)); AZCHK(tr_alloc(t, &t->bP, (size_t)B * A));
  AZCHK(tr_alloc(t, &t->logits, (size_t)B * A)); AZCHK(tr_alloc(t, &t->dlogits, (size_t)B * A));
  AZCHK(tr_alloc(t, &t->v1, (size_t)B * F)); AZCHK(tr_alloc(t, &t->dv1, (size_t)B * F));
  AZCHK(tr_alloc(t, &t->tpre, B)); AZCHK(tr_alloc(t, &t->dt, B));
  AZCHK(tr_alloc(t, &t->dact, (size_t)R * F)); AZCHK(tr_alloc(t, &t->dact2, (size_t)R * F)); AZCHK(tr_alloc(t, &t->dact3, (size_t)R * F)); AZCHK(tr_alloc(t, &t->dcol, (size_t)R * F));
  const int nchunks = (int)((R + TR_CHUNK - 1) / TR_CHUNK);
  AZCHK(tr_alloc(t, &t->part, (size_t)std::max(nchunks, B + 1) * 2 * std::max(F, 64)));   // chunks of k_tr_colsum or workgroups of k_conv16_layer
  AZCHK(tr_alloc(t, &t->sums, (size_t)2 * std::max(F, 64))); AZCHK(tr_alloc(t, &t->bn_mf, (size_t)2 * std::max(F, 64)));
  { const char* fi = getenv("AZHIP_TRAIN_FINISH_INSIDE"); t->fin_inside = fi && atoi(fi) != 0; }
  { const char* n6 = getenv("AZHIP_TRAIN_NT6"); t->conv_nt6 = !(n6 && atoi(n6) == 0); }
  AZCHK(tr_alloc(t, &t->fin_counter, 4, true));                       // tr_finish: workgroups done (every launch leaves it at zero)
  AZCHK(tr_alloc(t, &t->terms, (size_t)4 * B)); AZCHK(tr_alloc(t, &t->bsums, 8 + 1024));
  // k_wgrad16: one round of workgroups over the chip
  t->wg_part = nullptr; t->wg_splits = 0; t->wg_bpw = 0;
  {
    const int tg = F == 128 ? 3 : 1, per_cu = 1;                    // 82 KB (64 filters) / 145 KB (128) of LDS: one workgroup per CU either way
    // as many workgroups per tap group as the chip holds (one round), the boards spread evenly over them: 1024 boards at
    // 128 filters = 85 workgroups x 3 tap groups with 12 or 13 boards each (whole 3-board LDS chunks per workgroup made it
    // 69 x 3 workgroups with 15 boards: a fifth of the CUs idle and a fifth chunk for everyone)
    t->wg_splits = std::max(1, std::min(B, t->e->num_cu * per_cu / tg));
    t->wg_bpw = (B + t->wg_splits - 1) / t->wg_splits;
    AZCHK(tr_alloc(t, &t->wg_part, (size_t)t->wg_splits * 9 * F * F));
  }
  return AZ_OK;
}

template <class Gm, int F, bool STATS, int NT> static int tr_conv16_f(az_trainer* t, const float* in, const float* frag, float* out, const float* addend, const BnIn& bn, const TrFinal& fin) {
  using T = T16<Gm, F, NT>;
  static bool attr_done = false;
  if (!attr_done) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv16_layer<Gm, F, STATS, false, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, T::BYTES)); attr_done = true; }
  hipLaunchKernelGGL((k_conv16_layer<Gm, F, STATS, false, NT>), dim3((t->B + T::TB - 1) / T::TB), dim3(T::THREADS), T::BYTES, t->stream, in, (const float4*)frag, out, t->B, t->e->d_geo[NT == 11 ? 0 : 4], t->part, addend, (long long*)nullptr, bn, fin);
  return AZ_OK;
}
// 3x3 F -> F convolution of [R][F] activations on the MFMA layer kernel; stats: also the first stage of the column sums
// (sum, sum of squares) of the output in t->part, *nparts workgroup partials
static int tr_conv16(az_trainer* t, const float* in, const float* frag, float* out, bool stats = false, int* nparts = nullptr, const float* addend = nullptr,
                     const BnIn& bn = BnIn{}, const TrFinal& fin = TrFinal{}) {
  DISPATCH_GAME(t->game, {
    using T = T16<Gm, 64, 11>;
    using T6 = T16<Gm, 64, 6>;
    // (r4) 64 filters and a batch that gives the 11-tile form at most one workgroup per CU: the 6-tile form (half the boards per
    // workgroup, two workgroups per CU) overlaps one workgroup's fill and epilogue with the other's products
    const bool small = t->F == 64 && t->conv_nt6 && (t->B + T::TB - 1) / T::TB <= (t->e->num_cu > 0 ? t->e->num_cu : 256);
    if (nparts) *nparts = small ? (t->B + T6::TB - 1) / T6::TB : (t->B + T::TB - 1) / T::TB;
    if (t->F == 128) { if (stats) AZCHK((tr_conv16_f<Gm, 128, true, 11>(t, in, frag, out, addend, bn, fin))); else AZCHK((tr_conv16_f<Gm, 128, false, 11>(t, in, frag, out, addend, bn, fin))); }
    else if (small) { if (stats) AZCHK((tr_conv16_f<Gm, 64, true, 6>(t, in, frag, out, addend, bn, fin))); else AZCHK((tr_conv16_f<Gm, 64, false, 6>(t, in, frag, out, addend, bn, fin))); }
    else { if (stats) AZCHK((tr_conv16_f<Gm, 64, true, 11>(t, in, frag, out, addend, bn, fin))); else AZCHK((tr_conv16_f<Gm, 64, false, 11>(t, in, frag, out, addend, bn, fin))); }
  });
  return AZ_OK;
}

template <class Gm, int F> static int tr_wgrad16_f(az_trainer* t, const float* a, const float* dg, float* out, hipStream_t st) {
  // 128 filters: the 4-wavefront form (half the input channels per workgroup, one board per LDS chunk, 44 KB): the same sums in
  // the same order as the 8-wavefront form, and a workgroup of it fits on a CU beside one of k_conv16_layer
  constexpr int CS = (F == 128 && Gm::P <= 48) ? 2 : 1, RPC = CS == 2 ? 48 : 128;
  using G = WG16<F, CS, RPC>;
  static bool attr_done = false;
  if (!attr_done) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad16<Gm, F, 0, CS, RPC>), hipFuncAttributeMaxDynamicSharedMemorySize, G::BYTES)); attr_done = true; }
  hipLaunchKernelGGL((k_wgrad16<Gm, F, 0, CS, RPC>), dim3(t->wg_splits, G::TG * CS), dim3(G::THREADS), G::BYTES, st, a, dg, t->wg_part, t->B, t->wg_splits, (long long*)nullptr);
  const long long n = 9LL * F * F;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(tr_grid(n / 2)), dim3(256), 0, st, t->wg_part, t->wg_splits, n, out);
  return AZ_OK;
}
// weight gradient of a 3x3 F -> F convolution on the MFMA kernel: out = [9][F][F] in the Wm layout
static int tr_wgrad16(az_trainer* t, const float* a, const float* dg, float* out, hipStream_t st) {
  DISPATCH_GAME(t->game, { if (t->F == 128) AZCHK((tr_wgrad16_f<Gm, 128>(t, a, dg, out, st))); else AZCHK((tr_wgrad16_f<Gm, 64>(t, a, dg, out, st))); });
  return AZ_OK;
}

static void tr_bn_bwd(az_trainer* t, const float* da, const float* a, const float* g, const float* mean, const float* invstd, const float* gamma,
                      long long n, int C, float* dg, float* dy_out) {
  if (C % 4 == 0) hipLaunchKernelGGL((k_tr_bn_bwd<4>), dim3(tr_grid(n / 2)), dim3(256), 0, t->stream, da, a, g, mean, invstd, gamma, t->bn_mf, n, C, dg, dy_out);
  else hipLaunchKernelGGL((k_tr_bn_bwd<1>), dim3(tr_grid(n)), dim3(256), 0, t->stream, da, a, g, mean, invstd, gamma, t->bn_mf, n, C, dg, dy_out);
}
// column sums of mode MODE over R rows, result in t->sums
template <int MODE>
static int tr_colsum(az_trainer* t, const float* x, const float* out_act, const float* g, const float* mean, const float* invstd, long long R, int C,
                     TrFinal fin = TrFinal{}) {
  const int nchunks = (int)((R + TR_CHUNK - 1) / TR_CHUNK);
  // (tr_finish -- the second stage in the producer's last workgroup -- is off: measured slower, see k_tr_colsum_final)
  const bool inside = t->fin_inside && C <= 128;
  fin.counter = inside ? t->fin_counter : nullptr;
  if (MODE == 1 && C % 4 == 0 && C <= 128 && 256 % (C / 4) == 0)
    hipLaunchKernelGGL(k_tr_colsum1v, dim3(nchunks), dim3(256), 0, t->stream, x, out_act, g, mean, invstd, R, C, t->part, fin);
  else hipLaunchKernelGGL((k_tr_colsum<MODE>), dim3(nchunks, (C + 63) / 64), dim3(256), 0, t->stream, x, out_act, g, mean, invstd, R, C, t->part, fin);
  if (!inside) hipLaunchKernelGGL(k_tr_colsum_final, dim3(C), dim3(64), 0, t->stream, t->part, nchunks, C, t->sums, fin);
  return AZ_OK;
}

// forward (train mode) + loss + backward for the batch idx[0..B); the gradient of the DATA terms is left in t->gblob
// (blob layout, L2 term not included); the step's five sums (w, kl, mse, inv, sum of squares) go to d_sums (device).
// Nothing here waits for the GPU: idx_host must stay alive until the stream has consumed it.
static int tr_forward_backward(az_trainer* t, const int* idx_host, double* d_sums) {
  const GameInfo& gi = t->gi;
  const int F = t->F, C = gi.C, P = gi.P, A = gi.A, npf = t->npf, nvf = t->nvf, B = t->B;
  const long long R = t->R;
  hipStream_t st = t->stream;
  az_dataset* d = t->d;
  const int ntower = 1 + 2 * t->nblocks;
  TrConv& hp = t->convs[ntower];
  TrConv& hv = t->convs[ntower + 1];
  float* blob = t->blob;
  HIPCHK(hipMemcpyAsync(t->d_idx, idx_host, sizeof(int) * B, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_tr_batch, dim3(B), dim3(64), 0, st, t->d_idx, B, C * P, A, d->d_W, d->d_X, d->d_A, d->d_P, d->d_V, t->bW, t->bX, t->bA, t->bP, t->bV);
  hipLaunchKernelGGL(k_tr_gather, dim3(tr_grid((long long)t->nwork)), dim3(256), 0, st, blob, t->map, (long long)t->nwork, t->work);
  HIPCHK(hipMemsetAsync(t->gblob, 0, sizeof(float) * t->nparams, st));
  // ---------------- forward ----------------
  // (r4) one launch per tower layer: the convolution of layer l computes its INPUT a(l-1) = relu(bn(g(l-1)) (+ skip)) while the rows
  // go into LDS (BnIn, and writes it for the backward pass), leaves the column sums of its output and its last workgroup turns them
  // into the layer's batch statistics (tr_finish).  Were three launches (convolution, k_tr_colsum_final, k_tr_bn_apply).
  auto bn_of = [&](int l) {                                          // what turns g(l) into a(l)
    TrConv& c = t->convs[l];
    const bool second = l > 0 && (l % 2) == 0;                       // conv2 of a block: skip connection from the block input
    return BnIn{c.g, c.mean, c.invstd, blob + c.off_bn, blob + c.off_bn + c.cout, second ? t->convs[l - 2].a : nullptr, c.a};
  };
  for (int l = 0; l < ntower; ++l) {
    TrConv& c = t->convs[l];
    TrFinal fin{}; fin.mode = 1; fin.R = R; fin.momentum = t->cfg.batch_norm_momentum; fin.bias = blob + c.off_b; fin.mean = c.mean; fin.invstd = c.invstd;
    fin.run_mean = blob + c.off_bn + 2 * c.cout; fin.run_var = blob + c.off_bn + 3 * c.cout;
    if (c.mfma) {
      int nparts = 0;
      fin.counter = t->fin_inside ? t->fin_counter : nullptr;
      AZCHK(tr_conv16(t, nullptr, t->work + c.wk_ffwd, c.g, true, &nparts, nullptr, bn_of(l - 1), fin));
      if (!fin.counter) hipLaunchKernelGGL(k_tr_colsum_final, dim3(c.cout), dim3(64), 0, st, t->part, nparts, c.cout, t->sums, fin);
    } else {                                                          // the stem: K = 9 C, im2col + GEMM
      hipLaunchKernelGGL((k_tr_im2col<true>), dim3(tr_grid(R * 9 * c.cin)), dim3(256), 0, st, t->bX, R, c.cin, gi.W, gi.H, c.col);
      AZCHK(tr_gemm(t, false, false, (int)R, c.cout, 9 * c.cin, 1.f, c.col, 9 * c.cin, t->work + c.wk_wm, c.cout, 0.f, c.g, c.cout));
      AZCHK(tr_colsum<0>(t, c.g, nullptr, nullptr, nullptr, nullptr, R, c.cout, fin));
    }
  }
  { // the trunk: the last layer's activation has no convolution after it to ride in
    const BnIn b = bn_of(ntower - 1);
    TrConv& c = t->convs[ntower - 1];
    hipLaunchKernelGGL(k_tr_bn_apply, dim3(tr_grid(R * c.cout)), dim3(256), 0, st, b.g, b.mean, b.invstd, b.gamma, b.beta, b.res, R * c.cout, c.cout, c.a);
  }
  const float* trunk = t->convs[ntower - 1].a;
  for (TrConv* c : {&hp, &hv}) {
    AZCHK(tr_gemm(t, false, false, (int)R, c->cout, F, 1.f, trunk, F, t->work + c->wk_wm, c->cout, 0.f, c->g, c->cout));
    { TrFinal fin{}; fin.mode = 1; fin.R = R; fin.momentum = t->cfg.batch_norm_momentum; fin.bias = blob + c->off_b; fin.mean = c->mean; fin.invstd = c->invstd;
      fin.run_mean = blob + c->off_bn + 2 * c->cout; fin.run_var = blob + c->off_bn + 3 * c->cout;
      AZCHK(tr_colsum<0>(t, c->g, nullptr, nullptr, nullptr, nullptr, R, c->cout, fin)); }
    hipLaunchKernelGGL(k_tr_bn_apply, dim3(tr_grid(R * c->cout)), dim3(256), 0, st, c->g, c->mean, c->invstd, blob + c->off_bn, blob + c->off_bn + c->cout, (const float*)nullptr, R * c->cout, c->cout, c->a);
  }
  // dense heads: rows of hp.a / hv.a of one board are contiguous: [B][P*nf] with k = p*nf + f
  AZCHK(tr_gemm(t, false, false, B, A, P * npf, 1.f, hp.a, P * npf, t->work + t->wk_pd, A, 0.f, t->logits, A));
  hipLaunchKernelGGL(k_tr_bias_act, dim3(tr_grid((long long)B * A)), dim3(256), 0, st, t->logits, blob + t->off_pd_b, (long long)B * A, A, 0);
  AZCHK(tr_gemm(t, false, false, B, F, P * nvf, 1.f, hv.a, P * nvf, t->work + t->wk_v1, F, 0.f, t->v1, F));
  hipLaunchKernelGGL(k_tr_bias_act, dim3(tr_grid((long long)B * F)), dim3(256), 0, st, t->v1, blob + t->off_v1_b, (long long)B * F, F, 1);
  AZCHK(tr_gemm(t, false, false, B, 1, F, 1.f, t->v1, F, t->work + t->wk_v2, 1, 0.f, t->tpre, 1));
  hipLaunchKernelGGL(k_tr_bias_act, dim3(tr_grid(B)), dim3(256), 0, st, t->tpre, blob + t->off_v2_b, (long long)B, 1, 0);
  // ---------------- loss ----------------
  // mean(W)/Wmean / sum(W) = 1 / (B Wmean): the gradient scale needs no reduction first (learning.jl:88)
  double *tw = t->terms, *tkl = t->terms + B, *tmse = t->terms + 2 * (size_t)B, *tinv = t->terms + 3 * (size_t)B;
  hipLaunchKernelGGL(k_tr_loss, dim3(tr_grid(B)), dim3(256), 0, st, t->logits, t->tpre, t->bA, t->bP, t->bV, t->bW, B, A,
                     (float)t->cfg.nonvalidity_penalty, (float)t->cfg.rewards_renormalization, 1.0f / ((float)B * d->Wmean), t->dlogits, t->dt, tw, tkl, tmse, tinv);
  hipLaunchKernelGGL(k_tr_sumsq, dim3(1024), dim3(256), 0, st, blob, t->trainable, (long long)t->nparams, t->bsums + 8);
  hipLaunchKernelGGL(k_tr_step_sums, dim3(1), dim3(256), 0, st, tw, tkl, tmse, tinv, B, t->bsums + 8, 1024, d_sums);
  // ---------------- backward: dense heads ----------------
  float* gw = t->gwork;
  float* gb = t->gblob;
  // policy: dWpd = hp_flat^T dlogits ; dbp = colsum(dlogits) ; dhp = dlogits Wpd^T
  AZCHK(tr_gemm(t, true, false, P * npf, A, B, 1.f, hp.a, P * npf, t->dlogits, A, 0.f, gw + t->wk_pd, A));
  { TrFinal fin{}; fin.mode = 3; fin.out0 = gb + t->off_pd_b; AZCHK(tr_colsum<2>(t, t->dlogits, nullptr, nullptr, nullptr, nullptr, B, A, fin)); }
  float* dhp = t->dact;                                            // [R][npf]
  AZCHK(tr_gemm(t, false, true, B, P * npf, A, 1.f, t->dlogits, A, t->work + t->wk_pd, A, 0.f, dhp, P * npf));
  // value: dt -> dv1 = (dt wv2^T) .* (v1 > 0) ; dwv2 = v1^T dt ; db2 = sum dt ; dWv1 = hv_flat^T dv1 ; db1 ; dhv = dv1 Wv1^T
  AZCHK(tr_gemm(t, true, false, F, 1, B, 1.f, t->v1, F, t->dt, 1, 0.f, gw + t->wk_v2, 1));
  { TrFinal fin{}; fin.mode = 3; fin.out0 = gb + t->off_v2_b; AZCHK(tr_colsum<2>(t, t->dt, nullptr, nullptr, nullptr, nullptr, B, 1, fin)); }
  AZCHK(tr_gemm(t, false, true, B, F, 1, 1.f, t->dt, 1, t->work + t->wk_v2, 1, 0.f, t->dv1, F));
  hipLaunchKernelGGL(k_tr_relu_bwd, dim3(tr_grid((long long)B * F)), dim3(256), 0, st, t->dv1, t->v1, (long long)B * F);
  AZCHK(tr_gemm(t, true, false, P * nvf, F, B, 1.f, hv.a, P * nvf, t->dv1, F, 0.f, gw + t->wk_v1, F));
  { TrFinal fin{}; fin.mode = 3; fin.out0 = gb + t->off_v1_b; AZCHK(tr_colsum<2>(t, t->dv1, nullptr, nullptr, nullptr, nullptr, B, F, fin)); }
  float* dhv = t->dact2;                                           // [R][nvf]
  AZCHK(tr_gemm(t, false, true, B, P * nvf, F, 1.f, t->dv1, F, t->work + t->wk_v1, F, 0.f, dhv, P * nvf));
  // head convolutions: BN backward, weight / bias gradients, gradient into the trunk
  float* dtrunk = t->dcol;                                         // [R][F] (front of the big scratch)
  bool first = true;
  for (TrConv* c : {&hp, &hv}) {
    float* dh = c == &hp ? dhp : dhv;
    { TrFinal fin{}; fin.mode = 2; fin.R = R; fin.mf = t->bn_mf; fin.dgamma = gb + c->off_bn; fin.dbeta = gb + c->off_bn + c->cout;
      AZCHK(tr_colsum<1>(t, dh, c->a, c->g, c->mean, c->invstd, R, c->cout, fin)); }
    tr_bn_bwd(t, dh, c->a, c->g, c->mean, c->invstd, blob + c->off_bn, R * c->cout, c->cout, dh, nullptr);
    AZCHK(tr_gemm(t, true, false, F, c->cout, (int)R, 1.f, trunk, F, dh, c->cout, 0.f, gw + c->wk_wm, c->cout));
    // the bias of a convolution that feeds a train-mode BatchNorm has gradient sum(dg) = gamma invstd (sum dy - R m0 - m1 sum xhat)
    // = 0 exactly (the batch mean absorbs it): its slot in gblob stays 0 and only the L2 term moves it
    AZCHK(tr_gemm(t, false, true, (int)R, F, c->cout, 1.f, dh, c->cout, t->work + c->wk_wm, c->cout, first ? 0.f : 1.f, dtrunk, F));
    first = false;
  }
  // ---------------- backward: tower ----------------
  // da = gradient w.r.t. the activation convs[l].a; dskip (block input share) in dact2.  The MFMA data-gradient convolution
  // writes out of place, so the running gradient moves through three buffers (the front of the big scratch, where the head
  // convolutions left the trunk gradient, dact and dact3) instead of being copied.
  // (r3) The weight gradient of a layer is off the chain dg(l) -> da(l-1) -> dg(l-1): it runs on a second stream, so the
  // HBM-bound batch-norm backward passes of layer l-1 share the chip with the MFMA-bound k_wgrad16 of layer l.  Three
  // buffers because the data-gradient convolution of layer l-1 must not overwrite the dg that k_wgrad16 of layer l still
  // reads: it writes the buffer last read by the weight gradient of layer l+1 and waits for that one only.
  float* ring[3] = {dtrunk, t->dact, t->dact3};
  int cur = 0;
  for (int l = ntower - 1; l >= 0; --l) {
    TrConv& c = t->convs[l];
    float* da = ring[cur];
    const bool second = l > 0 && (l % 2) == 0;
    { TrFinal fin{}; fin.mode = 2; fin.R = R; fin.mf = t->bn_mf; fin.dgamma = gb + c.off_bn; fin.dbeta = gb + c.off_bn + c.cout;
      AZCHK(tr_colsum<1>(t, da, c.a, c.g, c.mean, c.invstd, R, c.cout, fin)); }
    // dg overwrites da; for conv2 the masked gradient dy also flows to the block input (dact2)
    tr_bn_bwd(t, da, c.a, c.g, c.mean, c.invstd, blob + c.off_bn, R * c.cout, c.cout, da, second ? t->dact2 : nullptr);
    if (!c.mfma) AZCHK(tr_gemm(t, true, false, 9 * c.cin, c.cout, (int)R, 1.f, c.col, 9 * c.cin, da, c.cout, 0.f, gw + c.wk_wm, c.cout));
    if (c.mfma && !t->wg_late) HIPCHK(hipEventRecord(t->ev_dg[l], st));    // dg(l) is ready: the weight gradient may start beside the data gradient
    if (l > 0) {
      // data gradient da_prev = conv(dg, mirrored taps, ci <-> co): the same MFMA layer kernel with the wk_fdg fragments, out of place
      const bool first_of_block = (l % 2) == 1;                    // conv1: its input is the block input, which also gets the skip share
      const int nxt = (cur + 1) % 3;                               // last read by the weight gradient of layer l + 2
      if (l + 2 < ntower && t->convs[l + 2].mfma) HIPCHK(hipStreamWaitEvent(st, t->ev_wg[l + 2], 0));
      AZCHK(tr_conv16(t, da, t->work + c.wk_fdg, ring[nxt], false, nullptr, first_of_block ? t->dact2 : nullptr));   // (r3) the add rides in the epilogue
      cur = nxt;
    }
    if (c.mfma) {
      // the weight gradient starts when the data gradient of its layer is done: two MFMA kernels side by side only halve each
      // other's share of the chip (measured), while the HBM-bound passes of layer l-1 do fit beside k_wgrad16
      if (t->wg_late) HIPCHK(hipEventRecord(t->ev_dg[l], st));
      if (!t->one_stream) HIPCHK(hipStreamWaitEvent(t->side, t->ev_dg[l], 0));
      AZCHK(tr_wgrad16(t, t->convs[l - 1].a, da, gw + c.wk_wm, t->one_stream ? st : t->side));
      HIPCHK(hipEventRecord(t->ev_wg[l], t->one_stream ? st : t->side));
    }
  }
  for (int l = 1; l < ntower && l <= 2; ++l) if (t->convs[l].mfma) HIPCHK(hipStreamWaitEvent(st, t->ev_wg[l], 0));   // the side stream is in order: its last two launches
  // working-layout weight gradients -> blob layout (the rotated copies carry no gradient of their own: their slots in
  // gwork stay zero and map to the same blob entries, so scatter only the primary ranges)
  hipLaunchKernelGGL(k_tr_scatter, dim3(tr_grid(t->nscat)), dim3(256), 0, st, gw, t->map, t->scat, t->nscat, gb);
  return AZ_OK;
}
// `losses` from the five sums of a step (learning.jl:67-90): L and (Lp, Lv, Lreg, Linv, mean(W)/Wmean)
static void tr_losses(const az_trainer* t, const double* sums, float* L, float* parts) {
  const double sw = sums[0];
  const float scale = (float)(sw / (double)t->B) / t->d->Wmean;
  const float Lp = (float)(-sums[1] / sw) - t->d->Hp;
  const float Lv = (float)(sums[2] / sw);
  const float Lreg = t->cfg.l2_regularization == 0.0 ? 0.f : (float)((double)(float)t->cfg.l2_regularization * sums[4]);
  const float Linv = t->cfg.nonvalidity_penalty == 0.0 ? 0.f : (float)t->cfg.nonvalidity_penalty * (float)(sums[3] / sw);
  *L = scale * (Lp + Lv + Lreg + Linv);
  if (parts) { parts[0] = Lp; parts[1] = Lv; parts[2] = Lreg; parts[3] = Linv; parts[4] = scale; }
}

static void tr_next_batch(az_trainer* t, std::vector<int>& idx) {
  // Flux.DataLoader(shuffle = true, partial = false) cycled (learning.jl:114-119): a fresh permutation per epoch,
  // Fisher-Yates driven by the RNG contract's shuffle stream (seed; epoch)
  const int64_t n = t->d->n;
  idx.resize(t->B);
  if (t->perm.empty() || t->perm_pos + t->B > n) {
    t->perm.resize(n);
    for (int64_t i = 0; i < n; ++i) t->perm[i] = (int)i;
    az_rng r = az_rng_make(t->cfg.seed, (uint32_t)t->epoch, 0, AZ_RNG_SHUFFLE);
    for (int64_t i = n - 1; i > 0; --i) {
      int64_t j = (int64_t)(az_rng_f64(&r) * (double)(i + 1));
      if (j > i) j = i;
      std::swap(t->perm[i], t->perm[j]);
    }
    t->perm_pos = 0; t->epoch++;
  }
  for (int b = 0; b < t->B; ++b) idx[b] = t->perm[t->perm_pos + b];
  t->perm_pos += t->B;
}

extern "C" int az_trainer_create(az_engine* e, az_dataset* d, const az_train_cfg* cfg, az_trainer** out) {
  ENGINE(e);
  if (!d || !cfg || !out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(az_train_cfg)) return fail(AZ_ERR_BAD_ARG, "az_train_cfg size mismatch: call az_train_cfg_init");
  if (e->cfg.oracle != AZ_ORACLE_RESNET || !e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  if (d->game != e->cfg.game || d->device != e->device) return fail(AZ_ERR_BAD_ARG, "data set and engine differ in game or device");
  if (e->cfg.game == AZ_GAME_GO9_PLANES) return fail(AZ_ERR_BAD_ARG, "game id %d is a network-only tensor geometry: no device trainer", e->cfg.game);
  if (cfg->optimiser != AZ_OPT_ADAM && cfg->optimiser != AZ_OPT_CYCLIC_NESTEROV) return fail(AZ_ERR_BAD_ARG, "unknown optimiser %d", cfg->optimiser);
  const int64_t B = std::min<int64_t>(cfg->batch_size, d->n);     // batchsize = min(params.batch_size, length(W)), learning.jl:113
  if (cfg->batch_size < 2 || B < 2) return fail(AZ_ERR_BAD_ARG, "batch_size and the data set must have at least 2 samples (batch statistics)");
  if (!(cfg->rewards_renormalization > 0.0)) return fail(AZ_ERR_BAD_ARG, "rewards_renormalization must be > 0");
  az_trainer* t = new (std::nothrow) az_trainer();
  if (!t) return fail(AZ_ERR_HIP, "out of host memory");
  t->e = e; t->d = d; t->cfg = *cfg; t->game = e->cfg.game; t->device = e->device; t->gi = e->gi; t->stream = nullptr; t->side = nullptr; t->gemm_ws = nullptr; t->gemm_ws_floats = 0;
  t->B = (int)B; t->nblocks = e->cfg.num_blocks; t->F = e->cfg.num_filters; t->npf = e->cfg.num_policy_head_filters; t->nvf = e->cfg.num_value_head_filters;
  t->nA = e->gi.A; t->R = (long long)B * e->gi.P; t->nparams = e->blob.size();
  t->perm_pos = 0; t->epoch = 0; t->step = 0; t->b1t = 1.0f; t->b2t = 1.0f;
  int st = [&]() -> int {
    // the chain of the step on a high-priority stream, the weight gradients beside it on a low-priority one
    int pr_least = 0, pr_greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
    HIPCHK(hipStreamCreateWithPriority(&t->stream, hipStreamDefault, pr_greatest));
    HIPCHK(hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, pr_least));
    { const char* one = getenv("AZHIP_TRAIN_ONE_STREAM"); t->one_stream = one && atoi(one) != 0; }
    { const char* la = getenv("AZHIP_TRAIN_WG_LATE"); t->wg_late = la && atoi(la) != 0; }   // diagnosis: k_wgrad16(l) only after the data gradient of layer l   // diagnosis: the weight gradients in line with everything else
    t->gemm_ws_floats = (size_t)4 << 20;                            // 16 MB: partial tiles of the split weight-gradient reductions
    AZCHK(tr_alloc(t, &t->gemm_ws, t->gemm_ws_floats));
    AZCHK(trainer_build(t));
    return AZ_OK;
  }();
  if (st != AZ_OK) { az_trainer_destroy(t); return st; }
  *out = t;
  return AZ_OK;
}

#define TRAINER(t) if (!(t)) return fail(AZ_ERR_BAD_ARG, "trainer is NULL"); HIPCHK(hipSetDevice((t)->device))

// loss and data gradient (blob layout, WITHOUT the L2 term) of one given batch; no update.  parts = Lp, Lv, Lreg, Linv, scale.
extern "C" int az_trainer_gradients(az_trainer* t, const int32_t* sample_idx, float* loss, float* parts, float* grad, int64_t n) {
  TRAINER(t);
  if (!sample_idx || !loss) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  for (int b = 0; b < t->B; ++b) if (sample_idx[b] < 0 || sample_idx[b] >= t->d->n) return fail(AZ_ERR_BAD_ARG, "sample index %d out of range", sample_idx[b]);
  if (grad && n != (int64_t)t->nparams) return fail(AZ_ERR_BAD_ARG, "gradient buffer must hold %zu floats", t->nparams);
  // the running statistics must not move in a gradient probe: save / restore them around the pass
  std::vector<float> saved(t->nparams);
  HIPCHK(hipMemcpy(saved.data(), t->blob, sizeof(float) * t->nparams, hipMemcpyDeviceToHost));
  AZCHK(tr_forward_backward(t, sample_idx, t->bsums));
  double sums[5];
  HIPCHK(hipMemcpyAsync(sums, t->bsums, sizeof sums, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  HIPCHK(hipGetLastError());
  tr_losses(t, sums, loss, parts);
  if (grad) HIPCHK(hipMemcpy(grad, t->gblob, sizeof(float) * t->nparams, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(t->blob, saved.data(), sizeof(float) * t->nparams, hipMemcpyHostToDevice));
  return AZ_OK;
}

// debug aid (not part of the ABI in azhip.h): the post-ReLU activation the last forward pass left for layer `which` -- tower
// layers 0 .. 2 num_blocks ([R][F], R = batch x positions), then the policy / value head convolutions ([R][npf], [R][nvf]), then
// the value head's first dense layer ([batch][F]).  tests/test_train_gpu.py reads the ReLU masks (a > 0) off them: an fp32 chain
// and an fp64 reference disagree on activations within rounding of zero, and ONE flipped unit moves single weight-gradient
// entries by more than the tolerance the rest of the gradient meets.
extern "C" int az_debug_trainer_activation(az_trainer* t, int32_t which, float* out, int64_t n) {
  TRAINER(t);
  const int ntower = 1 + 2 * t->nblocks;
  const float* src = nullptr;
  int64_t want = 0;
  if (which >= 0 && which < ntower + 2) { src = t->convs[which].a; want = (int64_t)t->R * t->convs[which].cout; }
  else if (which == ntower + 2) { src = t->v1; want = (int64_t)t->B * t->F; }
  else return fail(AZ_ERR_BAD_ARG, "layer %d of %d", which, ntower + 3);
  if (!out || n != want) return fail(AZ_ERR_BAD_ARG, "buffer must hold %lld floats", (long long)want);
  HIPCHK(hipStreamSynchronize(t->stream));
  HIPCHK(hipMemcpy(out, src, sizeof(float) * (size_t)want, hipMemcpyDeviceToHost));
  return AZ_OK;
}

// batch_updates!(tr, n) (learning.jl:131-141): n optimiser steps on successive batches; losses[i] = L before update i
extern "C" int az_trainer_batch_updates(az_trainer* t, int32_t n, float* losses) {
  TRAINER(t);
  if (n < 0) return fail(AZ_ERR_BAD_ARG, "n must be >= 0");
  if (n == 0) return AZ_OK;
  // Network.train! builds a fresh optimiser state on every call (Flux.setup inside train!, flux.jl:69,83): the Adam
  // moments / Nesterov velocities do not survive from one batch_updates! (checkpoint) to the next
  HIPCHK(hipMemsetAsync(t->opt_m, 0, sizeof(float) * t->nparams, t->stream));
  HIPCHK(hipMemsetAsync(t->opt_v, 0, sizeof(float) * t->nparams, t->stream));
  t->b1t = 1.0f; t->b2t = 1.0f;
  // the n steps are enqueued back to back: no host round trip inside the loop (the step sums of every step stay on the
  // device until the end; the batch indices of all steps stay alive in `idx`)
  std::vector<std::vector<int>> idx((size_t)n);
  double* d_sums = nullptr;
  HIPCHK(hipMalloc((void**)&d_sums, sizeof(double) * 5 * (size_t)n));
  const float reg_c = (float)(2.0 * (double)(float)t->cfg.l2_regularization / ((double)t->B * (double)t->d->Wmean));   // d/dw of scale * creg * sum(w^2) = (sum W) * reg_c * w
  int rc = AZ_OK;
  for (int i = 0; i < n && rc == AZ_OK; ++i) {
    tr_next_batch(t, idx[i]);
    double* ds = d_sums + 5 * (size_t)i;
    rc = tr_forward_backward(t, idx[i].data(), ds);
    if (rc != AZ_OK) break;
    if (t->cfg.optimiser == AZ_OPT_ADAM) {
      t->b1t *= 0.9f; t->b2t *= 0.999f;
      hipLaunchKernelGGL(k_tr_adam, dim3(tr_grid((long long)t->nparams)), dim3(256), 0, t->stream, t->blob, t->gblob, t->opt_m, t->opt_v, t->trainable,
                         (long long)t->nparams, t->cfg.lr, t->b1t, t->b2t, ds, reg_c);
    } else {
      // lr = CyclicSchedule(lr_base, lr_high, lr_low; n), momentum = CyclicSchedule(momentum_high, momentum_low,
      // momentum_high; n) (flux.jl:78-94); CyclicSchedule = PLSchedule([1, floor(.45 n), floor(.9 n), n], [base, mid, base,
      // term]) (schedule.jl:130-134).  Flux.adjust! follows update!, so step i (1-based) runs with the values of index i-1
      // and the first step with Nesterov(lr_low, momentum_high).
      auto cyc = [&](double base, double mid, double term, int k) -> float {
        const int xs[4] = {1, (int)(0.45 * n), (int)(0.90 * n), n};
        const double ys[4] = {base, mid, base, term};
        int pt = -1;
        for (int q = 0; q < 4; ++q) if (xs[q] <= k) pt = q;
        if (pt < 0) return (float)ys[0];
        if (pt == 3) return (float)ys[3];
        if (xs[pt + 1] == xs[pt]) return (float)ys[pt];
        return (float)(ys[pt] + (ys[pt + 1] - ys[pt]) / (double)(xs[pt + 1] - xs[pt]) * (double)(k - xs[pt]));
      };
      const float lr = i == 0 ? t->cfg.lr_low : cyc(t->cfg.lr_base, t->cfg.lr_high, t->cfg.lr_low, i);
      const float rho = i == 0 ? t->cfg.momentum_high : cyc(t->cfg.momentum_high, t->cfg.momentum_low, t->cfg.momentum_high, i);
      hipLaunchKernelGGL(k_tr_nesterov, dim3(tr_grid((long long)t->nparams)), dim3(256), 0, t->stream, t->blob, t->gblob, t->opt_m, t->trainable,
                         (long long)t->nparams, lr, rho, ds, reg_c);
    }
    t->step++;
  }
  std::vector<double> sums(5 * (size_t)n);
  hipError_t e1 = hipMemcpyAsync(sums.data(), d_sums, sizeof(double) * sums.size(), hipMemcpyDeviceToHost, t->stream);
  hipError_t e2 = hipStreamSynchronize(t->stream);
  hipError_t e3 = hipGetLastError();
  (void)hipFree(d_sums);
  AZCHK(rc);
  HIPCHK(e1); HIPCHK(e2); HIPCHK(e3);
  if (losses) for (int i = 0; i < n; ++i) tr_losses(t, &sums[5 * (size_t)i], &losses[i], nullptr);
  return AZ_OK;
}

// get_trained_network (learning.jl:127-129): the current parameters (Flux order, running statistics included)
extern "C" int az_trainer_get_params(az_trainer* t, float* blob, int64_t n) {
  TRAINER(t);
  if (!blob || n != (int64_t)t->nparams) return fail(AZ_ERR_BAD_ARG, "blob must hold %zu floats", t->nparams);
  HIPCHK(hipMemcpy(blob, t->blob, sizeof(float) * t->nparams, hipMemcpyDeviceToHost));
  return AZ_OK;
}
