// gemm.h -- the small fp32 GEMMs of the optimiser step (train.hip): stem (K = 27), 1x1 head convolutions, dense layers,
// in their three passes (forward NN, weight gradient TN with the batch as reduction dimension, data gradient NT).
// Hand-written on v_mfma_f32_32x32x2_f32 (round 1 went through a dlopen'ed rocBLAS for these, 1.4 % of a step plus ~6 s
// of Tensile start-up).
//
//   C[M][N] = alpha * op(A) * op(B) + beta * C,   row-major, op = identity or transpose
//
// One workgroup (4 wavefronts) owns a 64 x 64 tile of C, wavefront (wm, wn) a 32 x 32 quarter = one MFMA accumulator; the
// reduction runs in slices of 32 through LDS (A slice [64][32 + 1], B slice [32][64 + 1]: conflict-free fragment reads, the next
// slice prefetched into registers under the MFMAs,
// global loads coalesced along whichever index is contiguous for the transpose case), out-of-range elements load as 0 so
// any M, N, K works (N = 1: the value head's last layer; K = 27: the stem).  Long reductions with few output tiles (the
// weight gradients: K = batch x positions = 43 k rows for 27 x 64 outputs) are split over grid.z into partial tiles that
// k_gemm_reduce adds in a fixed order: deterministic, no atomics.
#pragma once
#include <hip/hip_runtime.h>

typedef float gemm_f32x16 __attribute__((ext_vector_type(16)));

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) k_gemm_f32(int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                  float* __restrict__ Cout, int ldc, float alpha, float beta, int ksplit,
                                                  float* __restrict__ partial) {
  constexpr int KS = 32;                                            // reduction slice per LDS stage
  __shared__ float As[64][KS + 1];
  __shared__ float Bs[KS][65];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int nsplit = gridDim.z;
  const int kbeg = blockIdx.z * ksplit, kend = (kbeg + ksplit) < K ? (kbeg + ksplit) : K;
  gemm_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float ra[8], rb[8];
  // the slice after the current one travels from memory into registers while the MFMAs of the current one run
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int m, k;
      if (TA) { m = t & 63; k = (t >> 6) + 4 * j; } else { k = t & 31; m = (t >> 5) + 8 * j; }
      const int gm = m0 + m, gk = k0 + k;
      ra[j] = (gm < M && gk < kend) ? (TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk]) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int n, k;
      if (TB) { k = t & 31; n = (t >> 5) + 8 * j; } else { n = t & 63; k = (t >> 6) + 4 * j; }
      const int gn = n0 + n, gk = k0 + k;
      rb[j] = (gn < N && gk < kend) ? (TB ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn]) : 0.0f;
    }
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (TA) As[t & 63][(t >> 6) + 4 * j] = ra[j]; else As[(t >> 5) + 8 * j][t & 31] = ra[j];
      if (TB) Bs[t & 31][(t >> 5) + 8 * j] = rb[j]; else Bs[(t >> 6) + 4 * j][t & 63] = rb[j];
    }
    __syncthreads();
    if (k0 + KS < kend) fetch(k0 + KS);
#pragma unroll
    for (int kk = 0; kk < KS; kk += 2) {
      const float a = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];
      const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row >= M) continue;
    if (nsplit > 1) partial[((size_t)blockIdx.z * M + row) * N + col] = acc[r];
    else {
      float v = alpha * acc[r];
      if (beta != 0.0f) v += beta * Cout[(size_t)row * ldc + col];
      Cout[(size_t)row * ldc + col] = v;
    }
  }
}
// a block = 64 outputs x 4 lanes over the splits (lane l adds splits l, l+4, ... with four running sums: sixteen loads in
// flight per output instead of a chain of nsplit dependent ones), combined in a fixed order
static __global__ void __launch_bounds__(256) k_gemm_reduce(const float* __restrict__ partial, int nsplit, int M, int N, float* __restrict__ Cout, int ldc, float alpha, float beta) {
  __shared__ float sh[4][64];
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + ol;
  const long long mn = (long long)M * N;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < mn) {
    int k = sl;
    for (; k + 12 < nsplit; k += 16)
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += partial[(size_t)(k + 4 * u) * mn + i];
    for (; k < nsplit; k += 4) s[0] += partial[(size_t)k * mn + i];
  }
  sh[sl][ol] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (sl == 0 && i < mn) {
    const float t = (sh[0][ol] + sh[1][ol]) + (sh[2][ol] + sh[3][ol]);
    const int row = (int)(i / N), col = (int)(i % N);
    float v = alpha * t;
    if (beta != 0.0f) v += beta * Cout[(size_t)row * ldc + col];
    Cout[(size_t)row * ldc + col] = v;
  }
}

// ws: workspace of ws_floats floats for the split reductions (may be nullptr: no splitting)
static inline int gemm_f32(hipStream_t st, float* ws, size_t ws_floats, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda,
                           const float* B, int ldb, float beta, float* C, int ldc) {
  if (M <= 0 || N <= 0) return 0;
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  int splits = 1;
  if (ws && tiles < 256 && K >= 1024) {
    splits = (512 + tiles - 1) / tiles;
    const int maxs = (K + 255) / 256;
    if (splits > maxs) splits = maxs;
    while (splits > 1 && (size_t)splits * M * N > ws_floats) --splits;
  }
  int ksplit = (K + splits - 1) / splits;
  ksplit = (ksplit + 31) / 32 * 32;
  splits = (K + ksplit - 1) / ksplit;
  if (splits < 1) splits = 1;
  const dim3 grid((N + 63) / 64, (M + 63) / 64, splits);
  if (!ta && !tb) hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, ksplit, ws);
  else if (ta && !tb) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, ksplit, ws);
  else if (!ta && tb) hipLaunchKernelGGL((k_gemm_f32<false, true>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, ksplit, ws);
  else hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, ksplit, ws);
  if (splits > 1) hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)(((long long)M * N + 63) / 64)), dim3(256), 0, st, ws, splits, M, N, C, ldc, alpha, beta);
  return 0;
}
