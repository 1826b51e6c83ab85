// resnet.h -- the two-headed ResNet oracle (src/networks/architectures/resnet.jl:53-92, test mode): the first
// tower kernel (k_tower, 32x32x2 MFMA; still used for launch sizes it quantises better, see pick_tower in azhip.hip),
// and the dense heads.  The 16x16x4 tower lives in resnet16.h.
//
//  k_tower   one workgroup owns TB = 128/P whole boards (3 Connect-Four boards = 126 of 128 GEMM rows) and runs
//            stem + every residual block + the two 1x1 head convolutions WITHOUT leaving the CU: activations
//            live in two LDS buffers ([129 rows][F + 4 pad] fp32, row 128 = zeros for the padding taps), so the
//            tower's only HBM traffic is 16 B of state in and P*F fp32 of head features out per board.
//            Every convolution (stem included) is an implicit GEMM on v_mfma_f32_32x32x2_f32: wave (rh, nh)
//            owns rows rh*64..+63 x output channels nh*32..+31 (two accumulators sharing one weight-fragment
//            stream); the A operand (activations) is a ds_read_b128 of 4 consecutive channels of the tap-shifted
//            row, the B operand (weights) a coalesced 1 KiB global_load_dwordx4 from a buffer pre-packed in
//            fragment order (L2-resident), requested one tap ahead and threaded through the MFMA stream.
//            BatchNorm (folded to scale/shift), residual add and ReLU are the accumulator epilogue.
//            The MFMA K order is the fp32 contract of include/azhip.h: lanes 0-31 carry channel j,
//            lanes 32-63 channel F/2 + j, so each output is the fma chain the oracle restates.
//  k_heads_mfma / k_heads
//            flatten + Dense layers + softmax / tanh + Network.forward_normalized
//            (src/networks/network.jl:264-271): K = P*nf ascending chain per output, on MFMA (32-board tiles)
//            or, for head widths that are not multiples of 4, one VALU thread per output.
//  Debug aid: NetDev::dbg cycle stamps (az_debug_tower_timeline / az_debug_heads_timeline).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/az_numerics.h"
#include "games.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N - 1>): register arrays indexed by the constant are
// split into registers by the front end (a `#pragma unroll` loop inside a run-time loop leaves them in scratch)
template <int I, int N, class Fn> __device__ __forceinline__ void static_for_impl(Fn& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_impl<I + 1, N>(f); }
}
template <int N, class Fn> __device__ __forceinline__ void static_for(Fn f) { static_for_impl<0, N>(f); }

struct NetDev {
  int nblocks, F, npf, nvf, HF;       // HF = padded npf + nvf (multiple of 32)
  const float* stem_w;                // [F/32][K2][64] MFMA B fragments, K2 = ceil(9C/2): W[k = (lane>>5)*K2 + j][co]
  const float* stem_ss;               // [2][F] scale, shift
  const float4* conv_w;               // [2*nblocks][9][F/32][F/8][64] float4 (fragment order)
  const float* conv_ss;               // [2*nblocks][2][F]
  const float4* head_w;               // [1][HF/32][F/8][64] float4
  const float* head_ss;               // [2][HF]
  const float* pol_w;                 // [P*npf][APAD]  k-major
  const float* pol_b;                 // [APAD]
  const float* val_w;                 // [P*nvf][F]     k-major
  const float* val_b;                 // [F]
  const float* val2_w;                // [F]
  float val2_b;
  unsigned long long* dbg;            // optional [blocks][16] s_memtime stamps (az_debug_tower_timeline)
  const float2* hd_w;                 // MFMA dense heads: [F/32 value tiles + 1 policy tile][K/4][64] float2
  int hd_ok;                          // 1 when npf % 4 == 0 && nvf % 4 == 0 (k_heads_mfma usable)
  const float4* hd16_w;               // k_heads16: [F/16 value tiles + ceil(A/16) policy tiles][K/16][64] float4: W[16j + 4s + (lane>>4)][16 tile + (lane&15)], s = x..w
  int hd16_ok;                        // 1 when npf == nvf == 32
};

static constexpr int TOWER_ROWS = 128;

template <int F> struct TowerLds {
  static constexpr int STRIDE = F + 4;                       // +16 B: conflict-free ds_read_b128 columns
  static constexpr int BUF = (TOWER_ROWS + 1) * STRIDE;      // floats per buffer (row 128 = zeros)
  static constexpr int BYTES = 2 * BUF * 4;
};

// One 3x3 (NTAP = 9) or 1x1 (NTAP = 1) convolution over the workgroup's 128 rows.
// Wave tiling: wave = (rh, nh) owns rows rh*64..rh*64+63 (two 32-row MFMA tiles) x output channels
// nh*32..nh*32+31, i.e. two accumulators that share ONE B fragment stream (8 KiB of weights per tap per
// wave instead of 16: the VMEM issue slots of the weight loads were the largest non-MFMA cost).
// Software pipeline: the B fragments of tap t+1 are requested from L2 while the 64 MFMAs of tap t issue,
// one global_load per 8 MFMAs (sched_group_barrier); each tap is fenced by sched_barrier(0) so hipcc cannot
// sink the prefetch next to its first use; the tap loop is fully unrolled so both register sets are
// statically indexed.
template <int F>
__device__ __forceinline__ void load_b_tap(const float4* __restrict__ wt, float4 (&b)[F / 8]) {
#pragma unroll
  for (int jq = 0; jq < F / 8; ++jq) b[jq] = wt[(size_t)jq * 64];
}
// `af` = channels 0..3 of the two A rows of this tap, requested one tap earlier so the first MFMAs never
// wait for LDS; the other 7 float4 per row are read here and land under the first MFMAs.
template <int F>
__device__ __forceinline__ void mfma_tap(const float4 (&af)[2], const float* __restrict__ arow0,
                                         const float* __restrict__ arow1, const float4 (&b)[F / 8], f32x16 (&acc)[2]) {
  // A is consumed in chunks of 8 float4 per row (32 channels) so that at most 64 VGPRs hold activations
#pragma unroll
  for (int c0 = 0; c0 < F / 8; c0 += 8) {
    float4 a0[8], a1[8];
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
      if (c0 == 0 && jq == 0) { a0[0] = af[0]; a1[0] = af[1]; }
      else { a0[jq] = *(const float4*)(arow0 + (c0 + jq) * 4); a1[jq] = *(const float4*)(arow1 + (c0 + jq) * 4); }
    }
    // consecutive MFMAs alternate between the two accumulators (each still sees its k order unchanged)
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[jq].x, b[c0 + jq].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[jq].x, b[c0 + jq].x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[jq].y, b[c0 + jq].y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[jq].y, b[c0 + jq].y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[jq].z, b[c0 + jq].z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[jq].z, b[c0 + jq].z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[jq].w, b[c0 + jq].w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[jq].w, b[c0 + jq].w, acc[1], 0, 0, 0);
    }
  }
}
template <int F, int NTAP>
__device__ __forceinline__ void conv_mfma(const float* __restrict__ in, float* __restrict__ out,
                                          const float4* __restrict__ wpk, const float* __restrict__ ss, int out_ch,
                                          bool residual, const int (&nbr)[2][9], int rh, int nh, int lane,
                                          unsigned long long* stamp = nullptr) {
  static_assert(F % 32 == 0 && F >= 64, "wave tiling: 2 row halves x F/32 channel groups");
  static_assert(NTAP == 1 || NTAP == 9, "tap count");
  constexpr int STRIDE = TowerLds<F>::STRIDE;
  constexpr int JQ = F / 8, NT = F / 32;
  constexpr size_t TAPW = (size_t)NT * JQ * 64;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const int khalf = (lane >> 5) * (F / 2);
  const float4* wl = wpk + (size_t)nh * JQ * 64 + lane;
  float4 bA[JQ], bB[JQ];
  float4 af0[2], af1[2];
  load_b_tap<F>(wl, bA);
  if (NTAP == 1) {
    af0[0] = *(const float4*)(in + nbr[0][4] + khalf);
    af0[1] = *(const float4*)(in + nbr[1][4] + khalf);
    mfma_tap<F>(af0, in + nbr[0][4] + khalf, in + nbr[1][4] + khalf, bA, acc);
    __builtin_amdgcn_sched_barrier(0);
  } else {
#define AZ_TAP_PIPELINE()                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < JQ; ++i_) {                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
    }
    af0[0] = *(const float4*)(in + nbr[0][0] + khalf);
    af0[1] = *(const float4*)(in + nbr[1][0] + khalf);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
      load_b_tap<F>(wl + (size_t)(t + 1) * TAPW, bB);
      af1[0] = *(const float4*)(in + nbr[0][t + 1] + khalf);
      af1[1] = *(const float4*)(in + nbr[1][t + 1] + khalf);
      mfma_tap<F>(af0, in + nbr[0][t] + khalf, in + nbr[1][t] + khalf, bA, acc);
      AZ_TAP_PIPELINE();
      __builtin_amdgcn_sched_barrier(0);
      load_b_tap<F>(wl + (size_t)(t + 2) * TAPW, bA);
      af0[0] = *(const float4*)(in + nbr[0][t + 2] + khalf);
      af0[1] = *(const float4*)(in + nbr[1][t + 2] + khalf);
      mfma_tap<F>(af1, in + nbr[0][t + 1] + khalf, in + nbr[1][t + 1] + khalf, bB, acc);
      AZ_TAP_PIPELINE();
      __builtin_amdgcn_sched_barrier(0);
    }
#undef AZ_TAP_PIPELINE
    mfma_tap<F>(af0, in + nbr[0][8] + khalf, in + nbr[1][8] + khalf, bA, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (stamp) { asm volatile("s_nop 0" ::: "memory"); stamp[0] = __builtin_readcyclecounter(); }
  // the epilogue + barrier is the only non-MFMA stretch of a layer: run it at raised priority so it is not
  // starved by the co-resident workgroup's MFMA stream
  __builtin_amdgcn_s_setprio(2);
  // epilogue: folded BN, residual, ReLU.  C/D layout of the 32x32 MFMA: col = lane & 31,
  // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  const int col = nh * 32 + (lane & 31);
  const float sc = ss[col], sh = ss[out_ch + col];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rh * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float v = az_fmaf(acc[t][r], sc, sh);
      if (residual) v = v + out[row * STRIDE + col];
      v = v > 0.0f ? v : 0.0f;
      out[row * STRIDE + col] = v;
    }
  }
  if (stamp) stamp[1] = __builtin_readcyclecounter();
}

// FROM_PLANES = false: inputs are the leaf states of the evaluation batch (encode fused);
// FROM_PLANES = true : inputs are Flux-layout planes X[n][C][H][W] (Network.forward seam).
template <int F> struct TowerCfg {
  static constexpr int WAVES = 2 * (F / 32);        // (row half, 32-channel group)
  static constexpr int THREADS = 64 * WAVES;        // 256 for F = 64 (2 workgroups per CU), 512 for F = 128 (1 per CU)
};
template <class Gm, int F, bool FROM_PLANES>
__global__ void __launch_bounds__(TowerCfg<F>::THREADS, 2)
k_tower(NetDev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
        const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat) {
  constexpr int P = Gm::P, W = Gm::W, H = Gm::H, C = Gm::C;
  constexpr int TB = TOWER_ROWS / P;
  constexpr int STRIDE = TowerLds<F>::STRIDE;
  constexpr int BUF = TowerLds<F>::BUF;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufX = lds;
  float* bufT = lds + BUF;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  const int board0 = blockIdx.x * TB;
  if (board0 >= n) return;
  constexpr int NTHR = TowerCfg<F>::THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* dbg = net.dbg ? net.dbg + (size_t)blockIdx.x * 16 : nullptr;
  int dbgi = 0;
#define AZ_STAMP() do { if (dbg && tid == 0) dbg[dbgi++] = __builtin_readcyclecounter(); } while (0)
  AZ_STAMP();

  // ---- stage the input planes in the (still unused) T buffer: [128 rows + zero row][C] -------
  float* planes = bufT;
  for (int i = tid; i < (TOWER_ROWS + 1) * C; i += NTHR) {
    const int row = i / C, c = i % C;
    const int b = row / P, q = row % P;
    float val = 0.0f;
    if (row < TOWER_ROWS && b < TB && board0 + b < n) {
      if (FROM_PLANES) val = X[((size_t)(board0 + b) * C + c) * P + q];
      else val = Gm::plane(leaf_env[eval_slots[board0 + b]], q, c);
    }
    planes[i] = val;
  }
  for (int i = tid; i < STRIDE; i += NTHR) { bufX[TOWER_ROWS * STRIDE + i] = 0.0f; }
  // tap-shifted row of this lane's A row (row 128 = zeros for out-of-board taps)
  const int srt = wave & 3, sng = wave >> 2;     // stem tiling: 32-row tile, pair of 32-channel tiles
  int nrow[9];
  {
    const int row = 32 * srt + (lane & 31);
    const int b = row / P, q = row % P, x = q % W, y = q / W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const bool ok = (b < TB) && (y + dy >= 0) && (y + dy < H) && (x + dx >= 0) && (x + dx < W);
      nrow[t] = ok ? row + dy * W + dx : TOWER_ROWS;
    }
  }
  __syncthreads();

  // ---- stem: Conv(3x3, C=>F) + BN + ReLU (resnet.jl:75-77) on the same MFMA: K = 9C padded to even,
  //      k = t*C + c, lanes 0-31 carry k = j, lanes 32-63 k = K2 + j (the paired order of the contract)
  {
    constexpr int NTS = 2, KK = 9 * C, K2 = (KK + 1) / 2;
    const int h = lane >> 5;
    f32x16 acc[NTS];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bw[NTS][K2];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
      for (int j = 0; j < K2; ++j) bw[t][j] = net.stem_w[(size_t)((sng * 2 + t) * K2 + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < K2; ++j) {
      const int k0 = j, k1 = K2 + j;
      const int a0 = nrow[k0 / C] * C + k0 % C;
      const int a1 = (k1 < KK) ? nrow[(k1 < KK ? k1 : 0) / C] * C + k1 % C : TOWER_ROWS * C;
      const float a = planes[h ? a1 : a0];
#pragma unroll
      for (int t = 0; t < NTS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[t][j], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
      const int col = (sng * 2 + t) * 32 + (lane & 31);
      const float sc = net.stem_ss[col], sh = net.stem_ss[F + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * srt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = az_fmaf(acc[t][r], sc, sh);
        bufX[row * STRIDE + col] = v > 0.0f ? v : 0.0f;
      }
    }
  }
  AZ_STAMP();
  // conv wave tiling: (rh, nh) = (wave >> 1, wave & 1); tap-shifted LDS offsets of this lane's two A rows
  const int rh = wave / (F / 32), nh = wave % (F / 32);
  int nbr[2][9];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int row = rh * 64 + tt * 32 + (lane & 31);
    const int b = row / P, q = row % P, x = q % W, y = q / W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const bool ok = (b < TB) && (y + dy >= 0) && (y + dy < H) && (x + dx >= 0) && (x + dx < W);
      nbr[tt][t] = (ok ? row + dy * W + dx : TOWER_ROWS) * STRIDE;
    }
  }
  __syncthreads();
  // the T buffer's zero row (the planes lived there until now)
  for (int i = tid; i < STRIDE; i += NTHR) bufT[TOWER_ROWS * STRIDE + i] = 0.0f;
  __syncthreads();

  // ---- residual tower (resnet.jl:53-63,78) ---------------------------------------------------
  constexpr size_t LAYER_W = (size_t)9 * (F / 32) * (F / 8) * 64;   // float4 per conv layer
  for (int blk = 0; blk < net.nblocks; ++blk) {
    const float4* w1 = net.conv_w + (size_t)(2 * blk) * LAYER_W;
    const float4* w2 = w1 + LAYER_W;
    if (dbg && tid == 0 && blk == 1) dbg[9] = __builtin_readcyclecounter();
    conv_mfma<F, 9>(bufX, bufT, w1, net.conv_ss + (size_t)(2 * blk) * 2 * F, F, false, nbr, rh, nh, lane, (dbg && tid == 0 && blk == 1) ? dbg + 10 : nullptr);
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    if (dbg && tid == 0 && blk == 1) dbg[12] = __builtin_readcyclecounter();
    conv_mfma<F, 9>(bufT, bufX, w2, net.conv_ss + (size_t)(2 * blk + 1) * 2 * F, F, true, nbr, rh, nh, lane);
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    AZ_STAMP();
  }
  // ---- both 1x1 head convolutions + BN + ReLU as one F => HF GEMM (resnet.jl:80-81,86-87) ----
  // (padded head channels have zero weights, scale 0, shift 0)
  conv_mfma<F, 1>(bufX, bufT, net.head_w, net.head_ss, net.HF, false, nbr, rh, nh, lane);
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  AZ_STAMP();
  // head features -> HBM, [board][P][HF] (the flatten order of the dense contract)
  {
    const int HF = net.HF;
    const int nb = (n - board0) < TB ? (n - board0) : TB;
    const int total = nb * P * (HF / 4);
    for (int i = tid; i < total; i += NTHR) {
      const int row = i / (HF / 4), c4 = i % (HF / 4);
      const float4 v4 = *(const float4*)(bufT + row * STRIDE + c4 * 4);
      *(float4*)(hfeat + ((size_t)board0 * P + row) * HF + c4 * 4) = v4;
    }
  }
  AZ_STAMP();
  if (dbg && tid == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); dbg[14] = xcc; dbg[15] = hw; }
#undef AZ_STAMP
}

// Last step of the heads for ONE board by one lane: softmax over the logits and forward_normalized (network.jl:264-271)
// -- heads_finish_policy --, Dense(F => 1, tanh) over the value-hidden units -- heads_finish_value.
// `logit`: the board's A logits (bias added), `vh`: its F hidden units.
template <class Gm>
__device__ __forceinline__ void heads_finish_policy(const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
                                                    const float* __restrict__ Amask, float* __restrict__ Pout, float* __restrict__ Pinv,
                                                    int pstride, int e, const float* __restrict__ logit) {
  constexpr int A = Gm::A;
  float pr[A];
  float mx = logit[0];
  for (int a = 1; a < A; ++a) mx = logit[a] > mx ? logit[a] : mx;
  float s = 0.0f;
  for (int a = 0; a < A; ++a) { pr[a] = az_expf(logit[a] - mx); s += pr[a]; }
  for (int a = 0; a < A; ++a) pr[a] = pr[a] / s;                   // softmax, resnet.jl:84
  float sp = 0.0f;
  for (int a = 0; a < A; ++a) {
    float mk;
    if (Amask) mk = Amask[(size_t)e * A + a];
    else mk = (float)((Gm::mask(leaf_env[eval_slots[e]]) >> a) & 1);
    pr[a] = pr[a] * mk;
    sp += pr[a];
  }
  for (int a = 0; a < A; ++a) Pout[(size_t)e * pstride + a] = pr[a] / (sp + 1.1920929e-7f);
  for (int a = A; a < pstride; ++a) Pout[(size_t)e * pstride + a] = 0.0f;
  if (Pinv) Pinv[e] = 1.0f - sp;
}
template <int F>
__device__ __forceinline__ void heads_finish_value(const NetDev& net, float* __restrict__ Vout, int e, const float* __restrict__ vh) {
  float av = 0.0f;
  for (int k = 0; k < F; ++k) av = az_fmaf(vh[k], net.val2_w[k], av);
  av = av + net.val2_b;
  Vout[e] = az_tanhf(av);                                          // Dense(F => 1, tanh), resnet.jl:90
}
template <class Gm, int F>
__device__ __forceinline__ void heads_finish(const NetDev& net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
                                             const float* __restrict__ Amask, float* __restrict__ Pout, float* __restrict__ Vout,
                                             float* __restrict__ Pinv, int pstride, int e, const float* __restrict__ logit, const float* __restrict__ vh) {
  heads_finish_policy<Gm>(leaf_env, eval_slots, Amask, Pout, Pinv, pstride, e, logit);
  heads_finish_value<F>(net, Vout, e, vh);
}

// Dense heads, softmax, tanh, forward_normalized.  HB boards per 4(F+16)-thread workgroup; thread
// (b, o): o < F -> value hidden unit, F <= o < F + A -> policy logit.
template <class Gm> constexpr int HEADS_LP = (Gm::A + 15) / 16 * 16;          // logit slots per board (16 for the device games, 96 for 82 actions)
template <class Gm> constexpr int HEADS_NPT = (Gm::A + 31) / 32;              // 32-column policy tiles of k_heads_mfma
template <class Gm, int F>
__global__ void __launch_bounds__(4 * (F + HEADS_LP<Gm>))
k_heads(NetDev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
        const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ Amask,
        const float* __restrict__ hfeat, float* __restrict__ Pout, float* __restrict__ Vout,
        float* __restrict__ Pinv, int pstride) {
  constexpr int P = Gm::P, A = Gm::A, HB = 4, PER = F + HEADS_LP<Gm>;   // F value-hidden units + the logits of a board
  static_assert(F + A <= PER, "thread map");
  __shared__ float s_logit[HB][HEADS_LP<Gm>];
  __shared__ float s_vh[HB][F];
  const int n = n_eval_ptr ? *n_eval_ptr : n_fixed;
  const int board0 = blockIdx.x * HB;
  if (board0 >= n) return;
  const int b = threadIdx.x / PER, o = threadIdx.x % PER;
  const int e = board0 + b;
  const int HF = net.HF;
  if (e < n && o < F + A) {
    const float* hf = hfeat + (size_t)e * P * HF;
    float acc = 0.0f;
    if (o < F) {                                   // Dense(P*nvf => F, relu), resnet.jl:89
      const int nvf = net.nvf;
      const float* w = net.val_w + o;
      for (int q = 0; q < P; ++q) {
        const float* h = hf + q * HF + net.npf;
        for (int f = 0; f < nvf; ++f) acc = az_fmaf(h[f], w[(size_t)(q * nvf + f) * F], acc);
      }
      acc = acc + net.val_b[o];
      s_vh[b][o] = acc > 0.0f ? acc : 0.0f;
    } else {                                       // Dense(P*npf => A), resnet.jl:83
      const int a = o - F, npf = net.npf;
      const float* w = net.pol_w + a;
      for (int q = 0; q < P; ++q) {
        const float* h = hf + q * HF;
        for (int f = 0; f < npf; ++f) acc = az_fmaf(h[f], w[(size_t)(q * npf + f) * Gm::APAD], acc);
      }
      s_logit[b][a] = acc + net.pol_b[a];
    }
  }
  __syncthreads();
  if (o == 0 && e < n) heads_finish<Gm, F>(net, leaf_env, eval_slots, Amask, Pout, Vout, Pinv, pstride, e, s_logit[b], s_vh[b]);
}

// Dense heads on MFMA.  One 32-board tile per call; wavefront w < F/32 computes value-hidden
// outputs 32w..32w+31, the next ceil(A / 32) wavefronts the (padded) policy logits (further wavefronts only take part in the
// barrier).  The A operand is the board's head-feature row (K = P*nf values, read as float4 = k 4i..4i+3), the
// B operand the dense matrix pre-packed per MFMA as (W[4i+h][o], W[4i+2+h][o]) with h = lane >> 5:
// v_mfma_f32_32x32x2_f32 consumes k = 2j (lanes 0-31) then 2j+1 (lanes 32-63), i.e. the ascending-k chain of the
// contract.  Called by every thread of a workgroup (k_heads_mfma; the last tower workgroup of a tile group,
// resnet16.h); s_vh [32][F + 1] and s_logit [32][16] are LDS scratch.
template <class Gm, int F>
__device__ __forceinline__ void heads_mfma_tile(const NetDev& net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
                                                int n, const float* __restrict__ Amask, const float* __restrict__ hfeat,
                                                float* __restrict__ Pout, float* __restrict__ Vout, float* __restrict__ Pinv, int pstride,
                                                int board0, float* __restrict__ s_vh, float* __restrict__ s_logit) {
  constexpr int P = Gm::P, A = Gm::A, NVT = F / 32, SV = F + 1, NPT = HEADS_NPT<Gm>, LP = HEADS_LP<Gm>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int HF = net.HF;
  const bool is_pol = wave >= NVT;
  const int pt = wave - NVT;                       // policy tile: logits 32 pt .. 32 pt + 31
  if (wave < NVT + NPT) {
  const int nf = is_pol ? net.npf : net.nvf, foff = is_pol ? 0 : net.npf;
  int e = board0 + (lane & 31);
  if (e >= n) e = n - 1;                         // clamp: rows past the batch are computed and dropped
  const float* hf = hfeat + (size_t)e * P * HF + foff;
  // fragment pointer: value tiles first (each P*nvf/4 steps), then the policy tile
  const float2* wp = net.hd_w + ((size_t)(is_pol ? NVT * (P * net.nvf / 4) + pt * (P * net.npf / 4) : wave * (P * net.nvf / 4))) * 64 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int q4 = nf / 4;
  if (q4 == 8) {
    // 32 head filters: one board position per step (8 float4 of A, 8 float2 of B, 16 MFMAs = 1024 cycles).  The A rows
    // come from HBM / L2 (the tower wrote them just before), so a step's loads are issued DEPTH steps ahead of its
    // MFMAs: AZ_HEADS_DEPTH register stages (2: measured equal to 3 with fewer VGPRs) rotate through the fully unrolled position loop (the wave was latency-bound at
    // one memory round trip per position: 73 us per workgroup against 18 us of MFMA work).
#ifndef AZ_HEADS_DEPTH
#define AZ_HEADS_DEPTH 2
#endif
    constexpr int DEPTH = AZ_HEADS_DEPTH;
    float4 a4[DEPTH][8];
    float2 b2[DEPTH][8];
    auto load_step = [&](int q, int st) {
      const float* hq = hf + q * HF;
#pragma unroll
      for (int f4 = 0; f4 < 8; ++f4) { a4[st][f4] = *(const float4*)(hq + f4 * 4); b2[st][f4] = wp[(size_t)(q * 8 + f4) * 64]; }
    };
#pragma unroll
    for (int q = 0; q < DEPTH - 1; ++q) if (q < P) load_step(q, q);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      if (q + DEPTH - 1 < P) load_step(q + DEPTH - 1, (q + DEPTH - 1) % DEPTH);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f4 = 0; f4 < 8; ++f4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4[q % DEPTH][f4].y : a4[q % DEPTH][f4].x, b2[q % DEPTH][f4].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4[q % DEPTH][f4].w : a4[q % DEPTH][f4].z, b2[q % DEPTH][f4].y, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int q = 0; q < P; ++q) {
      const float* hq = hf + q * HF;
      for (int f4 = 0; f4 < q4; ++f4) {
        const float4 a4 = *(const float4*)(hq + f4 * 4);
        const float2 b2 = wp[(size_t)(q * q4 + f4) * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4.y : a4.x, b2.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4.w : a4.z, b2.y, acc, 0, 0, 0);
      }
    }
  }
  const int col = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (is_pol) { if (pt * 32 + col < A) s_logit[row * LP + pt * 32 + col] = acc[r] + net.pol_b[pt * 32 + col]; }
    else {
      const int o = wave * 32 + col;
      const float v = acc[r] + net.val_b[o];
      s_vh[row * SV + o] = v > 0.0f ? v : 0.0f;
    }
  }
  }
  __syncthreads();
  const int b = threadIdx.x;
  if (b < 32 && board0 + b < n) heads_finish<Gm, F>(net, leaf_env, eval_slots, Amask, Pout, Vout, Pinv, pstride, board0 + b, s_logit + b * LP, s_vh + b * SV);
}
template <class Gm, int F>
__global__ void __launch_bounds__(64 * (F / 32 + HEADS_NPT<Gm>))
k_heads_mfma(NetDev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
             const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ Amask,
             const float* __restrict__ hfeat, float* __restrict__ Pout, float* __restrict__ Vout,
             float* __restrict__ Pinv, int pstride) {
  __builtin_amdgcn_s_setprio(3);   // few workgroups on the critical path of the group's next wave
  __shared__ float s_vh[32 * (F + 1)];
  __shared__ float s_logit[32 * HEADS_LP<Gm>];
  const int n = n_eval_ptr ? *n_eval_ptr : n_fixed;
  const int board0 = blockIdx.x * 32;
  if (board0 >= n) return;
  heads_mfma_tile<Gm, F>(net, leaf_env, eval_slots, n, Amask, hfeat, Pout, Vout, Pinv, pstride, board0, s_vh, s_logit);
}

// Dense heads in 16-board tiles on v_mfma_f32_16x16x4_f32, one wavefront per 16 outputs (F/16 value tiles, ceil(A/16)
// policy tiles): a board tile's K = 32 P chain is 8 P MFMAs of 32 cycles instead of 16 P of 64 (k_heads_mfma's 32-board
// tiles: 75 us for 128 boards at F = 128, one workgroup's latency) and more CUs take part.  32 head filters only
// (NetDev::hd16_ok).  Same fp32 chain as k_heads / k_heads_mfma: bit-identical outputs.
//
// The MFMA's k slot is the lane group g = lane >> 4 and the contract wants ascending k, so the lane (board m, g) needs
// features 4 s + g (s = 0..3) of every 16-feature block, while a coalesced float4 load gives it 4 g .. 4 g + 3: a 4 x 4
// transposition through LDS.  The kernel is a matrix-vector product at heart -- every weight is used for 16 boards --
// and what binds is the CU's vector-memory path, not the MFMA chain (a dependent chain issues every 32.5 cycles,
// tools/probes/mfma_latency.hip; tools/heads_timeline.py showed 400-700 cycles per 4-MFMA block when every wavefront
// fetched and transposed the board features for itself).  Two forms:
//   SPLIT (launches of up to num_cu workgroups): the value tiles and the policy tiles of a board tile are TWO workgroups
//     (blockIdx & 1; softmax / mask and the value's Dense(F => 1) need nothing from each other), so a SIMD carries two
//     chains instead of three; and, when they fit (P <= 48), phase 1 brings the workgroup's features into LDS ONCE,
//     already transposed ([block][board][g][s], the g slot rotated by 2 for boards 4..7 and 12..15: the b128 reads of
//     phase 2 are conflict-free), phase 2 = per block one ds_read_b128 + one float4 of weights (HEADS16_WDEPTH blocks
//     ahead: a small launch has one workgroup per XCD, so every weight comes from HBM).
//   unsplit (full launches, 9 x 9 boards): all tiles in one workgroup, policy wavefronts first (the oldest wavefronts of
//     a SIMD win its issue slots), every wavefront transposes its own features through a private patch (row stride 20
//     floats: the b128 writes and the b32 reads are both conflict-free), triple-buffered, two blocks ahead of the MFMAs.
// (Four stride-4 dword loads straight from global memory instead of the LDS hop were tried: 94 us instead of 34.)
#ifndef HEADS16_DEPTH
#define HEADS16_DEPTH 8      // blocks of global loads in flight, wavefronts that transpose for themselves
#endif
#ifndef HEADS16_WDEPTH
#define HEADS16_WDEPTH 16    // weight fragments in flight, staged form
#endif
template <class Gm> constexpr int HEADS16_NPT = (Gm::A + 15) / 16;
template <class Gm, int F, bool SPLIT> struct H16 {
  static constexpr int P = Gm::P, NVT = F / 16, NPT = HEADS16_NPT<Gm>;
  static constexpr int WAVES = SPLIT ? (NVT > NPT ? NVT : NPT) : NVT + NPT, THREADS = 64 * WAVES;
  static constexpr int NB = 2 * P, RS = 20, SV = F + 1, LP = HEADS_LP<Gm>;
  static constexpr bool STAGE = SPLIT && 16 * P * 32 * 4 <= 96 * 1024;   // a workgroup's features fit in LDS
  static constexpr int A_FLOATS = STAGE ? NB * 256 : 0, PATCH_FLOATS = STAGE ? 0 : WAVES * 3 * 16 * RS;
  static constexpr int BYTES = (A_FLOATS + PATCH_FLOATS + 16 * SV + 16 * LP) * 4;
};
template <class Gm, int F, bool SPLIT> struct H16Threads { static constexpr int V = H16<Gm, F, SPLIT>::THREADS; };
template <class Gm, int F, bool SPLIT>
__global__ void __launch_bounds__((H16Threads<Gm, F, SPLIT>::V))
k_heads16(NetDev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
          const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ Amask,
          const float* __restrict__ hfeat, float* __restrict__ Pout, float* __restrict__ Vout,
          float* __restrict__ Pinv, int pstride) {
  using H = H16<Gm, F, SPLIT>;
  constexpr int P = Gm::P, A = Gm::A, NVT = H::NVT, NPT = H::NPT, LP = H::LP, SV = H::SV, NB = H::NB, RS = H::RS;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) float h16_lds[];
  float* As = h16_lds;                                              // [NB][16 boards][4 g slots][4 s]
  float* s_tr = h16_lds + H::A_FLOATS;                              // [WAVES][3][16 * RS]
  float* s_vh = s_tr + H::PATCH_FLOATS;
  float* s_logit = s_vh + 16 * SV;
  const int n = n_eval_ptr ? *n_eval_ptr : n_fixed;
  const int board0 = (SPLIT ? (int)blockIdx.x >> 1 : (int)blockIdx.x) * 16;
  if (board0 >= n) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  unsigned long long* dbg = (net.dbg && lane == 0) ? net.dbg + ((size_t)blockIdx.x * H::WAVES + wave) * 4 : nullptr;   // az_debug_heads_timeline
  if (dbg) dbg[0] = __builtin_readcyclecounter();
  // this wavefront's chain: a policy tile pt or a value tile vt (split form: a workgroup has one kind, wavefronts beyond its
  // tiles only help with phase 1)
  const bool is_pol = SPLIT ? (blockIdx.x & 1) != 0 : wave < NPT;
  const int pt = wave, vt = SPLIT ? wave : wave - NPT;
  const bool active = !SPLIT || wave < (is_pol ? NPT : NVT);
  const int HF = net.HF, foff = is_pol ? 0 : net.npf;
  if constexpr (H::STAGE) {
    // phase 1: feature f = 4 c + i of (board b, position p) -> block 2 p + (c >> 2), s = c & 3, g = i
    for (int idx = threadIdx.x; idx < 16 * P * 8; idx += H::THREADS) {
      const int c = idx & 7, p = (idx >> 3) % P, b = idx / (8 * P);
      int eb = board0 + b;
      if (eb >= n) eb = n - 1;                     // clamp: rows past the batch are computed and dropped
      const f32x4v v = *(const f32x4v*)(hfeat + ((size_t)eb * P + p) * HF + foff + 4 * c);
      float* dst = As + ((2 * p + (c >> 2)) * 16 + b) * 16 + (c & 3);
      const int rot = 2 * ((b >> 2) & 1);
      dst[4 * ((0 + rot) & 3)] = v.x; dst[4 * ((1 + rot) & 3)] = v.y; dst[4 * ((2 + rot) & 3)] = v.z; dst[4 * ((3 + rot) & 3)] = v.w;
    }
    __syncthreads();
  }
  if (active) {
  const f32x4v* wp = (const f32x4v*)net.hd16_w + (size_t)(is_pol ? NVT + pt : vt) * NB * 64 + lane;   // fragments: value tiles, then policy tiles
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#define H16_MFMA(a0, a1, a2, a3, b) do {                                         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, (b).x, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, (b).y, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, (b).z, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, (b).w, acc, 0, 0, 0);         \
    __builtin_amdgcn_sched_barrier(0); } while (0)
  if constexpr (H::STAGE) {
    // phase 2: A from the staged features, two blocks ahead of the MFMAs; weights WD blocks ahead
    constexpr int WD = HEADS16_WDEPTH;
    static_assert(NB >= WD && WD % 2 == 0, "chain shorter than the pipeline");
    const float* ap = As + m * 16 + 4 * ((g + 2 * ((m >> 2) & 1)) & 3);
    f32x4v gb[WD], av[3];
    static_for<WD>([&](auto dc) { constexpr int d = decltype(dc)::value; gb[d] = wp[(size_t)d * 64]; });
    av[0] = *(const f32x4v*)ap;
    av[1] = *(const f32x4v*)(ap + 256);
    constexpr int CH = WD % 3 == 0 ? WD : 3 * WD, MAIN = NB >= WD + 2 ? (NB - WD - 2) / CH : 0, TAIL = NB - MAIN * CH;
    const f32x4v* wpj = wp;
    const float* apj = ap;
    for (int c = 0; c < MAIN; ++c) {
      static_for<CH>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        const f32x4v b = gb[d % WD];
        av[(d + 2) % 3] = *(const f32x4v*)(apj + (d + 2) * 256);
        gb[d % WD] = wpj[(size_t)(d + WD) * 64];
        __builtin_amdgcn_sched_barrier(0);
        H16_MFMA(av[d % 3].x, av[d % 3].y, av[d % 3].z, av[d % 3].w, b);
      });
      wpj += (size_t)CH * 64;
      apj += CH * 256;
    }
    static_for<TAIL>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      const f32x4v b = gb[d % WD];
      if constexpr (d + 2 < TAIL) av[(d + 2) % 3] = *(const f32x4v*)(apj + (d + 2) * 256);
      if constexpr (d + WD < TAIL) gb[d % WD] = wpj[(size_t)(d + WD) * 64];
      __builtin_amdgcn_sched_barrier(0);
      H16_MFMA(av[d % 3].x, av[d % 3].y, av[d % 3].z, av[d % 3].w, b);
    });
  } else {
    // a wavefront that fetches and transposes its own features
    constexpr int DEPTH = HEADS16_DEPTH;
    static_assert(NB >= DEPTH && DEPTH >= 3 && DEPTH % 2 == 0, "pipeline depth");
    int e = board0 + m;
    if (e >= n) e = n - 1;
    const float* hf = hfeat + (size_t)e * P * HF + foff + 4 * g;
    float* patch = s_tr + (size_t)wave * 3 * 16 * RS;
    float* trw = patch + m * RS + 4 * g;           // this lane's float4 of a patch
    const float* trr = patch + m * RS + g;         // ... and its column (stride 4)
    f32x4v ga[DEPTH], gb[DEPTH];
    float ta[3][4];
    // block j: features 16 (j & 1) .. + 15 of position j >> 1, fragment j of this wave's tile
#define H16_LOAD(hfj, wpj, d, st) do { ga[st] = *(const f32x4v*)((hfj) + ((d) >> 1) * HF + ((d) & 1) * 16); gb[st] = (wpj)[(size_t)(d) * 64]; } while (0)
    // ga[st] -> ta[par] through patch `par`
#define H16_TRANSPOSE(st, par) do {                                              \
      *(f32x4v*)(trw + (par) * 16 * RS) = ga[st];                                \
      __builtin_amdgcn_wave_barrier();                                           \
      ta[par][0] = trr[(par) * 16 * RS]; ta[par][1] = trr[(par) * 16 * RS + 4];  \
      ta[par][2] = trr[(par) * 16 * RS + 8]; ta[par][3] = trr[(par) * 16 * RS + 12]; } while (0)
    static_for<DEPTH>([&](auto dc) { constexpr int d = decltype(dc)::value; H16_LOAD(hf, wp, d, d); });
    H16_TRANSPOSE(0, 0);
    H16_TRANSPOSE(1, 1);
    // Block j's MFMAs run on ta[j % 3]; block j + 2 goes through the patch meanwhile; block j + DEPTH leaves global
    // memory.  Steady state in chunks of lcm(3, DEPTH) blocks (static register indices), then a fully unrolled tail.
    constexpr int CH = DEPTH % 3 == 0 ? DEPTH : 3 * DEPTH, MAIN = NB >= DEPTH + 2 ? (NB - DEPTH - 2) / CH : 0, TAIL = NB - MAIN * CH;
    const float* hfj = hf;
    const f32x4v* wpj = wp;
    for (int c = 0; c < MAIN; ++c) {
      static_for<CH>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        const f32x4v b = gb[d % DEPTH];
        H16_TRANSPOSE((d + 2) % DEPTH, (d + 2) % 3);
        H16_LOAD(hfj, wpj, d + DEPTH, d % DEPTH);  // ga[d % DEPTH] went through the patch two blocks ago, gb[d % DEPTH] is in `b`
        __builtin_amdgcn_sched_barrier(0);
        H16_MFMA(ta[d % 3][0], ta[d % 3][1], ta[d % 3][2], ta[d % 3][3], b);
      });
      hfj += (CH / 2) * HF;
      wpj += (size_t)CH * 64;
    }
    static_for<TAIL>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      const f32x4v b = gb[d % DEPTH];
      if constexpr (d + 2 < TAIL) H16_TRANSPOSE((d + 2) % DEPTH, (d + 2) % 3);
      if constexpr (d + DEPTH < TAIL) H16_LOAD(hfj, wpj, d + DEPTH, d % DEPTH);
      __builtin_amdgcn_sched_barrier(0);
      H16_MFMA(ta[d % 3][0], ta[d % 3][1], ta[d % 3][2], ta[d % 3][3], b);
    });
#undef H16_LOAD
#undef H16_TRANSPOSE
  }
#undef H16_MFMA
  if (dbg) dbg[1] = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * g + i;                     // board of the tile; column m = output
    if (is_pol) { if (pt * 16 + m < A) s_logit[row * LP + pt * 16 + m] = acc[i] + net.pol_b[pt * 16 + m]; }
    else {
      const int o = vt * 16 + m;
      const float v = acc[i] + net.val_b[o];
      s_vh[row * SV + o] = v > 0.0f ? v : 0.0f;
    }
  }
  }
  __syncthreads();
  if (dbg) dbg[2] = __builtin_readcyclecounter();
  const int b = threadIdx.x;
  if (b < 16 && board0 + b < n) {
    if constexpr (!SPLIT) heads_finish<Gm, F>(net, leaf_env, eval_slots, Amask, Pout, Vout, Pinv, pstride, board0 + b, s_logit + b * LP, s_vh + b * SV);
    else if (is_pol) heads_finish_policy<Gm>(leaf_env, eval_slots, Amask, Pout, Pinv, pstride, board0 + b, s_logit + b * LP);
    else heads_finish_value<F>(net, Vout, board0 + b, s_vh + b * SV);
  }
  if (dbg) dbg[3] = __builtin_readcyclecounter();
}
