// resnet16b.h -- k_tower16b: the residual tower in bfloat16 on v_mfma_f32_16x16x32_bf16 (BASELINE configs[4]: "ResNet-10x128 bf16").
//
// Same shape as k_tower16 (resnet16.h): a workgroup owns TB whole boards = NT row tiles of 16 in ONE LDS activation
// buffer, rows in Geo16's border-class order (taps that fall off the board skipped per tile), stem + every residual block +
// both 1x1 head convolutions without leaving the CU.  A wavefront owns half of the row tiles x 32 output channels (two
// column tiles): with 16x the MFMA rate of fp32 the LDS reads of the activations are what binds, and every fragment read
// now feeds two MFMAs (the first version, 16 channels x all tiles per wave, re-read each row 8 times per tap at F = 128).
// What differs:
//   * activations live in LDS as bf16 ([RPAD + 8][SH], 288-byte rows at F = 128), weights are bf16 fragments, accumulation
//     is fp32 in the MFMA (at F = 128 the kernel needs 192 VGPRs, so ONE 8-wave workgroup runs per CU and nothing covers its
//     barriers and epilogue; forcing 128 VGPRs spills 468 registers);
//     the folded batch norm, the residual add and the ReLU run in fp32 on the accumulators and the result is rounded to
//     bf16 (round to nearest even, v_cvt_pk_bf16_f32) when it is written back -- the skip connection adds the ROUNDED
//     block input, i.e. exactly what the next convolution reads;
//   * one MFMA consumes 32 input channels: lane (row r = lane & 15, group g = lane >> 4) supplies channels
//     32 ks + 8 g .. + 7 of row r as one ds_read_b128, and the matching weight fragment holds W[those channels][co]; the
//     hardware's pairing of k within a group is irrelevant because A and B use the same one;
//   * the stem (K = 27) stays on the fp32 MFMA with the fp32 planes (it is 0.1 % of the work), the head features leave
//     in fp32 for k_heads_mfma, so everything outside the tower is unchanged.
// Round 3 (10x128, 4096 boards: 753 -> 578 us per launch = 1.76 PFLOP/s = 70 % of the nominal bf16 peak, 86 % of the
// 2.05 PFLOP/s the chip sustains at the ~2.0 GHz it holds under matrix load, tools/probes/mfma_sustained.hip):
//   * NT = 22: 8 Connect-Four boards per workgroup, a wavefront = 11 row tiles x 2 column tiles: the weight fragments of a
//     tap serve twice as many rows (the vector-memory path delivers ~71 B/clk per CU, tools/probes/l1_bw.hip, and the
//     4-board form asked it for 45), 253-256 VGPRs, no spill;
//   * Geo16 places the rows so that tap-shifted gathers are bank-conflict free (resnet16.h): the LDS reads of the A operand,
//     at 98 % of the LDS cycle budget when the MFMA pipe is full, ran at 59 % of the LDS rate before;
//   * branch-free epilogue (store_bf16x4): 4.9 k -> 2.5 k cycles per layer and wavefront pair.
//   Measured and not kept: one row group per workgroup with two workgroups per CU (4 waves x 253 VGPRs each, the fp32
//   tower's arrangement: 609 us -- with both workgroups in their convolutions the LDS reads bind, not the MFMA pipe);
//   steps of half a tap's input channels with the rows in a ring of 2-4 register stages and look-ups two steps ahead
//   (no measurable change: 670 vs 669 us); 64 output channels per wavefront (halves the LDS reads, doubles the weight
//   stream to 91 B/clk per CU: more than the path delivers).
// Numerics: NOT the fp32 contract.  tests/test_net_bf16_gpu.py holds it to (a) a torch emulation of exactly this scheme
// (bf16-rounded weights and activations, wide accumulation) within 2e-3 and (b) the fp32 oracle within the stated bf16
// tolerance; searches run with it are deterministic but not comparable move for move with the fp32 oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "resnet16.h"

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

struct Net16bDev {
  int nblocks;
  const float* stem_w;      // fp32 fragments of the stem, as Net16Dev
  const float* stem_ss;     // [2][F]
  const bf16x8v* conv_w;    // [2*nblocks][9][F/16 col tiles][F/32 k steps][64] x 8 bf16
  const float* conv_ss;     // [2*nblocks][2][F]
  const bf16x8v* head_w;    // [F/16][F/32][64] x 8 bf16
  const float* head_ss;     // [2][F]
  const uint16_t* geo[3];   // Geo16 tables: [0] 11 tiles, [1] the latency variant's (NTS), [2] 22 tiles
  unsigned long long* dbg;  // optional [workgroups][8] cycle stamps, as Net16Dev::dbg (az_debug_tower_timeline)
};

template <class Gm, int F = 128, int NT = 11> struct T16B {
  static constexpr int NTILE = NT, RPAD = NTILE * 16;
  static constexpr int TB = RPAD / Gm::P, ROWS = TB * Gm::P;
  // bf16 elements per buffer row.  A ds_read_b128 of the A operand (lane = (row lrow, k group g), 16 bytes at
  // row * SH + 8 g) runs at the full 256 B/clk only if the lanes (lrow 0..7, g and g + 1) of a pass cover all 64 banks:
  // with rows of 4 d dwords that needs { d lrow + g } distinct mod 16 -- d = 9 (F = 64: 144-byte rows) or 18 (F = 128:
  // 288-byte rows); 272-byte rows (d = 17) halve the rate (tools/probes/lds_conflict.hip)
  static constexpr int SH = F == 128 ? F + 16 : F + 8;
  static constexpr int BUFH = (RPAD + GEO_NZ) * SH;      // rows RPAD .. RPAD + 7 = zeros
  static constexpr int PLANES = (RPAD + GEO_NZ) * Gm::C; // fp32 input planes for the stem
  static constexpr int TABLE = (10 * RPAD + 1) / 2;
  static constexpr int BYTES = BUFH * 2 + (PLANES + TABLE) * 4;
  static constexpr int WAVES = F / 16, THREADS = 64 * WAVES, CT = F / 16, KS = F / 32;
  using Game = Gm;
  using Geo = Geo16<Gm, NTILE, TB>;
  static constexpr int FILT = F;
  static_assert(BUFH * 2 % 16 == 0 && (SH * 2) % 16 == 0, "rows must stay 16-byte aligned");
};

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// Epilogue store of one accumulator tile column: this lane's four outputs v[i] (rows row0 + i, channel ch) are rounded
// to bf16 (packed[0] = rows 0, 1; packed[1] = rows 2, 3 -- what the skip connection adds later) and written to the buffer.
// The lanes of channels c and c + 1 (lrow even / odd) swap halves through DPP so that each writes TWO 32-bit words
// (channels c, c + 1 of one row) instead of four 16-bit ones: half the LDS instructions and no two lanes on one dword
// (the 16-bit version of this epilogue took 6.4 k of a layer's 19.9 k cycles at 10x128, tools/tower_timeline.py --bf16).
// Round 3: branch free.  The even / odd cases were two exec-masked blocks and ~25 VALU instructions per call; now the lane
// keeps `mine` (even: rows 0, 1; odd: rows 2, 3), sends the other pair, and two v_perm_b32 with per-lane selectors
// (EpiSel, computed once per wavefront) interleave the halves: 2 cvt + 2 select + 1 DPP + 2 perm + 1 ds_write2.
struct EpiSel { uint32_t s0, s1; bool odd; };
__device__ __forceinline__ EpiSel epi_sel(int lrow) {
  // v_perm_b32(a, b, sel): result byte i = byte sel[i] of {a (bytes 4..7), b (bytes 0..3)}; here a = received, b = mine
  const bool odd = lrow & 1;
  // even lane: w0 = lo(mine) | lo(recv) << 16, w1 = hi(mine) | hi(recv) << 16;  odd lane: w0 = lo(recv) | lo(mine) << 16, w1 = hi(recv) | hi(mine) << 16
  return EpiSel{odd ? 0x01000504u : 0x05040100u, odd ? 0x03020706u : 0x07060302u, odd};
}
template <class T>
__device__ __forceinline__ void store_bf16x4(uint16_t* __restrict__ dst, const EpiSel& es, const float (&v)[4], uint32_t (&packed)[2]) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  packed[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2v{v[0], v[1]}, bf16x2v));
  packed[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2v{v[2], v[3]}, bf16x2v));
  const uint32_t mine = es.odd ? packed[1] : packed[0], send = es.odd ? packed[0] : packed[1];
  const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, false);     // quad_perm [1, 0, 3, 2]: lane ^ 1
  *(uint32_t*)dst = __builtin_amdgcn_perm(recv, mine, es.s0);
  *(uint32_t*)(dst + T::SH) = __builtin_amdgcn_perm(recv, mine, es.s1);
}
// buffer address of the words a lane writes for (row0, ch): rows row0 (+ 2 for odd lanes), channels ch & ~1
template <class T> __device__ __forceinline__ uint16_t* epi_dst(uint16_t* __restrict__ buf, int row0, int ch, bool odd) {
  return buf + (row0 + (odd ? 2 : 0)) * T::SH + (ch & ~1);
}

// activation rows of one step: KS fragments of 8 channels per tile
template <class T>
__device__ __forceinline__ void load_rows16b(const uint16_t* __restrict__ buf, int off0, int off1, bool two, int g, bf16x8v (&a)[2][T::KS]) {
  const uint16_t* p0 = buf + off0 + g * 8;
#pragma unroll
  for (int ks = 0; ks < T::KS; ++ks) a[0][ks] = *(const bf16x8v*)(p0 + ks * 32);
  if (two) {
    const uint16_t* p1 = buf + off1 + g * 8;
#pragma unroll
    for (int ks = 0; ks < T::KS; ++ks) a[1][ks] = *(const bf16x8v*)(p1 + ks * 32);
  }
}
template <class T, class SL, int K, int TILE0>
__device__ __forceinline__ void load_idx16b(const uint16_t* __restrict__ nbr, int lrow, int (&idx)[2]) {
  if constexpr (K < SL::list.n) {
    constexpr int tap = SL::list.tap[K], t0 = SL::list.t0[K], t1 = SL::list.t1[K];
    idx[0] = (int)nbr[tap * T::RPAD + (TILE0 + t0) * 16 + lrow] * T::SH;
    if constexpr (t1 >= 0) idx[1] = (int)nbr[tap * T::RPAD + (TILE0 + t1) * 16 + lrow] * T::SH;
  }
}
// weights of one tap for this wave's two column tiles: b[ct][ks]
template <class T>
__device__ __forceinline__ void load_w16b(const bf16x8v* __restrict__ wl, int tap, bf16x8v (&b)[2][T::KS]) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < T::KS; ++ks) b[ct][ks] = wl[(size_t)((tap * T::CT + ct) * T::KS + ks) * 64];
}
template <class T, class SL, int NT, int TILE0, int NTAP, int K>
__device__ __forceinline__ void conv16b_steps(const uint16_t* __restrict__ buf, const uint16_t* __restrict__ nbr, const bf16x8v* __restrict__ wl,
                                              f32x4v (&acc)[NT][2], int lrow, int g, bf16x8v (&b0)[2][T::KS], bf16x8v (&b1)[2][T::KS],
                                              bf16x8v (&aA)[2][T::KS], bf16x8v (&aB)[2][T::KS], int (&idx)[2]) {
  if constexpr (K < SL::list.n) {
    constexpr int t0 = SL::list.t0[K], t1 = SL::list.t1[K], ord = SL::list.ord[K];
    bf16x8v (&cur)[2][T::KS] = (K & 1) ? aB : aA;
    bf16x8v (&nxt)[2][T::KS] = (K & 1) ? aA : aB;
    bf16x8v (&bc)[2][T::KS] = (ord & 1) ? b1 : b0;
    bf16x8v (&bn)[2][T::KS] = (ord & 1) ? b0 : b1;
    if constexpr (SL::list.first[K] && SL::list.next_tap[K] >= 0) load_w16b<T>(wl, NTAP == 1 ? 0 : SL::list.next_tap[K], bn);
    if constexpr (K + 1 < SL::list.n) {
      load_rows16b<T>(buf, idx[0], idx[1], SL::list.t1[K + 1] >= 0, g, nxt);
      load_idx16b<T, SL, K + 2, TILE0>(nbr, lrow, idx);
    }
    // every activation fragment feeds both column tiles: half the LDS reads per MFMA of a 16-channel wave
#pragma unroll
    for (int ks = 0; ks < T::KS; ++ks) {
      acc[t0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[0][ks], bc[0][ks], acc[t0][0], 0, 0, 0);
      acc[t0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[0][ks], bc[1][ks], acc[t0][1], 0, 0, 0);
      if constexpr (t1 >= 0) {
        acc[t1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[1][ks], bc[0][ks], acc[t1][0], 0, 0, 0);
        acc[t1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[1][ks], bc[1][ks], acc[t1][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    conv16b_steps<T, SL, NT, TILE0, NTAP, K + 1>(buf, nbr, wl, acc, lrow, g, b0, b1, aA, aB, idx);
  }
}
// One F -> F convolution (NTAP = 9: 3x3, NTAP = 1: 1x1) into this wave's NT x 2 accumulators: bf16 in, fp32 out.
template <class T, class G, int NT, int TILE0, int NTAP>
__device__ __forceinline__ void conv16b(const uint16_t* __restrict__ buf, const uint16_t* __restrict__ nbr, const bf16x8v* __restrict__ wl,
                                        f32x4v (&acc)[NT][2], int lrow, int g) {
  using SL = Steps16<G, NT, TILE0, 1, NTAP>;                        // one step = (tap, tile pair), all F channels
  if constexpr (SL::list.n > 0) {
    bf16x8v b0[2][T::KS], b1[2][T::KS], aA[2][T::KS], aB[2][T::KS];
    int idx[2] = {0, 0};
    load_w16b<T>(wl, NTAP == 1 ? 0 : SL::list.first_tap, b0);
    load_idx16b<T, SL, 0, TILE0>(nbr, lrow, idx);
    load_rows16b<T>(buf, idx[0], idx[1], SL::list.t1[0] >= 0, g, aA);
    load_idx16b<T, SL, 1, TILE0>(nbr, lrow, idx);
    __builtin_amdgcn_sched_barrier(0);
    conv16b_steps<T, SL, NT, TILE0, NTAP, 0>(buf, nbr, wl, acc, lrow, g, b0, b1, aA, aB, idx);
  }
}

// One wavefront's share: row tiles TILE0 .. TILE0 + NT - 1 of the workgroup's buffer, output channels 32 cg .. 32 cg + 31
// (two column tiles).  Every wavefront of the workgroup runs the same number of barriers.
template <class T, int NT, int TILE0>
__device__ __forceinline__ void tower16b_wave(const Net16bDev& net, uint16_t* __restrict__ buf, const float* __restrict__ planes,
                                              const uint16_t* __restrict__ nbr, const uint16_t* __restrict__ pos, int cg, int lane,
                                              int n, int board0, float* __restrict__ hfeat) {
  using Gm = typename T::Game;
  using G = typename T::Geo;
  constexpr int F = T::FILT, P = Gm::P, C = Gm::C, TB = T::TB, R0 = TILE0 * 16;
  const int lrow = lane & 15, g = lane >> 4;
  const int ch0 = cg * 32 + lrow;                                   // this lane's output channels: ch0 and ch0 + 16; rows tile*16 + 4 g + i
  unsigned long long* dbg = (net.dbg && threadIdx.x == 0) ? net.dbg + (size_t)blockIdx.x * 8 : nullptr;
#define AZ_STAMP16B(i) do { if (dbg) dbg[i] = __builtin_readcyclecounter(); } while (0)
  AZ_STAMP16B(0);
  f32x4v acc[NT][2];
  const EpiSel es = epi_sel(lrow);
  uint32_t xres[NT][2][2];                                          // the current block's input as the convolutions read it (bf16 pairs: rows 0, 1 | 2, 3): this lane wrote it
  // ---- stem on the fp32 MFMA (K = 9 C), output rounded to bf16 -------------------------------------------------------------
  {
    constexpr int KK = 9 * C, K2 = (KK + 1) / 2, NS = (2 * K2 + 3) / 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float bw0 = net.stem_w[(size_t)((cg * 2) * NS + s) * 64 + lane], bw1 = net.stem_w[(size_t)((cg * 2 + 1) * NS + s) * 64 + lane];
      const int p = 4 * s + g;
      const int k = (p & 1) * K2 + (p >> 1);
      const bool kin = k < KK && p < 2 * K2;
      const int tap = kin ? k / C : 4, c = kin ? k % C : 0;
#pragma unroll
      for (int tile = 0; tile < NT; ++tile) {
        const int row = kin ? (int)nbr[tap * T::RPAD + R0 + tile * 16 + lrow] : T::RPAD;
        const float a = planes[row * C + c];
        acc[tile][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw0, acc[tile][0], 0, 0, 0);
        acc[tile][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw1, acc[tile][1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float sc = net.stem_ss[ch0 + 16 * ct], sh = net.stem_ss[F + ch0 + 16 * ct];
#pragma unroll
      for (int tile = 0; tile < NT; ++tile) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float y = az_fmaf(acc[tile][ct][i], sc, sh); v[i] = y > 0.0f ? y : 0.0f; }
        store_bf16x4<T>(epi_dst<T>(buf, R0 + tile * 16 + g * 4, ch0 + 16 * ct, es.odd), es, v, xres[tile][ct]);
      }
    }
  }
  __syncthreads();
  AZ_STAMP16B(1);

  // ---- residual tower -------------------------------------------------------------------------------------------------------
  const size_t LAYER_W = (size_t)9 * T::CT * T::KS * 64;            // fragments (16 B) per layer
  for (int layer = 0; layer < 2 * net.nblocks; ++layer) {
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    int lrow_l = lrow;
    asm volatile("" : "+v"(lrow_l));                                // keeps the table look-ups inside the layer loop (registers)
    conv16b<T, G, NT, TILE0, 9>(buf, nbr, net.conv_w + (size_t)layer * LAYER_W + (size_t)(cg * 2) * T::KS * 64 + lane, acc, lrow_l, g);
    if (layer == 2) AZ_STAMP16B(4);
    __syncthreads();                                                // every wave has finished reading the buffer
    if (layer == 2) AZ_STAMP16B(5);
    // the two kinds of layer are two uniform branches (as one body the residual add was computed and then selected away
    // in every first convolution of a block)
    if (layer & 1) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float sc = net.conv_ss[(size_t)layer * 2 * F + ch0 + 16 * ct], sh = net.conv_ss[(size_t)layer * 2 * F + F + ch0 + 16 * ct];
#pragma unroll
        for (int tile = 0; tile < NT; ++tile) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {                             // second convolution of a block: + the block input (rounded, as read)
            const uint32_t pr = xres[tile][ct][i >> 1];
            const float y = az_fmaf(acc[tile][ct][i], sc, sh) + __builtin_bit_cast(float, (i & 1) ? (pr & 0xffff0000u) : (pr << 16));
            v[i] = y > 0.0f ? y : 0.0f;
          }
          store_bf16x4<T>(epi_dst<T>(buf, R0 + tile * 16 + g * 4, ch0 + 16 * ct, es.odd), es, v, xres[tile][ct]);   // also the next block's input
        }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float sc = net.conv_ss[(size_t)layer * 2 * F + ch0 + 16 * ct], sh = net.conv_ss[(size_t)layer * 2 * F + F + ch0 + 16 * ct];
#pragma unroll
        for (int tile = 0; tile < NT; ++tile) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float y = az_fmaf(acc[tile][ct][i], sc, sh); v[i] = y > 0.0f ? y : 0.0f; }
          uint32_t pk[2];
          store_bf16x4<T>(epi_dst<T>(buf, R0 + tile * 16 + g * 4, ch0 + 16 * ct, es.odd), es, v, pk);
        }
      }
    }
    __syncthreads();
    if (layer == 1 || layer == 2) AZ_STAMP16B(5 + layer);
  }
  AZ_STAMP16B(2);
  // ---- both 1x1 head convolutions + BN + ReLU, features out in fp32 -------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
  conv16b<T, G, NT, TILE0, 1>(buf, nbr, net.head_w + (size_t)(cg * 2) * T::KS * 64 + lane, acc, lrow, g);
  {
    const int nvalid = ((n - board0) < TB ? (n - board0) : TB) * P;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float sc = net.head_ss[ch0 + 16 * ct], sh = net.head_ss[F + ch0 + 16 * ct];
#pragma unroll
      for (int tile = 0; tile < NT; ++tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ps = pos[R0 + tile * 16 + g * 4 + i];
          const float v = az_fmaf(acc[tile][ct][i], sc, sh);
          if (ps < nvalid) hfeat[((size_t)board0 * P + ps) * F + ch0 + 16 * ct] = v > 0.0f ? v : 0.0f;
        }
    }
  }
  AZ_STAMP16B(3);
#undef AZ_STAMP16B
}

template <class Gm, int F, bool FROM_PLANES, int NT = 11>
__global__ void __launch_bounds__(64 * (F / 16), F == 64 ? 2 : 1)
k_tower16b(Net16bDev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
           const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat) {
  using T = T16B<Gm, F, NT>;
  constexpr int P = Gm::P, C = Gm::C, TB = T::TB, SH = T::SH, NCG = F / 32, NT0 = (NT + 1) / 2, NT1 = NT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  uint16_t* buf = (uint16_t*)ldsb;
  float* planes = (float*)(ldsb + (size_t)T::BUFH * 2);
  uint16_t* nbr = (uint16_t*)(planes + T::PLANES);
  uint16_t* pos = nbr + 9 * T::RPAD;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  const int board0 = blockIdx.x * TB;
  if (board0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* geo = net.geo[NT == 11 ? 0 : NT == 22 ? 2 : 1];
  // ---- input planes (fp32, permuted row order), tables, the buffer's zero row -------------------------------------------
  for (int i = tid; i < T::PLANES; i += T::THREADS) {
    const int row = i / C, c = i % C;
    float val = 0.0f;
    if (row < T::RPAD) {
      const int ps = geo[row];
      if (ps != 0xffff && board0 + ps / P < n) {
        const int b = ps / P, q = ps % P;
        if (FROM_PLANES) val = X[((size_t)(board0 + b) * C + c) * P + q];
        else val = Gm::plane(leaf_env[eval_slots[board0 + b]], q, c);
      }
    }
    planes[i] = val;
  }
  for (int i = tid; i < GEO_NZ * SH; i += T::THREADS) buf[T::RPAD * SH + i] = 0;
  for (int i = tid; i < T::RPAD; i += T::THREADS) pos[i] = geo[i];
  for (int i = tid; i < 9 * T::RPAD; i += T::THREADS) nbr[i] = geo[T::RPAD + i];
  __syncthreads();
  // wavefront = (row group, column group): row group 0 owns tiles 0 .. NT0 - 1, row group 1 the rest; a column group is 32
  // output channels, so an activation fragment read from LDS feeds two MFMAs
  if (wave < NCG) tower16b_wave<T, NT0, 0>(net, buf, planes, nbr, pos, wave, lane, n, board0, hfeat);
  else tower16b_wave<T, NT1, NT0>(net, buf, planes, nbr, pos, wave - NCG, lane, n, board0, hfeat);
}

