// resnet16.h -- k_tower16: the 64-filter residual tower on v_mfma_f32_16x16x4_f32 with 16x16 work units.
//
// Why a second tower kernel.  k_tower (resnet.h) quantises the batch in workgroups of 3 boards x 2 per CU:
// 4096 boards are 2.67 "rounds" that cost 3 (-11 %).  Here a workgroup owns TB = 4 Connect-Four boards =
// 168 rows = 10.5 -> 11 row tiles of 16, and keeps ONE activation buffer in LDS (51 KB, so two workgroups
// still share a CU): 4096 boards = 1024 workgroups = exactly two full rounds of 512.  Wave w owns the 16 output
// channels w*16..w*16+15 of every row tile: 11 accumulators of 4 VGPRs, 176 MFMAs per tap against 4 weight
// loads (k_tower: 64 MFMAs against 8 loads), so the VMEM issue cost of the weight stream disappears.
//
// Single-buffer schedule of one residual block (resnet.jl:53-63):
//   conv1: read X (LDS) -> acc            | barrier | own X values -> registers (residual), T = relu(bn(acc)) -> LDS | barrier
//   conv2: read T (LDS) -> acc            | barrier | y = relu(bn(acc) + X registers) -> LDS                        | barrier
//
// fp32 contract (include/azhip.h) unchanged: v_mfma_f32_16x16x4_f32 is a k-ordered fma chain with k = lane >> 4
// (probe: tools/probes/mfma16_chain.hip), and step s of a 64-channel tap feeds it the channels
// (2s, 32+2s, 2s+1, 32+2s+1) = the contract's order j, 32+j.  To make each lane's 16 values of a tap contiguous,
// LDS stores channel c at position pos(c) = ((c >> 5) + 2 (c & 1)) * 16 + ((c & 31) >> 1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/az_numerics.h"
#include "games.h"
#include "resnet.h"


struct Net16Dev {
  int nblocks;
  const float* stem_w;      // [4 col tiles][K2s steps][64] : W[k(4s + g)][col]
  const float* stem_ss;     // [2][64]
  const float4* conv_w;     // [2*nblocks][9][4 col tiles][4][64] float4
  const float* conv_ss;     // [2*nblocks][2][64]
  const float4* head_w;     // [4 col tiles][4][64] float4
  const float* head_ss;     // [2][64]
  const uint16_t* geo[5];   // row permutation tables (Geo16: pos [RPAD], nbr [9][RPAD]) of the 11-tile, 3-tile and 21-tile kernels, [3]: the exact-fit variant (NTM), [4]: the 19-tile paired form
  unsigned long long* dbg;  // optional [workgroups][8] s_memtime stamps (az_debug_tower_timeline): 0 start, 1 stem done, 2 tower done, 3 features written; layer 2: 6 start, 4 convolution done, 5 barrier passed, 7 epilogue done
};
// ---------------------------------------------------------------------------------------------------------------
// Row permutation of the tower kernels: skipping the taps that fall off the board.
//
// A 3x3 convolution on a W x H board multiplies zeros for every (position, tap) whose neighbour is outside: 19.6 % of
// the pairs on 7x6 (Connect-Four), two thirds on 14x1 (Mancala).  The implicit GEMM works on tiles of 16 rows, so a tap
// can be dropped for a tile only if it is outside for ALL 16 rows.  The rows of a workgroup's buffer are therefore
// ordered by border class instead of by position: interior cells first (every tap inside), then the left column, the
// right column, the bottom row, the top row (order of the four edge classes and the place of the padding rows chosen
// at compile time to minimise the number of (tile, tap) products).  Connect-Four, 4 boards in 11 tiles: 85 products
// instead of 99 (-14 %); 8 boards in 21 tiles: 159 instead of 189 (-16 %); Mancala: 31 of 99.  A skipped product would
// have added 0 x w to the accumulator, which leaves every fp32 chain of the contract bit-for-bit unchanged (the
// accumulators start at +0 and never become -0).
// What changes in the kernel: the tap-shifted row of a lane is no longer `row + delta` but a table look-up
// (nbr[tap][row], buffer offset of the neighbour or of the zero row; built on the host from the same constexpr
// function, staged in LDS), read one pipeline step ahead of the activation rows it addresses; the list of
// (tap, tile pair) steps is a compile-time list, so the unrolled MFMA stream simply has fewer steps.
//
// Bank-conflict-free gathers (round 3).  The A operand of a tap is a GATHER of tap-shifted rows, and a ds_read_b128 pass
// (8 rows x 2 adjacent k groups, resnet16b.h / posF) runs at the full LDS rate only if its 8 rows are distinct mod 8 (rows
// of 72 or 136 dwords: the bank offset of a row is 8 (row mod 8)).  With the cells of a class in plain (board, position)
// order the shifted rows of a pass collide -- 119 extra LDS cycles over the 170 passes of a Connect-Four convolution, 59 %
// of the LDS rate, which is what bounds the bf16 tower once two workgroups keep the MFMA pipe fed.  So inside every class
// segment the cells are placed such that row = L(cell) (mod 8) for ONE linear function L = a x + b y + g board of the whole
// layout (a, b, g searched at compile time; Connect-Four, 4 boards: x + 4 y + 5 board with 2 of 168 cells off; 8 boards:
// L = board).  The 8 rows of a pass have 8 distinct residues, their tap-shifted neighbours have L + (a dx + b dy): distinct
// again.  Off-board neighbours read one of GEO_NZ = 8 zero rows, the one whose residue the neighbour would have had.
static constexpr int GEO_NZ = 8;
template <class Gm, int NTILES, int TBOARDS> struct Geo16 {
  static constexpr int P = Gm::P, W = Gm::W, H = Gm::H, RPAD = NTILES * 16, ROWS = TBOARDS * P, NPADR = RPAD - ROWS;
  static_assert(NPADR >= 0, "boards do not fit the tiles");
  struct Tab {
    uint16_t pos[RPAD];            // buffer row -> board * P + position, 0xffff = padding row
    uint16_t nbr[9 * RPAD];        // [tap][row] -> buffer row of the neighbour, RPAD + k = zero row k
    uint16_t tapmask[NTILES];      // bit t: tap t is inside the board for at least one row of the tile
    int cost;                      // sum of popcount(tapmask)
    int la, lb, lg, off;           // the residue function and the number of cells whose row is not = L (mod 8)
  };
  static constexpr int lres(int a, int b, int g, int cell) {
    const int bd = cell / P, q = cell % P;
    return (a * (q % W) + b * (q / W) + g * bd) & 7;
  }
  static constexpr int cls(int q) {
    const int x = q % W, y = q / W;
    if (W >= 2 && x == 0) return 1;
    if (W >= 2 && x == W - 1) return 2;
    if (y == 0) return H == 1 ? 0 : 3;
    if (y == H - 1) return 4;
    return 0;
  }
  // validity bits of the 9 taps at board position q
  static constexpr int valid9(int q) {
    int m = 0;
    for (int tap = 0; tap < 9; ++tap) {
      const int x = q % W + (tap % 3 - 1), y = q / W + (tap / 3 - 1);
      if (x >= 0 && x < W && y >= 0 && y < H) m |= 1 << tap;
    }
    return m;
  }
  struct Order { int ord[5]; };
  static constexpr Order order_of(int order_code) {     // order_code in 0..23 -> interior first, then a permutation of the edge classes
    Order o{{0, 1, 2, 3, 4}};
    int pool[4] = {1, 2, 3, 4}, n = 4, code = order_code;
    for (int i = 0; i < 4; ++i) {
      const int k = code % n; code /= n;
      o.ord[1 + i] = pool[k];
      for (int j = k; j + 1 < n; ++j) pool[j] = pool[j + 1];
      --n;
    }
    return o;
  }
  // number of (tile, tap) products of a candidate layout (no tables built: the search stays cheap for the compiler)
  static constexpr int cost_of(int order_code, int padpos) {
    const Order o = order_of(order_code);
    int v9[P] = {}, ncls[5] = {};
    for (int q = 0; q < P; ++q) { v9[q] = valid9(q); ncls[cls(q)]++; }
    int cost = 0, r = 0, mask = 0;
    auto put = [&](int m, int count) {
      for (int k = 0; k < count; ++k) {
        mask |= m;
        if (++r % 16 == 0) { for (int t = 0; t < 9; ++t) cost += (mask >> t) & 1; mask = 0; }
      }
    };
    for (int ci = 0; ci < 5; ++ci) {
      for (int b = 0; b < TBOARDS; ++b)
        for (int q = 0; q < P; ++q) if (cls(q) == o.ord[ci]) put(v9[q], 1);
      if (ci == padpos) put(0, NPADR);
    }
    return cost;
  }
  // class-ordered rows (as cost_of walks them), then inside every class segment cells and rows matched by residue
  static constexpr int arrange(int order_code, int padpos, int a, int b, int g, bool same_taps, uint16_t (&out)[RPAD]) {
    const Order o = order_of(order_code);
    int v9[P] = {};
    for (int q = 0; q < P; ++q) v9[q] = valid9(q);
    uint16_t cells[RPAD] = {};
    uint8_t res[RPAD] = {};
    int r = 0, off = 0;
    for (int ci = 0; ci < 5; ++ci) {
      int n = 0;
      for (int bd = 0; bd < TBOARDS; ++bd)
        for (int q = 0; q < P; ++q) if (cls(q) == o.ord[ci]) { cells[n] = (uint16_t)(bd * P + q); res[n] = (uint8_t)lres(a, b, g, bd * P + q); ++n; }
      // row r + i wants a cell with residue (r + i) & 7 -- any cell of the class, or (same_taps) only one with the tap set of
      // the plain layout's cell i, which keeps every tile's tap mask exactly what cost_of priced
      bool used[RPAD] = {};
      for (int i = 0; i < n; ++i) {
        const int want = (r + i) & 7, v = v9[cells[i] % P];
        out[r + i] = 0xfffe;
        for (int j = 0; j < n; ++j)
          if (!used[j] && res[j] == want && (!same_taps || v9[cells[j] % P] == v)) { out[r + i] = cells[j]; used[j] = true; break; }
      }
      for (int i = 0; i < n; ++i) if (out[r + i] == 0xfffe) {        // left-overs: any cell of the same tap set
        const int v = v9[cells[i] % P];
        for (int j = 0; j < n; ++j) if (!used[j] && (!same_taps || v9[cells[j] % P] == v)) { out[r + i] = cells[j]; used[j] = true; ++off; break; }
      }
      r += n;
      if (ci == padpos) for (int k = 0; k < NPADR; ++k) out[r++] = 0xffff;
    }
    return off;
  }
  static constexpr int products(const uint16_t (&pos)[RPAD]) {
    int cost = 0;
    for (int t = 0; t < NTILES; ++t) {
      int m = 0;
      for (int i = 0; i < 16; ++i) if (pos[t * 16 + i] != 0xffff) m |= valid9(pos[t * 16 + i] % P);
      for (int tap = 0; tap < 9; ++tap) cost += (m >> tap) & 1;
    }
    return cost;
  }
  // cells that cannot sit on a row of their residue, by histograms only (cheap enough to try all 512 functions)
  static constexpr int mismatch(int order_code, int padpos, int a, int b, int g) {
    const Order o = order_of(order_code);
    int r = 0, off = 0;
    for (int ci = 0; ci < 5; ++ci) {
      int have[8] = {}, n = 0;
      for (int q = 0; q < P; ++q) if (cls(q) == o.ord[ci])
        for (int bd = 0; bd < TBOARDS; ++bd) { have[(a * (q % W) + b * (q / W) + g * bd) & 7]++; ++n; }
      for (int k = 0; k < 8; ++k) {
        const int need = n / 8 + (((k - r) & 7) < n % 8 ? 1 : 0);    // rows r .. r + n - 1 with residue k
        off += have[k] > need ? have[k] - need : 0;
      }
      r += n;
      if (ci == padpos) r += NPADR;
    }
    return off;
  }
  static constexpr Tab build(int order_code, int padpos) {
    Tab t{};
    int ba = 0, bb = 0, bg = 0, boff = RPAD + 1;
    for (int a = 0; a < 8; ++a)
      for (int b = 0; b < 8; ++b)
        for (int g = 0; g < 8; ++g) {
          const int off = mismatch(order_code, padpos, a, b, g);
          if (off < boff) { boff = off; ba = a; bb = b; bg = g; }
        }
    t.la = ba; t.lb = bb; t.lg = bg;
    // free matching inside a class first; if that moved a corner cell into a tile whose mask it widens (one more product
    // than the plain layout has), the matching restricted to equal tap sets
    t.off = arrange(order_code, padpos, ba, bb, bg, false, t.pos);
    if (products(t.pos) > cost_of(order_code, padpos)) t.off = arrange(order_code, padpos, ba, bb, bg, true, t.pos);
    uint16_t row_of[ROWS > 0 ? ROWS : 1] = {};
    for (int i = 0; i < RPAD; ++i) if (t.pos[i] != 0xffff) row_of[t.pos[i]] = (uint16_t)i;
    for (int i = 0; i < NTILES; ++i) t.tapmask[i] = 0;
    for (int i = 0; i < RPAD; ++i) {
      for (int tap = 0; tap < 9; ++tap) {
        const int dx = tap % 3 - 1, dy = tap / 3 - 1;
        int nb = RPAD + ((i + ba * dx + bb * dy + 16) & 7);            // the zero row with the residue the neighbour would have
        if (t.pos[i] != 0xffff) {
          const int b = t.pos[i] / P, q = t.pos[i] % P, x = q % W + dx, y = q / W + dy;
          if (x >= 0 && x < W && y >= 0 && y < H) { nb = row_of[b * P + y * W + x]; t.tapmask[i / 16] |= (uint16_t)(1u << tap); }
        }
        t.nbr[tap * RPAD + i] = (uint16_t)nb;
      }
    }
    t.cost = 0;
    for (int i = 0; i < NTILES; ++i) for (int tap = 0; tap < 9; ++tap) t.cost += (t.tapmask[i] >> tap) & 1;
    return t;
  }
  static constexpr Tab best() {
    int bo = 0, bp = 4, bc = cost_of(0, 4);
    for (int oc = 0; oc < 24; ++oc)
      for (int pp = 0; pp < 5; ++pp) {
        const int c = cost_of(oc, pp);
        if (c < bc) { bc = c; bo = oc; bp = pp; }
      }
    return build(bo, bp);
  }
  static constexpr Tab tab = best();
};

// NT = row tiles per workgroup: 11 (throughput: 4 Connect-Four boards, 2 workgroups per CU at 64 filters) or 3
// (latency: ONE Connect-Four board per workgroup, so a small batch spreads over 4x as many CUs and the
// sequential layer chain of a workgroup is 3.7x shorter -- the reference's 128-worker configurations).
template <class Gm, int F = 64, int NT = 11> struct T16 {
  static constexpr int NTILE = NT, RPAD = NTILE * 16;
  static constexpr int TB = RPAD / Gm::P;            // NT = 11: 4 Connect-Four boards, 19 Tic-tac-toe, 12 Mancala; NT = 3: 1, 5, 3
  static constexpr int ROWS = TB * Gm::P;
  static constexpr int STRIDE = F + 8;               // rows of 4 d dwords with d = 18 / 34: see posF
  static constexpr int BUF = (RPAD + GEO_NZ) * STRIDE;   // rows RPAD .. RPAD + 7 = zeros
  static constexpr int PLANES = (RPAD + GEO_NZ) * Gm::C;
  static constexpr int TABLE = (10 * RPAD + 1) / 2;  // floats holding nbr [9][RPAD] + pos [RPAD] as u16 (Geo16)
  static constexpr int BYTES = (BUF + PLANES + TABLE) * 4;   // 57 KB at F = 64 (2 workgroups per CU), 102 KB at F = 128 (1)
  static constexpr int WAVES = F / 16, THREADS = 64 * WAVES;
  static constexpr int CT = F / 16;                  // channel tiles of 16 = wavefronts per row-tile set
  static constexpr int KH = F / 64;                  // 64-channel halves of a tap (one pipeline step each)
  static constexpr int SQ = F / 16;                  // float4 of B per tap and lane
  using Game = Gm;
  using Geo = Geo16<Gm, NTILE, TB>;
  static constexpr int FILT = F;
};
// Paired geometry (k_tower16x2, 64 filters): TWO sets of F/16 wavefronts share one LDS buffer of 21 row tiles =
// 336 rows = exactly 8 Connect-Four boards (24 Mancala, 37 Tic-tac-toe).  Set 0 owns tiles 0..10, set 1 tiles
// 11..20: 21 tile-units per 8 boards instead of the 22 (2 x 11, 8 padding rows per 4 boards) of two k_tower16
// workgroups -- the same two wavefronts per SIMD, 4.5 % fewer MFMAs per board.  92 KB of LDS, one workgroup per CU.
// (r6) NT0_ / NT1_ = 10 / 9: 19 row tiles = 304 rows = 7 Connect-Four boards -- the form of a launch's SECOND round of workgroups when
// the batch is between 15 and 16 boards per CU (a free-running wave's 3700-3840 boards: 256 workgroups of 8 + up to 256 of 7 instead
// of two rounds of 8, the second 86 % full: net_impl.h wave_net_f).
template <class Gm, int F = 64, int NT0_ = 11, int NT1_ = 10> struct T16P {
  static constexpr int NT0 = NT0_, NT1 = NT1_, NTW = NT0 + NT1, RPAD = NTW * 16;
  static constexpr int TB = RPAD / Gm::P;
  static constexpr int ROWS = TB * Gm::P;
  static constexpr int STRIDE = F + 8;
  static constexpr int BUF = (RPAD + GEO_NZ) * STRIDE;
  static constexpr int PLANES = (RPAD + GEO_NZ) * Gm::C;
  static constexpr int TABLE = (10 * RPAD + 1) / 2;
  static constexpr int BYTES = (BUF + PLANES + TABLE) * 4;
  static constexpr int CT = F / 16, WAVES = 2 * CT, THREADS = 64 * WAVES;
  static constexpr int KH = F / 64, SQ = F / 16;
  using Game = Gm;
  using Geo = Geo16<Gm, NTW, TB>;
  static constexpr int FILT = F;
};

// Position of channel c inside a buffer row.  MFMA step (kh, q, j) of a tap feeds k group G = lane >> 4 the channel with
// G(c) = (c >= F/2) + 2 (c & 1) and w(c) = (c mod F/2) >> 1 = 16 kh + 4 q + j (the contract's order j, F/2 + j); the lane
// reads its four values of step (kh, q) as ONE float4 at position 64 kh + 16 q + 4 G.  The four k groups of a read are
// thus 16 bytes apart and the rows 4 (F + 8) bytes: a ds_read_b128 pass covers the lanes (lrow 0..7, G and G + 1), and
// with rows of 4 d dwords it reaches all 64 banks iff { d lrow + G } is distinct mod 16 -- d = 18 (F = 64) or 34 (F = 128).
// (The first layout -- k groups F bytes apart, rows of F + 4 floats -- read at half the LDS rate, 8 cycles per wave
// instruction instead of 4: tools/probes/lds_conflict.hip.)
template <int F> __device__ __forceinline__ int posF(int c) {
  const int G = (c >= F / 2 ? 1 : 0) + 2 * (c & 1), w = (c % (F / 2)) >> 1;
  return (w >> 4) * 64 + ((w >> 2) & 3) * 16 + G * 4 + (w & 3);
}

// One F -> F convolution (conv16p below) is a fully unrolled sequence of steps (tap, 64-channel half, tile pair); the A
// rows of step k+1 are read from LDS and (at the first step of a tap) the B fragments of the next tap requested from L2
// while the MFMAs of step k issue.  Tiles go in pairs so that consecutive MFMAs never hit the same accumulator
// (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency); an odd tile rides alone.  sched_barrier(0)
// after every step keeps hipcc from hoisting further loads, which bounds the live A registers to two pairs (the kernel
// must stay near 200 VGPRs so that the tree kernels of the other slot group can co-reside on the SIMD).
static constexpr int T16_GMAX = 2;

// Compile-time list of the pipeline steps of one convolution for a wavefront that owns tiles TILE0 .. TILE0 + NT - 1:
// (tap, 64-channel half, tile pair).  Tiles go in pairs so that consecutive MFMAs never hit the same accumulator.
template <class G, int NT, int TILE0, int KH, int NTAP> struct Steps16 {
  static constexpr int MAXS = 9 * KH * ((NT + 1) / 2);
  struct L {
    int n;
    int8_t tap[MAXS], kh[MAXS], t0[MAXS], t1[MAXS];   // t1 = -1: single tile
    int8_t first[MAXS];                                 // first step of its tap: the next tap's weights are requested here
    int8_t next_tap[MAXS];                              // tap whose weights to request (-1: none)
    int8_t ord[MAXS];                                   // ordinal of the tap among the taps that have steps (weight stage = ord & 1)
    int8_t first_tap;
  };
  static constexpr L make() {
    L l{};
    l.n = 0; l.first_tap = -1;
    int taps[9] = {}, ntaps = 0;
    for (int tap = 0; tap < 9; ++tap) {
      if (NTAP == 1 && tap != 4) continue;
      bool any = false;
      for (int t = 0; t < NT; ++t) any = any || ((G::tab.tapmask[TILE0 + t] >> tap) & 1);
      if (any) taps[ntaps++] = tap;
    }
    for (int ti = 0; ti < ntaps; ++ti) {
      const int tap = taps[ti];
      if (l.first_tap < 0) l.first_tap = (int8_t)tap;
      int act[NT] = {}, na = 0;
      for (int t = 0; t < NT; ++t) if ((G::tab.tapmask[TILE0 + t] >> tap) & 1) act[na++] = t;
      for (int kh = 0; kh < KH; ++kh)
        for (int i = 0; i < na; i += 2) {
          const int k = l.n++;
          l.tap[k] = (int8_t)tap; l.kh[k] = (int8_t)kh; l.t0[k] = (int8_t)act[i]; l.t1[k] = (int8_t)(i + 1 < na ? act[i + 1] : -1);
          l.first[k] = (kh == 0 && i == 0);
          l.next_tap[k] = (int8_t)(ti + 1 < ntaps ? taps[ti + 1] : -1);
          l.ord[k] = (int8_t)ti;
        }
    }
    return l;
  }
  static constexpr L list = make();
};

// activation rows of one step: 4 float4 (16 channels of this lane's k group) per tile; `off` = float offsets of the two
// tap-shifted rows in the buffer (from the nbr table)
template <class T>
__device__ __forceinline__ void load_rows16p(const float* __restrict__ buf, int off0, int off1, bool two, int kh, int g, float4 (&a)[T16_GMAX][4]) {
  const float* p0 = buf + off0 + kh * 64 + g * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) a[0][q] = *(const float4*)(p0 + q * 16);
  if (two) {
    const float* p1 = buf + off1 + kh * 64 + g * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[1][q] = *(const float4*)(p1 + q * 16);
  }
}
template <int T0, int T1, int KHI, int SQ, int NT>
__device__ __forceinline__ void mfma_tiles16(const float4 (&a)[T16_GMAX][4], const float4 (&bfull)[SQ], f32x4v (&acc)[NT]) {
  const float4* b = bfull + KHI * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc[T0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][q].x, b[q].x, acc[T0], 0, 0, 0);
    if constexpr (T1 >= 0) acc[T1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][q].x, b[q].x, acc[T1], 0, 0, 0);
    acc[T0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][q].y, b[q].y, acc[T0], 0, 0, 0);
    if constexpr (T1 >= 0) acc[T1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][q].y, b[q].y, acc[T1], 0, 0, 0);
    acc[T0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][q].z, b[q].z, acc[T0], 0, 0, 0);
    if constexpr (T1 >= 0) acc[T1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][q].z, b[q].z, acc[T1], 0, 0, 0);
    acc[T0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][q].w, b[q].w, acc[T0], 0, 0, 0);
    if constexpr (T1 >= 0) acc[T1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][q].w, b[q].w, acc[T1], 0, 0, 0);
  }
}
// look-up of the tap-shifted rows of step K for this lane -> float offsets row * STRIDE
template <class T, class SL, int K, int TILE0>
__device__ __forceinline__ void load_idx16p(const uint16_t* __restrict__ nbr, int lrow, int (&idx)[2]) {
  if constexpr (K < SL::list.n) {
    constexpr int tap = SL::list.tap[K], t0 = SL::list.t0[K], t1 = SL::list.t1[K];
    idx[0] = (int)nbr[tap * T::RPAD + (TILE0 + t0) * 16 + lrow] * T::STRIDE;
    if constexpr (t1 >= 0) idx[1] = (int)nbr[tap * T::RPAD + (TILE0 + t1) * 16 + lrow] * T::STRIDE;
  }
}
template <class T, class SL, int NT, int TILE0, int K>
__device__ __forceinline__ void conv16p_steps(const float* __restrict__ buf, const uint16_t* __restrict__ nbr, const float4* __restrict__ wl,
                                              f32x4v (&acc)[NT], int lrow, int g, float4 (&b0)[T::FILT / 16], float4 (&b1)[T::FILT / 16],
                                              float4 (&aA)[T16_GMAX][4], float4 (&aB)[T16_GMAX][4], int (&idx)[2]) {
  constexpr int F = T::FILT;
  if constexpr (K < SL::list.n) {
    constexpr int kh = SL::list.kh[K], t0 = SL::list.t0[K], t1 = SL::list.t1[K], ord = SL::list.ord[K];
    float4 (&cur)[T16_GMAX][4] = (K & 1) ? aB : aA;
    float4 (&nxt)[T16_GMAX][4] = (K & 1) ? aA : aB;
    float4 (&bc)[F / 16] = (ord & 1) ? b1 : b0;
    float4 (&bn)[F / 16] = (ord & 1) ? b0 : b1;
    if constexpr (SL::list.first[K] && SL::list.next_tap[K] >= 0) {
      constexpr int ntap = SL::list.next_tap[K];
#pragma unroll
      for (int q = 0; q < T::SQ; ++q) bn[q] = wl[(size_t)(ntap * T::CT * T::SQ + q) * 64];
    }
    if constexpr (K + 1 < SL::list.n) {
      // rows of step K + 1 (their offsets were looked up during step K - 1), then the offsets of step K + 2
      load_rows16p<T>(buf, idx[0], idx[1], SL::list.t1[K + 1] >= 0, SL::list.kh[K + 1], g, nxt);
      load_idx16p<T, SL, K + 2, TILE0>(nbr, lrow, idx);
    }
    mfma_tiles16<t0, t1, kh, F / 16, NT>(cur, bc, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv16p_steps<T, SL, NT, TILE0, K + 1>(buf, nbr, wl, acc, lrow, g, b0, b1, aA, aB, idx);
  }
}
// One F -> F convolution (NTAP = 9: 3x3, NTAP = 1: 1x1) into the NT accumulators of this wave, permuted rows.
// (Requesting the first tap's weight fragments before the previous layer's epilogue and barriers was tried: the kernel
// alone is unchanged, 0.844 ms per 4096 boards, and two slot groups lose 2 % -- 0.857 vs 0.838 ms per step.)
template <class T, class G, int NT, int TILE0, int NTAP>
__device__ __forceinline__ void conv16p(const float* __restrict__ buf, const uint16_t* __restrict__ nbr, const float4* __restrict__ wl,
                                        f32x4v (&acc)[NT], int lrow, int g) {
  constexpr int F = T::FILT;
  using SL = Steps16<G, NT, TILE0, T::KH, NTAP>;
  if constexpr (SL::list.n > 0) {
    float4 b0[F / 16], b1[F / 16], aA[T16_GMAX][4], aB[T16_GMAX][4];
    int idx[2] = {0, 0};
    constexpr int tap0 = NTAP == 1 ? 0 : SL::list.first_tap;       // a 1x1 convolution has one weight tap (its rows are tap 4, the centre)
#pragma unroll
    for (int q = 0; q < F / 16; ++q) b0[q] = wl[(size_t)(tap0 * T::CT * T::SQ + q) * 64];
    load_idx16p<T, SL, 0, TILE0>(nbr, lrow, idx);
    load_rows16p<T>(buf, idx[0], idx[1], SL::list.t1[0] >= 0, SL::list.kh[0], g, aA);
    load_idx16p<T, SL, 1, TILE0>(nbr, lrow, idx);
    __builtin_amdgcn_sched_barrier(0);
    conv16p_steps<T, SL, NT, TILE0, 0>(buf, nbr, wl, acc, lrow, g, b0, b1, aA, aB, idx);
  }
}

template <int F> struct T16Threads { static constexpr int V = 64 * (F / 16); };
// row tiles of the latency variant: 3 (one Connect-Four board, 5 Tic-tac-toe, 3 Mancala), or what one board needs (9x9: 6)
template <class Gm> constexpr int NTS = Gm::P <= 48 ? 3 : (Gm::P + 15) / 16;
// (r4) row tiles of the EXACT-FIT variant: the smallest tile count (4 .. 10) whose rows are a whole number of boards -- no
// padding rows and, for the slot counts of the BASELINE configurations, a whole number of workgroups per CU: Mancala 7 tiles =
// 112 rows = 8 boards (8192 slots = 1024 workgroups = 4 per CU; the 11-tile form: 12 boards of 176 rows, 683 workgroups = 2.67
// per CU), Tic-tac-toe 9 tiles = 16 boards.  0 = the game has none (Connect-Four's is the paired 21-tile kernel).
// (r5) A game without one (Connect-Four: 42 positions) gets the HALF-SIZE form instead -- 6 tiles = 96 rows = 2 boards + 12 padding
// rows: since the evaluation cache answers part of every wave, a launch holds 1000 ... 1600 boards instead of 2048 / 4096, and what
// a launch costs is set by the CU with the most boards: with 4-board workgroups that is 8 boards wherever 256 CUs share more than
// 1024, with 2-board workgroups 6 (pick_tower prices both with the launch size the device reported for the last waves).
template <class Gm> constexpr int ntm_of() { for (int nt = 4; nt <= 10; ++nt) if ((16 * nt) % Gm::P == 0) return nt; return Gm::P == 42 ? 6 : 0; }
template <class Gm> constexpr int NTM = ntm_of<Gm>();

// What a workgroup does after it has written a layer's outputs (its own channels) into the activation buffer.
// NoXch: the workgroup owns every channel -- a barrier.
struct NoXch {
  static constexpr int STEM_HALVES = 1;
  static constexpr bool EARLY_PUBLISH = false;
};
// PairXch (k_tower16s): two workgroups share a board tile, each computes one half of the output channels of every
// residual layer and needs the other half before the next convolution.  Every lane publishes its NT x 4 outputs, in
// fragment order, as 64-bit words (value, tag) to the workgroup's area in HBM -- two areas, by layer parity; tag =
// launch epoch x 256 + layer -- and then polls the partner lane's words until they carry the tag: no flag, no
// store-completion wait, one store -> load latency per layer.  Agent-scope relaxed atomics (write-through stores,
// cache-bypassing loads; 8-byte single-copy atomicity keeps value and tag together): no cache-wide write-back or
// invalidate.  An area is rewritten two layers later, which the partner can only reach after it has read this layer.
// (tools/probes/xcd_exchange.hip: a one-way hand-over with sc1 stores and sc1 loads takes ~1000 cycles inside an XCD and
// ~1270 across XCDs; sc0 or plain loads, also RMW atomics at workgroup scope, keep returning the CU's cached copy and
// never see the partner's word, even on the same XCD -- so pairs are simply (b, b ^ 1).)
// (Waiting on one word per wavefront before reading all twelve cost a round trip more than it saved: 172 vs 167 us.)
// The poll is bounded IN TIME (50 ms of the 100 MHz constant clock -- round 5; rounds 3-4: 2 s -- looked at every 256 polls): a partner that never
// arrives raises DERR_EXCHANGE instead of hanging the GPU.  Workgroups of a launch are dispatched in order and a pair is
// (b, b ^ 1), so at any moment at most ONE pair of a launch is half resident and every other resident workgroup of it
// belongs to a complete pair, which finishes and frees its CU: a pair cannot deadlock, it can only wait for CUs that
// other work holds (another engine's tower in an arena, a trainer, another process) -- for as long as that work runs,
// which is why the bound is wall time (round 2 counted 2^20 polls, ~0.7 s alone but arbitrarily little under contention).  A co-resident
// partner answers in ~1 us and the longest kernel this library puts beside a split tower runs ~1 ms, so 50 ms is still four orders
// of magnitude of slack; waiting longer only delays the fallback (recover_split, azhip.hip), which costs nothing but the split.
enum { DERR_EXCHANGE = 6 };
static constexpr unsigned long long XCH_WAIT_TICKS = 5000000ull;
struct PairXch {
  static constexpr int STEM_HALVES = 2;              // the stem is cheap: both halves computed locally, one exchange less
  unsigned long long* mine;          // [2][NT * 4][THREADS]
  const unsigned long long* theirs;
  uint32_t tag0;                     // launch epoch << 8
  int* err;
  int* hflag;                        // host-visible copy of "an exchange gave up" (the host looks at it before every launch, no sync)
  int pch;                           // the partner lane's output channel
  static constexpr bool EARLY_PUBLISH = true;        // outputs leave for the partner before the workgroup's own barrier
  template <class T, int NT>
  __device__ __forceinline__ void publish(int step, const float (&ov)[NT][4], int tid) const {
    const uint32_t tag = tag0 + (uint32_t)step;
    unsigned long long* m = mine + (size_t)(step & 1) * NT * 4 * T::THREADS + tid;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __hip_atomic_store(m + (t * 4 + i) * T::THREADS, ((unsigned long long)tag << 32) | __float_as_uint(ov[t][i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  template <class T, int NT, int R0>
  __device__ __forceinline__ void collect(int step, float* __restrict__ buf, int g, int lrow, int tid) const {
    constexpr int THREADS = T::THREADS, F = T::FILT;
    const uint32_t tag = tag0 + (uint32_t)step;
    const unsigned long long* th = theirs + (size_t)(step & 1) * NT * 4 * THREADS + tid;
    unsigned long long pv[NT][4];
    int spins = 0;
    unsigned long long t_start = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pv[t][i] = __hip_atomic_load(th + (t * 4 + i) * THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (uint32_t)(pv[t][i] >> 32) == tag;
        }
      if (ok) break;
      if ((++spins & 255) == 0) {
        const unsigned long long now = wall_clock64();
        if (!t_start) t_start = now;
        else if (now - t_start > XCH_WAIT_TICKS) {
          atomicCAS(err, 0, (int)DERR_EXCHANGE);
          if (hflag) __hip_atomic_store(hflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    const int ppos = posF<F>(pch);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) buf[(R0 + t * 16 + g * 4 + i) * T::STRIDE + ppos] = __uint_as_float((uint32_t)pv[t][i]);
    __syncthreads();
  }
};

// One wavefront's share of the tower: its NT row tiles start at row tile TILE0 of the workgroup's LDS buffer (rows in
// Geo16's permuted order), it owns the 16 output channels of channel tile cw.  The caller has filled `planes` and the
// tables, zeroed the buffer's zero row and synchronised; every wavefront of the workgroup runs the same number of barriers.
template <class T, bool FROM_PLANES, int NT, int TILE0, class X = NoXch>
__device__ __forceinline__ void tower16_wave(const Net16Dev& net, float* __restrict__ buf, const float* __restrict__ planes,
                                             const uint16_t* __restrict__ nbr, const uint16_t* __restrict__ pos, int cw,
                                             int lane, int n, int board0, float* __restrict__ hfeat, const X& xch = X{}) {
  using Gm = typename T::Game;
  using G = typename T::Geo;
  constexpr int F = T::FILT, P = Gm::P, C = Gm::C, TB = T::TB, STRIDE = T::STRIDE, R0 = TILE0 * 16;
  const int lrow = lane & 15, g = lane >> 4;
  unsigned long long* dbg = (net.dbg && threadIdx.x == 0) ? net.dbg + (size_t)blockIdx.x * 8 : nullptr;
#define AZ_STAMP16(i) do { if (dbg) dbg[i] = __builtin_readcyclecounter(); } while (0)
  AZ_STAMP16(0);
  // this lane's output element (tile, i): row = R0 + tile*16 + g*4 + i, channel = cw*16 + lrow
  const int ch = cw * 16 + lrow;
  const int opos = posF<F>(ch);

  f32x4v acc[NT];
  float ov[NT][4];                                   // this lane's outputs of the layer just finished (what a partner workgroup needs)
  float xres[NT][4];                                 // the current block's input at this lane's elements (skip connection): the lane wrote them itself
  // ---- stem: Conv(3x3, C => F) + BN + ReLU, K = 9C padded to a multiple of 4 (every tap: its k order mixes them) ----
  // (a pair of workgroups, X::STEM_HALVES = 2: this wavefront also computes the partner's channel tile -- cheaper than an exchange)
#pragma unroll
  for (int hh = 0; hh < X::STEM_HALVES; ++hh) {
    constexpr int KK = 9 * C, K2 = (KK + 1) / 2, NS = (2 * K2 + 3) / 4;
    const int cws = X::STEM_HALVES == 1 ? cw : hh * (F / 32) + cw % (F / 32);
    const int chs = cws * 16 + lrow, oposs = posF<F>(chs);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // sequence position p = 4s + g -> k = (p & 1) * K2 + (p >> 1)   (the paired order of the contract)
      const float bw = net.stem_w[(size_t)(cws * NS + s) * 64 + lane];
      const int p = 4 * s + g;
      const int k = (p & 1) * K2 + (p >> 1);
      const bool kin = k < KK && p < 2 * K2;
      const int tap = kin ? k / C : 4, c = kin ? k % C : 0;
#pragma unroll
      for (int tile = 0; tile < NT; ++tile) {
        const int row = kin ? (int)nbr[tap * T::RPAD + R0 + tile * 16 + lrow] : T::RPAD;
        const float a = planes[row * C + c];
        acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[tile], 0, 0, 0);
      }
    }
    const float sc = net.stem_ss[chs], sh = net.stem_ss[F + chs];
#pragma unroll
    for (int tile = 0; tile < NT; ++tile)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = az_fmaf(acc[tile][i], sc, sh);
        const float r = v > 0.0f ? v : 0.0f;
        buf[(R0 + tile * 16 + g * 4 + i) * STRIDE + oposs] = r;
        if (cws == cw) xres[tile][i] = r;            // this lane's own element: the first block's input
      }
  }
  __syncthreads();
  AZ_STAMP16(1);

  // ---- residual tower ---------------------------------------------------------------------------------
  const size_t LAYER_W = (size_t)9 * T::CT * T::SQ * 64;          // float4 per layer
  for (int layer = 0; layer < 2 * net.nblocks; ++layer) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // opaque copy: stops hipcc from hoisting the loop-invariant (tap, tile) table look-ups out of the layer loop,
    // which would cost ~100 VGPRs for the whole kernel
    int lrow_l = lrow;
    asm volatile("" : "+v"(lrow_l));
    conv16p<T, G, NT, TILE0, 9>(buf, nbr, net.conv_w + (size_t)layer * LAYER_W + (size_t)cw * T::SQ * 64 + lane, acc, lrow_l, g);
    const float sc = net.conv_ss[(size_t)layer * 2 * F + ch], sh = net.conv_ss[(size_t)layer * 2 * F + F + ch];
    if (layer == 2) AZ_STAMP16(4);
    if constexpr (X::EARLY_PUBLISH) {
      // a pair of workgroups: the outputs (they need only this lane's own elements of the buffer) go to the partner
      // first, then the barrier, the buffer update and the partner's half
#pragma unroll
      for (int tile = 0; tile < NT; ++tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = az_fmaf(acc[tile][i], sc, sh);
          if (layer & 1) v = v + xres[tile][i];
          ov[tile][i] = v > 0.0f ? v : 0.0f;
          if (layer & 1) xres[tile][i] = ov[tile][i];                // the next block's input
        }
      xch.template publish<T, NT>(layer, ov, threadIdx.x);
      __builtin_amdgcn_s_setprio(2);
      __syncthreads();                               // every wave has finished reading the buffer
      if (layer == 2) AZ_STAMP16(5);
#pragma unroll
      for (int tile = 0; tile < NT; ++tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[(R0 + tile * 16 + g * 4 + i) * STRIDE + opos] = ov[tile][i];
      xch.template collect<T, NT, R0>(layer, buf, g, lrow, threadIdx.x);
    } else {
    __builtin_amdgcn_s_setprio(2);
    __syncthreads();                                 // every wave has finished reading the buffer
    if (layer == 2) AZ_STAMP16(5);
    if (!(layer & 1)) {
#pragma unroll
      for (int tile = 0; tile < NT; ++tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = (R0 + tile * 16 + g * 4 + i) * STRIDE + opos;
          const float v = az_fmaf(acc[tile][i], sc, sh);
          buf[a] = v > 0.0f ? v : 0.0f;
        }
    } else {
#pragma unroll
      for (int tile = 0; tile < NT; ++tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = (R0 + tile * 16 + g * 4 + i) * STRIDE + opos;
          float v = az_fmaf(acc[tile][i], sc, sh);
          v = v + xres[tile][i];
          xres[tile][i] = v > 0.0f ? v : 0.0f;       // block output = the next block's input
          buf[a] = xres[tile][i];
        }
    }
    __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    if (layer == 1 || layer == 2) AZ_STAMP16(5 + layer);           // 6: layer 2 starts, 7: layer 2 done
  }
  AZ_STAMP16(2);
  // ---- both 1x1 head convolutions + BN + ReLU as one F => F GEMM ------------------------------------
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  conv16p<T, G, NT, TILE0, 1>(buf, nbr, net.head_w + (size_t)cw * T::SQ * 64 + lane, acc, lrow, g);
  {
    const float sc = net.head_ss[ch], sh = net.head_ss[F + ch];
    // head features straight from the accumulators to HBM / L2, [board][P][F] in natural position and channel order.
    // (The dense heads stay a separate launch with 32-board MFMA tiles: computing them here per workgroup -- tried on
    // MFMA and on the vector ALU, from LDS-resident features -- makes each of the 1024 workgroups stream the 382 KB of
    // dense weights through L2, 65 us per 4096-board launch against 41 us for k_heads_mfma; handing a 32-board tile to
    // the last of 8 workgroups needs device-scope fences across XCDs and cost 190 us.  DESIGN.md §4.)
    const int nvalid = ((n - board0) < TB ? (n - board0) : TB) * P;
#pragma unroll
    for (int tile = 0; tile < NT; ++tile)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ps = pos[R0 + tile * 16 + g * 4 + i];             // board * P + position of this buffer row (0xffff: padding)
        const float v = az_fmaf(acc[tile][i], sc, sh);
        if (ps < nvalid) hfeat[((size_t)board0 * P + ps) * F + ch] = v > 0.0f ? v : 0.0f;
      }
  }
  AZ_STAMP16(3);
#undef AZ_STAMP16
}

// input planes [RPAD + 1][C] in the permuted row order, the zero row of the activation buffer and the LDS copies of the
// permutation tables, by all threads of the workgroup
template <class T, bool FROM_PLANES>
__device__ __forceinline__ void tower16_fill(float* __restrict__ buf, float* __restrict__ planes, uint16_t* __restrict__ nbr,
                                             uint16_t* __restrict__ pos, const uint16_t* __restrict__ geo, const GEnv* __restrict__ leaf_env,
                                             const int* __restrict__ eval_slots, const float* __restrict__ X, int n, int board0, int tid) {
  using Gm = typename T::Game;
  constexpr int P = Gm::P, C = Gm::C;
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
  for (int i = tid; i < GEO_NZ * T::STRIDE; i += T::THREADS) buf[T::RPAD * T::STRIDE + i] = 0.0f;
  for (int i = tid; i < T::RPAD; i += T::THREADS) pos[i] = geo[i];
  for (int i = tid; i < 9 * T::RPAD; i += T::THREADS) nbr[i] = geo[T::RPAD + i];
}

template <class Gm, int F, bool FROM_PLANES, int NT = 11>
__global__ void __launch_bounds__(T16Threads<F>::V, 2)   // (three workgroups per CU for the LDS-light exact-fit variants: measured, no gain: 28.6 vs 28.9 M sims/s on Mancala)
k_tower16(Net16Dev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
          const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat) {
  using T = T16<Gm, F, NT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf = lds;
  float* planes = lds + T::BUF;
  uint16_t* nbr = (uint16_t*)(planes + T::PLANES);
  uint16_t* pos = nbr + 9 * T::RPAD;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  const int board0 = blockIdx.x * T::TB;
  if (board0 >= n) return;
  tower16_fill<T, FROM_PLANES>(buf, planes, nbr, pos, net.geo[NT == 11 ? 0 : NT == NTS<Gm> ? 1 : 3], leaf_env, eval_slots, X, n, board0, threadIdx.x);
  __syncthreads();
  tower16_wave<T, FROM_PLANES, NT, 0>(net, buf, planes, nbr, pos, threadIdx.x >> 6, threadIdx.x & 63, n, board0, hfeat);
}

// The paired form (T16P): wavefronts 0..CT-1 run tiles 0..10, wavefronts CT..2CT-1 tiles 11..20 of ONE buffer.
// one workgroup of the paired form on the boards board0 .. board0 + T::TB - 1
template <class T, bool FROM_PLANES>
__device__ __forceinline__ void tower16x2_body(const Net16Dev& net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots, int n,
                                               const float* __restrict__ X, float* __restrict__ hfeat, int board0, const uint16_t* __restrict__ geo) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf = lds;
  float* planes = lds + T::BUF;
  uint16_t* nbr = (uint16_t*)(planes + T::PLANES);
  uint16_t* pos = nbr + 9 * T::RPAD;
  tower16_fill<T, FROM_PLANES>(buf, planes, nbr, pos, geo, leaf_env, eval_slots, X, n, board0, threadIdx.x);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < T::CT) tower16_wave<T, FROM_PLANES, T::NT0, 0>(net, buf, planes, nbr, pos, wave, lane, n, board0, hfeat);
  else tower16_wave<T, FROM_PLANES, T::NT1, T::NT0>(net, buf, planes, nbr, pos, wave - T::CT, lane, n, board0, hfeat);
}
// base = index of the launch's first board
template <class Gm, int F, bool FROM_PLANES, int NT0 = 11, int NT1 = 10>
__global__ void __launch_bounds__(2 * T16Threads<F>::V, 1)
k_tower16x2(Net16Dev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
            const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat, int base) {
  using T = T16P<Gm, F, NT0, NT1>;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  const int board0 = base + blockIdx.x * T::TB;
  if (board0 >= n) return;
  tower16x2_body<T, FROM_PLANES>(net, leaf_env, eval_slots, n, X, hfeat, board0, net.geo[NT0 == 11 ? 2 : 4]);
}
// (r6) The 8-board form within 176 registers per lane (the compiler takes 198 when it may: 92 B of scratch per lane here, +2 % per launch).
// A free-running phase runs k_tree's background launch UNDER the tower (azhip.hip wave_group), and a k_tree wavefront (152 registers)
// shares a SIMD with this kernel's two wavefronts only if they take 2 x 176 of its 512, not 2 x 200: beside the 198-register form
// every CU that holds a background workgroup is lost to the tower for as long as the background search runs.  With few such CUs that
// costs nothing (476 workgroups on 246 CUs are two rounds like on 256); when a third of the slots search in the background -- the
// first seconds of a phase, every game in its opening and the evaluation cache answering -- it costs a third round: 1.04 ms instead
// of 0.85 for 3930 boards (profiles/r6/README.md §12).  wave_net_f picks this form for the waves where the counts the device reports
// say so.  amdgpu_num_vgpr counts half of the unified file on gfx90a and later: 88 = 176 registers.
template <class Gm, int F, bool FROM_PLANES>
__global__ void __launch_bounds__(2 * T16Threads<F>::V, 1) __attribute__((amdgpu_num_vgpr(88)))
k_tower16x2c(Net16Dev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
             const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat, int base) {
  using T = T16P<Gm, F, 11, 10>;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  const int board0 = base + blockIdx.x * T::TB;
  if (board0 >= n) return;
  tower16x2_body<T, FROM_PLANES>(net, leaf_env, eval_slots, n, X, hfeat, board0, net.geo[2]);
}
// (r6) Both paired forms in ONE launch: workgroups 0 .. first - 1 take 8 boards each (21 row tiles), the workgroups behind them 7 (19 tiles).
// A batch between 15 and 16 boards per CU -- a free-running wave's 3700-3840 boards on 256 CUs -- is two rounds of workgroups either
// way; with first = the number of CUs the second round is an eighth lighter, and because it is one launch a CU that is done early
// simply takes the next workgroup (two separate launches each last as long as their slowest workgroup: resnet16.h T16P, profiles/r6).
template <class Gm, int F, bool FROM_PLANES>
__global__ void __launch_bounds__(2 * T16Threads<F>::V, 1)
k_tower16x2m(Net16Dev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
             const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat, int first) {
  using T8 = T16P<Gm, F, 11, 10>;
  using T7 = T16P<Gm, F, 10, 9>;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  if ((int)blockIdx.x < first) {
    const int board0 = blockIdx.x * T8::TB;
    if (board0 >= n) return;
    tower16x2_body<T8, FROM_PLANES>(net, leaf_env, eval_slots, n, X, hfeat, board0, net.geo[2]);
  } else {
    const int board0 = first * T8::TB + ((int)blockIdx.x - first) * T7::TB;
    if (board0 >= n) return;
    tower16x2_body<T7, FROM_PLANES>(net, leaf_env, eval_slots, n, X, hfeat, board0, net.geo[4]);
  }
}

// Split form for launches that leave most CUs idle (k_tower16s, 128 filters): TWO workgroups per board tile, each with
// the whole activation buffer in LDS but only HALF of the output channels to compute, exchanging halves after every
// layer (PairXch).  A workgroup's sequential layer chain carries half of the MFMA work of k_tower16<.., NT = 3> (one
// Connect-Four board: 277 us on one CU, the floor of that variant).  Four wavefronts, one per SIMD, each with all row
// tiles of one 16-channel tile -- the arrangement of the 64-filter k_tower16<.., NT = 3>, which keeps its MFMA pipe 95 %
// busy.  (Twelve wavefronts with one row tile each tripled the weight stream per MFMA: 74 % busy in the convolutions,
// a 17 k-cycle barrier wait per layer, 183 us per launch instead of 167.)
template <class Gm, int F> struct T16S : T16<Gm, F, NTS<Gm>> {
  static constexpr int SPLIT = 2, TPW = NTS<Gm>, CWL = F / 16 / SPLIT;      // row tiles per wavefront, channel tiles per workgroup
  static constexpr int WAVES = CWL, THREADS = 64 * WAVES;
  static constexpr int XCH_WORDS = 2 * TPW * 4 * THREADS;        // publish area of one workgroup (64-bit words)
};
template <class Gm, int F, bool FROM_PLANES>
__global__ void __launch_bounds__((T16S<Gm, F>::THREADS), 1)
k_tower16s(Net16Dev net, const GEnv* __restrict__ leaf_env, const int* __restrict__ eval_slots,
           const int* __restrict__ n_eval_ptr, int n_fixed, const float* __restrict__ X, float* __restrict__ hfeat,
           unsigned long long* __restrict__ xch, unsigned long long epoch, int* __restrict__ err, int* __restrict__ hflag) {
  using T = T16S<Gm, F>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf = lds;
  float* planes = lds + T::BUF;
  uint16_t* nbr = (uint16_t*)(planes + T::PLANES);
  uint16_t* pos = nbr + 9 * T::RPAD;
  const int n = FROM_PLANES ? n_fixed : *n_eval_ptr;
  const int pair = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int board0 = pair * T::TB;
  if (board0 >= n) return;                           // both workgroups of the pair
  // an exchange of an EARLIER launch gave up: the host has not noticed yet and the results of every launch queued behind it
  // will be recomputed (azhip.hip recover_split) -- do not wait for partners again
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)DERR_EXCHANGE) return;
  if ((epoch >> 63) && blockIdx.x == 1) return;      // fault injection (az_debug_exchange_timeout): workgroup 0 loses its partner
  tower16_fill<T, FROM_PLANES>(buf, planes, nbr, pos, net.geo[1], leaf_env, eval_slots, X, n, board0, threadIdx.x);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cwl = wave;
  PairXch x;
  x.mine = xch + (size_t)blockIdx.x * T::XCH_WORDS;
  x.theirs = xch + (size_t)(blockIdx.x ^ 1) * T::XCH_WORDS;
  x.tag0 = (uint32_t)(epoch << 8);                   // + layer index < 256 (pick_tower)
  x.err = err;
  x.hflag = hflag;
  x.pch = ((half ^ 1) * T::CWL + cwl) * 16 + (lane & 15);
  tower16_wave<T, FROM_PLANES, T::TPW, 0, PairXch>(net, buf, planes, nbr, pos, half * T::CWL + cwl, lane, n, board0, hfeat, x);
}

// ---- pieces of the optimiser step that ride inside its kernels instead of in launches of their own (round 4) --------------------
// A step at 5x64 is ~170 dependent launches of 5 to 50 us: the ~5 us between two dependent kernels were a third of it.
// (a) TrFinal / tr_finish: the second stage of a column sum.  Every workgroup leaves its partial sums in part[workgroup][2][C];
//     the workgroup that finishes LAST (a counter) adds them in index order -- deterministic -- and does with a channel's two sums
//     what the mode says.  Was k_tr_colsum_final, one launch after every producer (29 per step).
struct TrFinal {
  int mode;                     // 0: sums only; 1: batch-norm statistics (forward); 2: dgamma / dbeta (backward); 3: out0 = sum0 (bias gradient)
  long long R; float momentum;
  const float* bias; float *mean, *invstd, *run_mean, *run_var;      // mode 1
  float *dgamma, *dbeta, *mf;                                         // mode 2 (mf[2][C]: the two sums / R as floats, what k_tr_bn_bwd subtracts)
  float* out0;                                                        // mode 3
  int* counter;                 // workgroups of the launch that have left their partials (back to 0 when the last one is done)
};
// The partials cross workgroups (and XCDs, whose L2s are not coherent with each other) inside one launch: they are written and
// read with agent-scope atomics (write-through stores, cache-bypassing loads -- the exchange idiom of k_tower16s).  Ordering
// (ADVICE r4): the counter's increment is a RELEASE at agent scope -- everything the workgroup's threads stored before the barrier
// happens-before it -- and the workgroup that sees the last count takes an ACQUIRE fence before it reads the partials, so the
// reduction is ordered by the HIP memory model, not by gfx9's write-through behaviour.  (The release writes the XCD's dirty L2
// lines back, which is part of why this form is SLOWER than a dependent launch and off by default: AZHIP_TRAIN_FINISH_INSIDE=1,
// tests/test_train_gpu.py holds it bit-identical to the separate launch.)
__device__ __forceinline__ void part_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double part_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// scratch: (blockDim.x + 2 C) doubles of LDS nobody else uses any more.  All threads of the workgroup call it, after part_store.
__device__ __forceinline__ void tr_finish(const double* __restrict__ part, int C, const TrFinal& f, double* __restrict__ scratch) {
  if (!f.counter) return;
  __shared__ int s_last;
  __builtin_amdgcn_s_waitcnt(0);                                     // this thread's partials have been performed at the device's coherence point
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(f.counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == (int)(gridDim.x * gridDim.y) - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // pairs with every other workgroup's release on the counter
  const int nparts = (int)gridDim.x, T = (int)blockDim.x, t = (int)threadIdx.x, npair = 2 * C;
  double* fin = scratch + T;                                         // [2][C] the finished sums
  if (npair <= T) {
    // pair (w, c) = t % npair; T / npair threads share it: partials k = sub, sub + nsub, ... then the subs in order
    const int nsub = T / npair, pair = t % npair, sub = t / npair, w = pair / C, c = pair % C;
    double acc = 0.0;
    if (sub < nsub) for (int k = sub; k < nparts; k += nsub) acc += part_load(part + ((size_t)k * 2 + w) * C + c);
    scratch[t] = acc;
    __syncthreads();
    if (sub == 0) { double v = scratch[pair]; for (int q = 1; q < nsub; ++q) v += scratch[pair + q * npair]; fin[pair] = v; }
  } else {
    for (int pair = t; pair < npair; pair += T) {
      const int w = pair / C, c = pair % C;
      double acc = 0.0;
      for (int k = 0; k < nparts; ++k) acc += part_load(part + ((size_t)k * 2 + w) * C + c);
      fin[pair] = acc;
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += T) {
    const double s0 = fin[c], s1 = fin[C + c];
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
  if (t == 0) __hip_atomic_store(f.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch on this counter starts from zero (stream order)
}
// (b) BnIn: the INPUT of a forward convolution computed on the fly from the previous layer's pre-activation,
//     a = relu(gamma * ((g - mean) * invstd) + beta (+ res)) -- exactly k_tr_bn_apply's operations -- while the rows go into LDS;
//     the workgroup also writes a (the backward pass needs it): boards do not share rows, so every element has one writer.
//     Was a pass of its own over the tensor between every two convolutions.
struct BnIn { const float *g, *mean, *invstd, *gamma, *beta, *res; float* a_out; };

// One 3x3 F -> F convolution as a stand-alone layer (HBM -> HBM), for the optimiser step (train.hip): forward
// g = conv(a) and data gradient da = conv_rot(dg) are the same kernel with different weight fragments.  A workgroup
// takes 4 Connect-Four boards (T16<Game, F, 11>): rows [R][F] in natural channel order go into the LDS buffer in the
// channel order conv16p expects and in Geo16's row order (`geo`: the engine's table of the 11-tile geometry), so the
// (tile, tap) products that fall off the board are skipped as in the tower (85 of 99); the 11 accumulator tiles of each
// wavefront come back out in natural order.  No bias, no activation: batch norm follows with batch statistics (the
// bias cancels there).
// STATS (the forward pass): the workgroup also leaves the column sums (sum v, sum v^2) of its rows of the output in
// part[blockIdx.x][2][F] (double), the first stage of the batch-norm statistics -- the accumulators are at hand, a
// separate pass over the output (k_tr_colsum<0>) is not needed.
// STAMP (tools/probes/conv_stamps.hip): clock readings of the first wavefront of every workgroup in stamps[workgroup][4]
// NT (r4): row tiles of a workgroup.  11 = 4 Connect-Four boards; 6 = 2 boards: at the reference's batch of 1024 the 11-tile form is
// 256 workgroups -- ONE per CU, whose fill, products and epilogue then run one after the other with nothing beside them -- the
// 6-tile form 512, two per CU.
template <class Gm, int F, bool STATS, bool STAMP = false, int NT_ = 11>
__global__ void __launch_bounds__(T16Threads<F>::V, 2)
k_conv16_layer(const float* __restrict__ in, const float4* __restrict__ wfrag, float* __restrict__ out, int nboards, const uint16_t* __restrict__ geo,
               double* __restrict__ part, const float* __restrict__ addend, long long* __restrict__ stamps, BnIn bn = BnIn{}, TrFinal fin = TrFinal{}) {
  long long ts[4] = {0, 0, 0, 0};
  if constexpr (STAMP) ts[0] = wall_clock64();
  using T = T16<Gm, F, NT_>;
  using G = typename T::Geo;
  constexpr int P = Gm::P, STRIDE = T::STRIDE, NT = NT_;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf = lds;
  uint16_t* nbr = (uint16_t*)(lds + T::BUF + T::PLANES);
  uint16_t* pos = nbr + 9 * T::RPAD;
  const int board0 = blockIdx.x * T::TB;
  if (board0 >= nboards) return;
  const int nb = (nboards - board0) < T::TB ? (nboards - board0) : T::TB;
  const int nvalid = nb * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* in4 = (const float4*)((bn.g ? bn.g : in) + (size_t)board0 * P * F);
  const float4* res4 = bn.res ? (const float4*)(bn.res + (size_t)board0 * P * F) : nullptr;
  float4* aout4 = bn.g ? (float4*)(bn.a_out + (size_t)board0 * P * F) : nullptr;
  for (int idx = tid; idx < (T::RPAD + GEO_NZ) * (F / 4); idx += T::THREADS) {
    const int row = idx / (F / 4), c4 = idx % (F / 4);
    const int ps = row < T::RPAD ? (int)geo[row] : 0xffff;          // board * P + position of this buffer row
    float4 v = ps < nvalid ? in4[(size_t)ps * (F / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (bn.g && ps < nvalid) {                                       // BnIn: v holds g; the activation is computed here and kept for the backward pass
      const float4 mu = ((const float4*)bn.mean)[c4], is = ((const float4*)bn.invstd)[c4], ga = ((const float4*)bn.gamma)[c4], be = ((const float4*)bn.beta)[c4];
      float4 y;
      y.x = ga.x * ((v.x - mu.x) * is.x) + be.x; y.y = ga.y * ((v.y - mu.y) * is.y) + be.y;
      y.z = ga.z * ((v.z - mu.z) * is.z) + be.z; y.w = ga.w * ((v.w - mu.w) * is.w) + be.w;
      if (res4) { const float4 r = res4[(size_t)ps * (F / 4) + c4]; y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w; }
      v = make_float4(y.x > 0.0f ? y.x : 0.0f, y.y > 0.0f ? y.y : 0.0f, y.z > 0.0f ? y.z : 0.0f, y.w > 0.0f ? y.w : 0.0f);
      aout4[(size_t)ps * (F / 4) + c4] = v;
    }
    float* dst = buf + row * STRIDE;
    dst[posF<F>(c4 * 4 + 0)] = v.x; dst[posF<F>(c4 * 4 + 1)] = v.y; dst[posF<F>(c4 * 4 + 2)] = v.z; dst[posF<F>(c4 * 4 + 3)] = v.w;
  }
  for (int i = tid; i < T::RPAD; i += T::THREADS) pos[i] = geo[i];
  for (int i = tid; i < 9 * T::RPAD; i += T::THREADS) nbr[i] = geo[T::RPAD + i];
  const int lrow = lane & 15, g = lane >> 4;
  __syncthreads();
  if constexpr (STAMP) ts[1] = wall_clock64();
  f32x4v acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  conv16p<T, G, NT, 0, 9>(buf, nbr, wfrag + (size_t)wave * T::SQ * 64 + lane, acc, lrow, g);
  if constexpr (STAMP) { ts[2] = wall_clock64(); __syncthreads(); }    // the barrier: the epilogue reading then starts when the LAST wavefront has its products
  const int ch = wave * 16 + lrow;
  float* o = out + (size_t)board0 * P * F;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int tile = 0; tile < NT; ++tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ps = pos[tile * 16 + g * 4 + i];
      if (ps < nvalid) {
        // addend (the data-gradient pass of a block's first convolution): the skip connection's share of the gradient joins here
        // instead of in a pass of its own over the 22 MB tensor (same single fp32 add)
        o[(size_t)ps * F + ch] = addend ? acc[tile][i] + (addend + (size_t)board0 * P * F)[(size_t)ps * F + ch] : acc[tile][i];
        if constexpr (STATS) { const double v = (double)acc[tile][i]; s0 += v; s1 += v * v; }
      }
    }
  if constexpr (STATS) {
    // the four row groups of a channel: (g0 + g1) + (g2 + g3), a fixed order
    s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
    s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
    if (g == 0) { part_store(part + ((size_t)blockIdx.x * 2) * F + ch, s0); part_store(part + ((size_t)blockIdx.x * 2 + 1) * F + ch, s1); }
    __syncthreads();                                                 // every wavefront is done with the LDS buffer: it becomes the finisher's scratch
    tr_finish(part, F, fin, (double*)lds);
  }
  if constexpr (STAMP) {
    __builtin_amdgcn_s_waitcnt(0);
    ts[3] = wall_clock64();
    if (tid == 0) for (int i = 0; i < 4; ++i) stamps[(size_t)blockIdx.x * 4 + i] = ts[i];
  }
}

// Weight gradient of a 3x3 F -> F convolution for the optimiser step (train.h):
//     dW[tap][ci][co] = sum over rows r of a[r + delta(tap)][ci] * dg[r][co]        (taps that leave the board contribute 0)
// as MFMA 16x16x4 products with the ROWS as the reduction dimension: A operand = 16 input channels x 4 rows of the layer
// input (shifted by the tap), B operand = 4 rows x 16 output channels of the output gradient.  A workgroup walks its
// boards three at a time (126 rows + 2 zero rows in LDS, natural channel order); wavefront w owns input-channel tile w
// and keeps accumulators for TPW taps x all F/16 output-channel tiles in registers (F = 128: 3 taps per workgroup,
// blockIdx.y selects the group; F = 64: all 9 taps).  Every workgroup writes its partial dW, a second kernel adds the
// partials in a fixed order (deterministic, no atomics).
// (r3) The reduction runs over the rows in ANY order, so each tap walks only the rows whose shifted neighbour is on the
// board: a table per tap lists them (a corner tap of a 7x6 board keeps 30 of 42 rows, an edge tap 35 or 36), and the three
// taps of a workgroup are chosen so that every group has the same work -- {two corners, centre}, {corner, two edges},
// {corner, two edges}: 78 / 77 / 77 four-row steps per 3-board chunk instead of 96.
// CS / RPC (r3): CS = 2 gives every workgroup one HALF of the input channels (4 wavefronts at F = 128) and RPC = 48 makes
// a chunk one board: 44 KB of LDS instead of 150 and one wavefront per SIMD, so that a workgroup fits on a CU BESIDE a
// workgroup of k_conv16_layer (104 KB, two wavefronts per SIMD at <= 128 registers) -- the weight gradient of layer l then
// shares the matrix pipes with the data-gradient convolution of layer l - 1 instead of waiting for it (train.hip).
template <int F, int CS_ = 1, int RPC_ = 128> struct WG16 {
  static constexpr int TPW = F == 128 ? 3 : 9;                    // taps per workgroup
  static constexpr int TG = 9 / TPW;                              // tap groups
  static constexpr int CS = CS_;                                  // splits of the input channels (grid.y = TG * CS)
  static constexpr int FA = F / CS;                               // input channels of a workgroup
  static constexpr int CT = F / 16, WAVES = FA / 16, THREADS = 64 * WAVES;
  static constexpr int RP = RPC_;                                 // padded rows of a chunk (128: 3 x 42 = 126 used; other games: RP / P boards)
  static constexpr int STRIDE_A = FA + 16, STRIDE = F + 16;       // 16-bank shift per row: the 4 row groups of a fragment read use disjoint bank halves in pairs
  static constexpr int RLS = RP + 8;                              // row-list entries per tap (the pipeline reads two steps ahead)
  static constexpr int BYTES = ((RP + 1) * STRIDE_A + RP * STRIDE + TPW * RLS) * 4;
  static constexpr int WGS_PER_CU = (RP <= 64 && CS > 1) ? 2 : 1;
  // tap k of group y: balanced groups for TPW == 3, natural order otherwise
  __host__ __device__ static constexpr int tap(int y, int k) {
    constexpr int sel[3][3] = {{0, 8, 4}, {2, 1, 3}, {6, 7, 5}};
    return TPW == 3 ? sel[y][k] : k;
  }
};
// STAMP (tools/probes/wgrad_stamps.hip): every workgroup leaves clock readings in stamps[workgroup][8]; 2: the steps read no
// LDS operands, 3: they issue no MFMAs (what bounds the loop)
template <class Gm, int F, int STAMP = 0, int CS = 1, int RPC = 128>
__global__ void __launch_bounds__((WG16<F, CS, RPC>::THREADS), (WG16<F, CS, RPC>::WGS_PER_CU))
k_wgrad16(const float* __restrict__ a, const float* __restrict__ dg, float* __restrict__ part, int nboards, int nsplits, long long* __restrict__ stamps) {
  long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (STAMP) ts[0] = wall_clock64();
  using G = WG16<F, CS, RPC>;
  constexpr int P = Gm::P, W = Gm::W, H = Gm::H, STRIDE = G::STRIDE, STRIDE_A = G::STRIDE_A, FA = G::FA, RP = G::RP, CT = G::CT, TPW = G::TPW, RLS = G::RLS;
  constexpr int NBC = RP / P < 1 ? 1 : RP / P;                    // boards per chunk for this game
  static_assert(RP >= P, "a chunk holds at least one board");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                                                // [(RP + 1)][STRIDE_A], row RP = zeros
  float* Ds = lds + (RP + 1) * STRIDE_A;                          // [RP][STRIDE]
  uint32_t* rl = (uint32_t*)(Ds + RP * STRIDE);                   // [TPW][RLS] row lists: (row of a) << 16 | row of dg
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, g = lane >> 4;
  const int tgroup = blockIdx.y % G::TG, half = blockIdx.y / G::TG;  // tap group, input-channel split
  // boards of this workgroup: nboards spread evenly over the nsplits workgroups of a tap group (the first nboards % nsplits
  // take one more; a last partial LDS chunk costs only its rows)
  const int bq = nboards / nsplits, br = nboards % nsplits;
  const int b_begin = blockIdx.x * bq + ((int)blockIdx.x < br ? (int)blockIdx.x : br);
  const int b_end = b_begin + bq + ((int)blockIdx.x < br ? 1 : 0);
  // The chunk after the current one travels from HBM into registers while the MFMAs of the current one run (every
  // workgroup reaches its chunk boundaries at the same time: without the overlap the chip alternates between a burst of
  // loads and a burst of MFMAs); it is stored to LDS once the current chunk has been consumed.
  constexpr int NPA = RP * (FA / 4) / G::THREADS, NPD = RP * (F / 4) / G::THREADS;   // float4 per thread: a, dg
  static_assert(RP * (FA / 4) % G::THREADS == 0 && RP * (F / 4) % G::THREADS == 0, "chunk must divide over the threads");
  float4 pa[NPA], pd[NPD];
  // (issued in one burst after the chunk's barrier: spread over the taps' loops they cost 11 more registers, and the registers
  // this kernel leaves free decide what can share the CU with it: 2 x 160 of k_conv16_layer + 192 of the 4-wavefront form = 512)
  auto prefetch = [&](int b0) {
    const int nbp = (b_end - b0) < NBC ? (b_end - b0) : NBC;
    const int nv = nbp > 0 ? nbp * P : 0;
    const float4* a4 = (const float4*)(a + (size_t)b0 * P * F + half * FA);
    const float4* d4 = (const float4*)(dg + (size_t)b0 * P * F);
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int idx = tid + q * G::THREADS, row = idx / (FA / 4), c4 = idx % (FA / 4);
      pa[q] = row < nv ? a4[(size_t)row * (F / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < NPD; ++q) {
      const int idx = tid + q * G::THREADS, row = idx / (F / 4), c4 = idx % (F / 4);
      pd[q] = row < nv ? d4[(size_t)row * (F / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  prefetch(b_begin);                                              // first: its latency covers the table build
  // row lists: for tap k the rows of a chunk whose neighbour at the tap's offset is on the board, board after board; the
  // rest of a list pairs the zero row of `a` with any row of dg
  for (int i = tid; i < TPW * RLS; i += G::THREADS) rl[i] = ((uint32_t)RP << 16) | (uint32_t)(RP - 1);
  for (int i = tid; i < STRIDE_A; i += G::THREADS) As[RP * STRIDE_A + i] = 0.0f;
  __syncthreads();
  int cnt[TPW];                                                   // valid rows per board of tap k
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int tap = G::tap(tgroup, k), dy = tap / 3 - 1, dx = tap % 3 - 1;
    cnt[k] = (W - (dx != 0)) * (H - (dy != 0));
  }
  if (tid < TPW * NBC) {
    const int k = tid / NBC, b = tid % NBC;
    const int tap = G::tap(tgroup, k), dy = tap / 3 - 1, dx = tap % 3 - 1;
    int n = b * (W - (dx != 0)) * (H - (dy != 0));
    for (int q = 0; q < P; ++q) {
      const int x = q % W, y = q / W;
      if ((y + dy >= 0) && (y + dy < H) && (x + dx >= 0) && (x + dx < W)) {
        const int r = b * P + q;
        rl[k * RLS + n++] = ((uint32_t)(r + dy * W + dx) << 16) | (uint32_t)r;
      }
    }
  }
  f32x4v acc[TPW][CT];
#pragma unroll
  for (int k = 0; k < TPW; ++k)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[k][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  if constexpr (STAMP) ts[1] = wall_clock64();
  for (int b0 = b_begin; b0 < b_end; b0 += NBC) {
    const int nb = (b_end - b0) < NBC ? (b_end - b0) : NBC;
    long long tb = 0;
    if constexpr (STAMP) tb = wall_clock64();
    __syncthreads();                                              // the previous chunk has been consumed (first pass: the row lists are complete)
    long long tb1 = 0;
    if constexpr (STAMP) tb1 = wall_clock64();
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int idx = tid + q * G::THREADS, row = idx / (FA / 4), c4 = idx % (FA / 4);
      *(float4*)(As + row * STRIDE_A + c4 * 4) = pa[q];
    }
#pragma unroll
    for (int q = 0; q < NPD; ++q) {
      const int idx = tid + q * G::THREADS, row = idx / (F / 4), c4 = idx % (F / 4);
      *(float4*)(Ds + row * STRIDE + c4 * 4) = pd[q];
    }
    __syncthreads();
    long long tb2 = 0;
    if constexpr (STAMP) tb2 = wall_clock64();
    if (b0 + NBC < b_end) prefetch(b0 + NBC);
    if constexpr (STAMP) { const long long tn = wall_clock64(); if (b0 == b_begin) ts[2] = tn; else { ts[3] += tn - tb; ts[6] += tb1 - tb; ts[7] += tb2 - tb1; } }
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      // the rows of tap k in this chunk, four per step (a list's tail is zero rows).  Two register stages: the LDS
      // operands of step s+1 are requested before the CT MFMAs of step s issue, the list entry of step s+2 before that;
      // an odd step at the end runs on the operands the last pair already requested.
      const int nsteps = (nb * cnt[k] + 3) >> 2;
      const uint32_t* list = rl + k * RLS + g;
      float bv0[CT], bv1[CT], av0, av1;
      auto load_step = [&](uint32_t e, float (&bv)[CT], float& av) {
        if constexpr (STAMP == 2) {
#pragma unroll
          for (int j = 0; j < CT; ++j) bv[j] = __uint_as_float(e + j);
          av = __uint_as_float(e);
          return;
        }
        const float* dp = Ds + (e & 0xffffu) * STRIDE + lrow;
#pragma unroll
        for (int j = 0; j < CT; ++j) bv[j] = dp[j * 16];
        av = As[(e >> 16) * STRIDE_A + wave * 16 + lrow];
      };
      auto mfma_step = [&](const float (&bv)[CT], float av) {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          if constexpr (STAMP == 3) acc[k][j][0] += av * bv[j];
          else acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[k][j], 0, 0, 0);
        }
      };
      // one MFMA, then a share of the next step's address arithmetic and LDS reads: a wavefront that runs alone on its SIMD
      // (its partner waits at the chunk barrier, or is in its own load phase) keeps the matrix pipe busy by itself
      auto interleave = [&]() {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      };
      uint32_t e0 = list[0], e1 = list[4];
      load_step(e0, bv0, av0);
      for (int s0 = 0; s0 + 2 <= nsteps; s0 += 2) {
        __builtin_amdgcn_sched_barrier(0);
        e0 = list[4 * (s0 + 2)];
        load_step(e1, bv1, av1);
        mfma_step(bv0, av0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        e1 = list[4 * (s0 + 3)];
        load_step(e0, bv0, av0);
        mfma_step(bv1, av1);
        interleave();
      }
      __builtin_amdgcn_sched_barrier(0);
      if (nsteps & 1) mfma_step(bv0, av0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (STAMP) ts[4] = wall_clock64();
  // partial dW of this workgroup: [split][tap][ci][co], ci = FA half + 16 wave + 4 g + i, co = 16 j + lrow
  float* o = part + (size_t)blockIdx.x * 9 * F * F;
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int tap = G::tap(tgroup, k);
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        o[((size_t)tap * F + (half * FA + wave * 16 + g * 4 + i)) * F + j * 16 + lrow] = acc[k][j][i];
  }
  if constexpr (STAMP) {
    __builtin_amdgcn_s_waitcnt(0);
    ts[5] = wall_clock64();
    if (tid == 0) for (int i = 0; i < 8; ++i) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = ts[i];
  }
}
// two outputs per thread (n is even: n = 9 F F), 8-byte loads, eight in flight: beside a convolution and a k_wgrad16
// workgroup 56 registers per SIMD lane are free -- with 4-byte loads this kernel moved 1 TB/s there (50 us for the 50 MB of a
// layer), with 16-byte loads it needs 70 registers and waits for a CU
static __global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ part, int nsplit, long long n, float* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);   // short and HBM-bound, and the next k_wgrad16 of its stream waits for it
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  // eight independent partial sums keep eight loads in flight; combined in a fixed order (per output the same sums as ever)
  float2 s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = make_float2(0.f, 0.f);
  int k = 0;
  for (; k + 8 <= nsplit; k += 8)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float2 v = *(const float2*)(part + (size_t)(k + u) * n + i);
      s[u].x += v.x; s[u].y += v.y;
    }
  for (; k < nsplit; ++k) {
    const float2 v = *(const float2*)(part + (size_t)k * n + i);
    s[0].x += v.x; s[0].y += v.y;
  }
  float2 r;
  r.x = ((s[0].x + s[1].x) + (s[2].x + s[3].x)) + ((s[4].x + s[5].x) + (s[6].x + s[7].x));
  r.y = ((s[0].y + s[1].y) + (s[2].y + s[3].y)) + ((s[4].y + s[5].y) + (s[6].y + s[7].y));
  *(float2*)(out + i) = r;
}
