// prims.h -- the three device-wide primitives memory.hip needs (merge_by_state, src/memory.jl:89-114, and the data-set sums of
// src/learning.jl:110-111), hand-written for gfx950: a stable LSD radix sort of (u64 key, u32 value) pairs, an inclusive
// scan of ints and a deterministic sum of doubles.  Round 2 took them from hipCUB, the last third-party device code in the
// library.  Off the hot path (a data set is built once per learning step), so the design goal is small and obviously
// stable / deterministic rather than fastest: 8-bit digits, one workgroup per tile of 2048 pairs, and a per-tile local
// split so that equal digits keep their order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prims {
static constexpr int RS_THREADS = 256, RS_ITEMS = 8, RS_TILE = RS_THREADS * RS_ITEMS, RS_BINS = 256;

// ---- block-wide exclusive scan of one int per thread (256 threads), returns the block total in *total --------------------------
__device__ __forceinline__ int block_excl_scan(int x, int* s_wave /* [4] */, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int v = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
  if (lane == 63) s_wave[wave] = v;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RS_THREADS / 64; ++w) { if (w < wave) base += s_wave[w]; tot += s_wave[w]; }
  __syncthreads();
  *total = tot;
  return base + v - x;
}

// ---- radix sort ------------------------------------------------------------------------------------------------------------------
// pass 1 of a digit: per-tile histogram, written bin-major (hist[bin * ntiles + tile]) so that ONE exclusive scan over the
// whole array gives every (bin, tile) its first output position
static __global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const unsigned long long* __restrict__ keys, long long n, int shift, int ntiles,
                                                               int* __restrict__ hist) {
  __shared__ int s_h[RS_BINS];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long p = base + (long long)i * RS_THREADS + threadIdx.x;
    if (p < n) atomicAdd(&s_h[(int)((keys[p] >> shift) & 0xff)], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];
}
// pass 3: the tile's pairs go to out[first(bin, tile) + rank among the tile's pairs of that bin, in input order].  The tile is
// walked in its 8 rows of 256 consecutive pairs; inside a row a thread finds the lanes of its wavefront that hold the same
// digit with one ballot per digit bit (a match-any), takes its rank among them, and the first of them fetches the bin's running
// counter from LDS -- the wavefronts take their turns in order, so earlier pairs always get lower ranks (stability).
static __global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ vals,
                                                                  long long n, int shift, int ntiles, const int* __restrict__ first,
                                                                  unsigned long long* __restrict__ keys_out, unsigned int* __restrict__ vals_out) {
  __shared__ int s_next[RS_BINS];                                   // next free rank of each bin inside this tile
  s_next[threadIdx.x] = first[(size_t)threadIdx.x * ntiles + blockIdx.x];
  __syncthreads();
  const long long base = (long long)blockIdx.x * RS_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long p = base + (long long)i * RS_THREADS + threadIdx.x;
    const bool ok = p < n;
    const unsigned long long k = ok ? keys[p] : 0ULL;
    const unsigned int d = (unsigned int)((k >> shift) & 0xff);
    // lanes of this wavefront with the same digit (invalid lanes match nobody)
    unsigned long long same = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bit = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? bit : ~bit;
    }
    const int before = __popcll(same & ((1ULL << lane) - 1ULL));   // equal digits in lower lanes
    const int count = __popcll(same);
    const bool leader = ok && before == 0;
    // wavefronts take their turns in order, so that earlier pairs get lower ranks
    int start = 0;
    for (int w = 0; w < RS_THREADS / 64; ++w) {
      if (w == wave && leader) { start = s_next[d]; s_next[d] = start + count; }
      __syncthreads();
    }
    // the leader's start reaches the other lanes of its digit
    const int src = __ffsll((unsigned long long)same) - 1;
    start = __shfl(start, src < 0 ? 0 : src);
    if (ok) {
      const long long o = (long long)start + before;
      keys_out[o] = k;
      vals_out[o] = vals[p];
    }
  }
}

// ---- scan of an int array (inclusive or exclusive), any length: tiles of 2048, tile sums scanned recursively -----------------------------
static __global__ void __launch_bounds__(RS_THREADS) k_scan_tiles(const int* __restrict__ in, long long n, int* __restrict__ out, int* __restrict__ tile_sums, int inclusive) {
  __shared__ int s_wave[4];
  const long long base = (long long)blockIdx.x * RS_TILE + (long long)threadIdx.x * RS_ITEMS;
  int v[RS_ITEMS], sum = 0;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) { v[i] = base + i < n ? in[base + i] : 0; sum += v[i]; }
  int total;
  int run = block_excl_scan(sum, s_wave, &total);
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    if (base + i < n) out[base + i] = inclusive ? run + v[i] : run;
    run += v[i];
  }
  if (threadIdx.x == 0 && tile_sums) tile_sums[blockIdx.x] = total;
}
static __global__ void __launch_bounds__(RS_THREADS) k_scan_add(int* __restrict__ out, long long n, const int* __restrict__ tile_offsets) {
  const int off = tile_offsets[blockIdx.x];
  const long long base = (long long)blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long p = base + (long long)i * RS_THREADS + threadIdx.x;
    if (p < n) out[p] += off;
  }
}
inline size_t scan_tmp_ints(long long n) {                        // ints of scratch the recursion needs
  size_t tot = 0;
  for (long long m = (n + RS_TILE - 1) / RS_TILE; m > 1; m = (m + RS_TILE - 1) / RS_TILE) tot += 2 * (size_t)m;
  return tot + 2;
}
// out may alias in.  tmp: scan_tmp_ints(n) ints.
inline hipError_t scan_ints(const int* in, int* out, long long n, bool inclusive, int* tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const long long tiles = (n + RS_TILE - 1) / RS_TILE;
  if (tiles == 1) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(RS_THREADS), 0, st, in, n, out, (int*)nullptr, inclusive ? 1 : 0);
    return hipGetLastError();
  }
  int* sums = tmp;
  int* offs = tmp + tiles;
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, in, n, out, sums, inclusive ? 1 : 0);
  hipError_t e = scan_ints(sums, offs, tiles, false, tmp + 2 * tiles, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, out, n, offs);
  return hipGetLastError();
}

// scratch of sort_pairs: ints
inline size_t sort_tmp_ints(long long n) {
  const long long tiles = (n + RS_TILE - 1) / RS_TILE;
  return (size_t)RS_BINS * tiles * 2 + scan_tmp_ints((long long)RS_BINS * tiles);
}
// Stable sort of n (key, value) pairs by key, ascending; result in keys_out / vals_out; keys_in / vals_in are used as the
// ping-pong partner (destroyed).  tmp: sort_tmp_ints(n) ints.
inline hipError_t sort_pairs(unsigned long long* keys_in, unsigned long long* keys_out, unsigned int* vals_in, unsigned int* vals_out,
                             long long n, int* tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const long long tiles = (n + RS_TILE - 1) / RS_TILE;
  int* hist = tmp;
  int* first = tmp + (size_t)RS_BINS * tiles;
  int* scan_tmp = first + (size_t)RS_BINS * tiles;
  unsigned long long *ka = keys_in, *kb = keys_out;
  unsigned int *va = vals_in, *vb = vals_out;
  for (int pass = 0; pass < 8; ++pass) {                            // an even number of passes over two buffers: pass 7 writes (keys_in, vals_in), copied to the caller's out below
    const int shift = 8 * pass;
    hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, ka, n, shift, (int)tiles, hist);
    hipError_t e = scan_ints(hist, first, (long long)RS_BINS * tiles, false, scan_tmp, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, ka, va, n, shift, (int)tiles, first, kb, vb);
    unsigned long long* tk = ka; ka = kb; kb = tk;
    unsigned int* tv = va; va = vb; vb = tv;
  }
  // after 8 passes the data is back in (keys_in, vals_in): one copy puts it where the caller wants it
  hipError_t e = hipMemcpyAsync(keys_out, keys_in, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(vals_out, vals_in, sizeof(unsigned int) * (size_t)n, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

// ---- deterministic sum of n doubles: fixed tiles of 2048 summed by a fixed tree, tile sums summed by one workgroup in order ------------
static __global__ void __launch_bounds__(RS_THREADS) k_sum_tiles(const double* __restrict__ in, long long n, double* __restrict__ out) {
  __shared__ double s[RS_THREADS];
  const long long base = (long long)blockIdx.x * RS_TILE;
  double a = 0.0;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long p = base + (long long)i * RS_THREADS + threadIdx.x;
    if (p < n) a += in[p];
  }
  s[threadIdx.x] = a;
  __syncthreads();
  for (int w = RS_THREADS / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}
inline size_t sum_tmp_doubles(long long n) {
  size_t tot = 0;
  for (long long m = (n + RS_TILE - 1) / RS_TILE; ; m = (m + RS_TILE - 1) / RS_TILE) { tot += (size_t)m; if (m <= 1) break; }
  return tot + 1;
}
// result in tmp[last level]; returns the device pointer of the scalar through *d_result
inline hipError_t sum_doubles(const double* in, long long n, double* tmp, double** d_result, hipStream_t st) {
  const double* src = in;
  double* dst = tmp;
  long long m = n;
  for (;;) {
    const long long tiles = (m + RS_TILE - 1) / RS_TILE < 1 ? 1 : (m + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(k_sum_tiles, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, src, m, dst);
    if (tiles == 1) { *d_result = dst; return hipGetLastError(); }
    src = dst; dst += tiles; m = tiles;
  }
}
}  // namespace prims
