// Probe: which XCD does workgroup b land on, and how long does a one-way hand-over of an 8-byte (value, tag) word take
// between two workgroups, for the store / load flavours k_tower16s could use.  Pairs (b, b ^ 8) [same XCD if workgroups
// go round-robin] and (b, b ^ 1) [neighbouring XCDs].
//   hipcc --offload-arch=gfx950 -O2 -o xcd_exchange xcd_exchange.hip && ./xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;
enum { LD_PLAIN, LD_SC0, LD_SC1, LD_RMW };
enum { ST_PLAIN, ST_SC0, ST_SC1 };

template <int LD> __device__ __forceinline__ u64 ld(u64* p) {
  u64 v;
  if constexpr (LD == LD_PLAIN) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else if constexpr (LD == LD_SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else if constexpr (LD == LD_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else v = __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return v;
}
template <int ST> __device__ __forceinline__ void st(u64* p, u64 v) {
  if constexpr (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  else if constexpr (ST == ST_SC0) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// ping-pong: workgroup `half 0` writes round r to its word, the partner waits for it and answers in its own word
template <int LD, int ST>
__global__ void k_pingpong(u64* words, int partner_xor, int rounds, u64 base, long long* cycles, int* fails, unsigned* xcc) {
  const int b = blockIdx.x, pb = b ^ partner_xor;
  if (threadIdx.x == 0) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); xcc[b] = x & 0xf; }
  u64* mine = words + (size_t)b * 64 + threadIdx.x;       // one word per lane: 64 lanes poll like a wavefront of the tower
  u64* theirs = words + (size_t)pb * 64 + threadIdx.x;
  const bool first = (b & partner_xor) == 0;
  long long t0 = __builtin_readcyclecounter();
  int bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    const u64 tag = base + r;
    if (first) st<ST>(mine, tag);
    int spins = 0;
    while (ld<LD>(theirs) != tag) { if (++spins > 20000) { bad = 1; break; } }
    if (bad) break;
    if (!first) st<ST>(mine, tag);
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { cycles[b] = t1 - t0; if (bad) atomicAdd(fails, 1); }
}

template <int LD, int ST> static void run(const char* name, int partner_xor, u64* words, long long* cyc, int* fails, unsigned* xcc, u64& base) {
  const int nb = 256, rounds = 200;
  hipMemset(fails, 0, sizeof(int));
  hipLaunchKernelGGL((k_pingpong<LD, ST>), dim3(nb), dim3(64), 0, 0, words, partner_xor, rounds, base, cyc, fails, xcc);
  base += 1000;
  hipError_t e = hipDeviceSynchronize();
  std::vector<long long> h(nb); int f = 0; std::vector<unsigned> hx(nb);
  hipMemcpy(h.data(), cyc, sizeof(long long) * nb, hipMemcpyDeviceToHost);
  hipMemcpy(&f, fails, sizeof(int), hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), xcc, sizeof(unsigned) * nb, hipMemcpyDeviceToHost);
  double s = 0; for (long long c : h) s += (double)c;
  int same = 0; for (int b = 0; b < nb; ++b) same += hx[b] == hx[b ^ partner_xor];
  printf("%-28s partner b^%d: %s, same-XCD pairs %d/%d, failed workgroups %d, one-way hand-over %.0f cycles\n", name, partner_xor,
         e == hipSuccess ? "ok" : hipGetErrorString(e), same, nb, f, s / nb / rounds / 2);
}

int main() {
  u64* words; long long* cyc; int* fails; unsigned* xcc;
  hipMalloc(&words, 256 * 64 * sizeof(u64)); hipMemset(words, 0, 256 * 64 * sizeof(u64));
  hipMalloc(&cyc, 256 * sizeof(long long)); hipMalloc(&fails, sizeof(int)); hipMalloc(&xcc, 256 * sizeof(unsigned));
  u64 base = 1000;
  hipLaunchKernelGGL((k_pingpong<LD_SC1, ST_SC1>), dim3(256), dim3(64), 0, 0, words, 1, 1, base, cyc, fails, xcc); base += 1000;
  hipDeviceSynchronize();
  std::vector<unsigned> hx(256);
  hipMemcpy(hx.data(), xcc, sizeof(unsigned) * 256, hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..31:"); for (int b = 0; b < 32; ++b) printf(" %u", hx[b]); printf("\n");
  for (int px : {8, 1}) {
    run<LD_SC1, ST_SC1>("load sc1 / store sc1", px, words, cyc, fails, xcc, base);
    run<LD_SC0, ST_SC1>("load sc0 / store sc1", px, words, cyc, fails, xcc, base);
    run<LD_SC0, ST_SC0>("load sc0 / store sc0", px, words, cyc, fails, xcc, base);
    run<LD_SC0, ST_PLAIN>("load sc0 / store plain", px, words, cyc, fails, xcc, base);
    run<LD_RMW, ST_SC1>("load rmw(L2) / store sc1", px, words, cyc, fails, xcc, base);
    run<LD_RMW, ST_PLAIN>("load rmw(L2) / store plain", px, words, cyc, fails, xcc, base);
    run<LD_PLAIN, ST_PLAIN>("load plain / store plain", px, words, cyc, fails, xcc, base);
  }
  return 0;
}
