// Probe: latency of a DEPENDENT global load (pointer chase, one lane) by working-set size -- what every ply of k_tree's
// descent and every hash probe costs.  128-byte stride, random cycle.
//   hipcc --offload-arch=gfx950 -O2 -o chase_latency chase_latency.hip && ./chase_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#include <algorithm>
#include <random>
__global__ void k_chase(const unsigned* __restrict__ next, int hops, long long* cycles, unsigned* sink) {
  unsigned p = 0;
  for (int i = 0; i < 1024; ++i) p = next[(size_t)p * 32];          // warm the TLB / caches a little
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < hops; ++i) p = next[(size_t)p * 32];
  long long t1 = __builtin_readcyclecounter();
  cycles[0] = t1 - t0; sink[0] = p;
}
int main() {
  std::mt19937 rng(1);
  for (size_t mb : {1, 2, 8, 32, 128, 512, 4096}) {
    const size_t n = mb * 1024 * 1024 / 128;                        // one 128-byte line per node
    std::vector<unsigned> perm(n); std::iota(perm.begin(), perm.end(), 0u);
    std::shuffle(perm.begin() + 1, perm.end(), rng);
    std::vector<unsigned> nxt(n * 32, 0);
    for (size_t i = 0; i < n; ++i) nxt[(size_t)perm[i] * 32] = perm[(i + 1) % n];
    unsigned* d; long long* c; unsigned* s;
    hipMalloc(&d, nxt.size() * 4); hipMalloc(&c, 8); hipMalloc(&s, 4);
    hipMemcpy(d, nxt.data(), nxt.size() * 4, hipMemcpyHostToDevice);
    const int hops = 20000;
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, hops, c, s);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("working set %5zu MB: %.0f cycles per dependent load\n", mb, (double)h / hops);
    hipFree(d); hipFree(c); hipFree(s);
  }
  return 0;
}
