// Where k_wgrad16's time goes (clock readings of every workgroup): set-up + first prefetch issue, first chunk in LDS,
// chunk boundaries, main loops, the partial-dW store.  1024 Connect-Four boards, 128 filters, the trainer's grids:
// the 8-wavefront form (3-board chunks, 150 KB of LDS) and the 4-wavefront form (half the input channels, 1-board chunks, 44 KB).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fconstexpr-steps=200000000 -I../../alphazero.jl_amd/csrc wgrad_stamps.hip -o wgrad_stamps
#include "resnet16.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static std::vector<float> g_ref;
template <int VAR, int CS, int RPC> static int run(int splits) {
  constexpr int F = 128, B = 1024, P = ConnectFour::P;
  using G = WG16<F, CS, RPC>;
  const int ny = G::TG * CS, nwg = splits * ny;
  float *a, *dg, *part, *out; long long* st;
  const size_t n = (size_t)B * P * F, nw = (size_t)9 * F * F;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&dg, n * 4)); CK(hipMalloc(&part, (size_t)splits * nw * 4)); CK(hipMalloc(&out, nw * 4)); CK(hipMalloc(&st, (size_t)nwg * 8 * 8));
  std::vector<float> h(n), h2(n);
  for (size_t i = 0; i < n; ++i) { h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f; h2[i] = (float)((i * 40503u + 12345u) >> 4 & 0xffff) / 65536.0f - 0.5f; }
  CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, h2.data(), n * 4, hipMemcpyHostToDevice));
  auto kern = k_wgrad16<ConnectFour, F, VAR, CS, RPC>;
  printf("variant %d (1: the kernel, 2: no LDS operand reads, 3: no MFMAs), %d-way channel split, %d-row chunks: %d x %d workgroups of %d threads, %d bytes of LDS\n",
         VAR, CS, RPC, splits, ny, G::THREADS, (int)G::BYTES);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(splits, ny), dim3(G::THREADS), G::BYTES, 0, a, dg, part, B, splits, st);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> s((size_t)nwg * 8);
    CK(hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = s[0], tend = 0;
    for (size_t w = 0; w < (size_t)nwg; ++w) { t0 = std::min(t0, s[w * 8]); tend = std::max(tend, s[w * 8 + 5]); }
    double av[8] = {0, 0, 0, 0, 0, 0, 0, 0}; double mx_start = 0;
    for (size_t w = 0; w < (size_t)nwg; ++w) {
      const long long* q = &s[w * 8];
      av[0] += q[1] - q[0]; av[1] += q[2] - q[1]; av[2] += q[3]; av[3] += (q[4] - q[2]) - q[3]; av[4] += q[5] - q[4]; av[5] += q[5] - q[0]; av[6] += q[6]; av[7] += q[7];
      mx_start = std::max(mx_start, (double)(q[0] - t0));
    }
    const double nwd = nwg, tick = 0.01;   // wall_clock64: 100 MHz
    if (it == 2)
      printf("  %.1f us by events | first start -> last end %.1f us, latest start +%.1f us | per workgroup (mean, us): set-up %.1f, first chunk in LDS %.1f, "
             "chunk boundaries %.1f (wait at the first barrier %.1f, LDS stores + second barrier %.1f), main loops %.1f, store %.1f, total %.1f\n", ms * 1e3, (tend - t0) * tick, mx_start * tick,
             av[0] / nwd * tick, av[1] / nwd * tick, av[2] / nwd * tick, av[6] / nwd * tick, av[7] / nwd * tick, av[3] / nwd * tick, av[4] / nwd * tick, av[5] / nwd * tick);
  }
  if (VAR == 1) {
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((nw / 2 + 255) / 256)), dim3(256), 0, 0, part, splits, (long long)nw, out);
    std::vector<float> r(nw);
    CK(hipMemcpy(r.data(), out, nw * 4, hipMemcpyDeviceToHost));
    if (g_ref.empty()) g_ref = r;
    else {
      double md = 0, mx = 0;
      for (size_t i = 0; i < nw; ++i) { md = std::max(md, (double)std::fabs(r[i] - g_ref[i])); mx = std::max(mx, (double)std::fabs(g_ref[i])); }
      printf("  against the first form: largest difference %.3g of %.3g (%.2g relative; the two sum the rows in different groups of four)\n", md, mx, md / mx);
    }
  }
  CK(hipFree(a)); CK(hipFree(dg)); CK(hipFree(part)); CK(hipFree(out)); CK(hipFree(st));
  return 0;
}
int main(int argc, char** argv) {
  if (argc > 1) { g_ref.clear(); return run<1, 2, 48>(85) || run<1, 2, 48>(64) || run<1, 2, 48>(43) || run<1, 2, 48>(128); }   // any argument: the split count of the 4-wavefront form
  return run<1, 1, 128>(85) || run<1, 2, 48>(85) || run<2, 2, 48>(85) || run<3, 2, 48>(85);
}
