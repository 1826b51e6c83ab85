// Where k_conv16_layer's time goes: fill (HBM -> LDS in Geo16 order), the 85 tile-tap products, epilogue (stores + the
// batch-norm column sums).  1024 Connect-Four boards, 128 filters = 256 workgroups of 4 boards, one per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fconstexpr-steps=200000000 -I../../alphazero.jl_amd/csrc conv_stamps.hip -o conv_stamps
#include "resnet16.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <bool STATS> static int run() {
  constexpr int F = 128, B = 1024, P = ConnectFour::P;
  using T = T16<ConnectFour, F, 11>;
  using G = typename T::Geo;
  const int nwg = (B + T::TB - 1) / T::TB;
  const size_t n = (size_t)B * P * F;
  float *in, *out, *w; double* part; long long* st; uint16_t* geo;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&w, (size_t)9 * F * F * 4)); CK(hipMalloc(&part, (size_t)nwg * 2 * F * 8)); CK(hipMalloc(&st, (size_t)nwg * 4 * 8));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
  CK(hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), (size_t)9 * F * F * 4, hipMemcpyHostToDevice));
  std::vector<uint16_t> hg((size_t)10 * G::RPAD);
  for (int i = 0; i < G::RPAD; ++i) hg[i] = G::tab.pos[i];
  for (int i = 0; i < 9 * G::RPAD; ++i) hg[G::RPAD + i] = G::tab.nbr[i];
  CK(hipMalloc(&geo, hg.size() * 2)); CK(hipMemcpy(geo, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
  auto kern = k_conv16_layer<ConnectFour, F, STATS, true>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("k_conv16_layer<ConnectFour, 128, STATS = %d>, %d workgroups, %d bytes of LDS\n", (int)STATS, nwg, (int)T::BYTES);
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(T::THREADS), T::BYTES, 0, in, (const float4*)w, out, B, geo, part, (const float*)nullptr, st, BnIn{}, TrFinal{});
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> s((size_t)nwg * 4);
    CK(hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = s[0], tend = 0; double av[4] = {0, 0, 0, 0}, late = 0;   // q[2] -> q[3] contains a barrier: wavefront 0's products, then everyone's
    for (int w_ = 0; w_ < nwg; ++w_) { t0 = std::min(t0, s[w_ * 4]); tend = std::max(tend, s[w_ * 4 + 3]); }
    for (int w_ = 0; w_ < nwg; ++w_) {
      const long long* q = &s[(size_t)w_ * 4];
      av[0] += q[1] - q[0]; av[1] += q[2] - q[1]; av[2] += q[3] - q[2]; av[3] += q[3] - q[0]; late = std::max(late, (double)(q[0] - t0));
    }
    const double tick = 0.01;
    printf("launch %d: %.1f us by events | first start -> last end %.1f us, latest start +%.1f us | per workgroup (mean, us): fill %.1f, products of wavefront 0 %.1f, wait for the other wavefronts + epilogue %.1f, total %.1f\n",
           it, ms * 1e3, (tend - t0) * tick, late * tick, av[0] / nwg * tick, av[1] / nwg * tick, av[2] / nwg * tick, av[3] / nwg * tick);
  }
  return 0;
}
int main() { return run<false>() || run<true>(); }
