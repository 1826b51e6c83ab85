// vmm_probe.hip -- can a node pool be a big VIRTUAL range whose physical pages are mapped on demand (hipMemAddressReserve /
// hipMemCreate / hipMemMap)?  Measures the granularity, the cost of a map + set-access call, that kernels can run on mapped
// parts while other parts are being mapped, and that an unmapped part of the range costs no HBM.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void touch(unsigned long long* p, size_t stride_words, int n, unsigned long long v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[(size_t)i * stride_words] = v + i;
}
__global__ void check(const unsigned long long* p, size_t stride_words, int n, unsigned long long v, int* bad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && p[(size_t)i * stride_words] != v + i) atomicAdd(bad, 1);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
  size_t free0, total;
  CK(hipMemGetInfo(&free0, &total));
  const size_t VA = (size_t)192 << 30;
  void* base = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  CK(hipMemAddressReserve(&base, VA, grec, nullptr, 0));
  auto t1 = std::chrono::steady_clock::now();
  size_t free1;
  CK(hipMemGetInfo(&free1, &total));
  printf("reserved %zu GB of VA in %.1f us; free HBM %.2f -> %.2f GB\n", VA >> 30, std::chrono::duration<double, std::micro>(t1 - t0).count(), free0 / 1e9, free1 / 1e9);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const int N = 2048;
  const size_t chunk = grec, stride = VA / N / chunk * chunk;
  std::vector<hipMemGenericAllocationHandle_t> h(N);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) {
    CK(hipMemCreate(&h[i], chunk, &prop, 0));
    CK(hipMemMap((char*)base + (size_t)i * stride, chunk, 0, h[i], 0));
    CK(hipMemSetAccess((char*)base + (size_t)i * stride, chunk, &acc, 1));
    if (i == N / 2) {   // a kernel on the first half while the second half is still being mapped
      hipLaunchKernelGGL(touch, dim3((N / 2 + 255) / 256), dim3(256), 0, 0, (unsigned long long*)base, stride / 8, N / 2, 77ull);
    }
  }
  t1 = std::chrono::steady_clock::now();
  printf("%d x (create + map + set-access) of %zu KB: %.1f us each\n", N, chunk >> 10, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(touch, dim3((N + 255) / 256), dim3(256), 0, 0, (unsigned long long*)base + (chunk - 8) / 8, stride / 8, N, 5ull);
  int* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(check, dim3((N / 2 + 255) / 256), dim3(256), 0, 0, (const unsigned long long*)base, stride / 8, N / 2, 77ull, bad);
  hipLaunchKernelGGL(check, dim3((N + 255) / 256), dim3(256), 0, 0, (const unsigned long long*)base + (chunk - 8) / 8, stride / 8, N, 5ull, bad);
  int hb = -1; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  size_t free2;
  CK(hipMemGetInfo(&free2, &total));
  printf("kernels on the mapped chunks: %d mismatches; free HBM now %.2f GB (mapped %.2f GB)\n", hb, free2 / 1e9, N * (double)chunk / 1e9);
  // one batched set-access over a contiguous run of chunks
  const int M = 512;
  std::vector<hipMemGenericAllocationHandle_t> h2(M);
  char* run = (char*)base + (size_t)(N - 1) * stride + chunk;
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < M; ++i) { CK(hipMemCreate(&h2[i], chunk, &prop, 0)); CK(hipMemMap(run + (size_t)i * chunk, chunk, 0, h2[i], 0)); }
  CK(hipMemSetAccess(run, (size_t)M * chunk, &acc, 1));
  t1 = std::chrono::steady_clock::now();
  printf("%d contiguous chunks, one set-access: %.1f us per chunk\n", M, std::chrono::duration<double, std::micro>(t1 - t0).count() / M);
  // bigger handles at 2 MB-aligned addresses: what one call costs as a function of its size
  char* bigp = (char*)(((uintptr_t)(run + (size_t)M * chunk) + ((size_t)1 << 30)) & ~(((uintptr_t)1 << 30) - 1));
  const size_t sizes[] = {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)16 << 30};
  hipMemGenericAllocationHandle_t big[4];
  size_t boff[4];
  size_t off = 0;
  for (int k = 0; k < 4; ++k) {
    boff[k] = off;
    printf("handle of %zu MB at +%zu MB ... ", sizes[k] >> 20, off >> 20);
    t0 = std::chrono::steady_clock::now();
    CK(hipMemCreate(&big[k], sizes[k], &prop, 0));
    auto ta = std::chrono::steady_clock::now();
    CK(hipMemMap(bigp + off, sizes[k], 0, big[k], 0));
    auto tb = std::chrono::steady_clock::now();
    CK(hipMemSetAccess(bigp + off, sizes[k], &acc, 1));
    t1 = std::chrono::steady_clock::now();
    printf("create %.1f us, map %.1f us, set-access %.1f us\n", std::chrono::duration<double, std::micro>(ta - t0).count(),
           std::chrono::duration<double, std::micro>(tb - ta).count(), std::chrono::duration<double, std::micro>(t1 - tb).count());
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, 0, (unsigned long long*)(bigp + off), sizes[k] / 8 / 64, 64, 9ull);
    CK(hipDeviceSynchronize());
    off += sizes[k];
  }
  {
    size_t f; CK(hipMemGetInfo(&f, &total));
    printf("free HBM now %.2f GB\n", f / 1e9);
  }
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) { CK(hipMemUnmap((char*)base + (size_t)i * stride, chunk)); CK(hipMemRelease(h[i])); }
  for (int i = 0; i < M; ++i) { CK(hipMemUnmap(run + (size_t)i * chunk, chunk)); CK(hipMemRelease(h2[i])); }
  for (int k = 0; k < 4; ++k) { CK(hipMemUnmap(bigp + boff[k], sizes[k])); CK(hipMemRelease(big[k])); }
  CK(hipMemAddressFree(base, VA));
  t1 = std::chrono::steady_clock::now();
  size_t free3;
  CK(hipMemGetInfo(&free3, &total));
  printf("unmap + release + free: %.1f ms; free HBM %.2f GB\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), free3 / 1e9);
  return 0;
}
