// Probe: issue interval of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 for ONE wavefront when every MFMA depends on
// the previous one's accumulator (the dense heads: one ascending-k chain per output) and with 2 / 4 independent chains.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_latency mfma_latency.hip && ./mfma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS> __global__ void k16(int reps, float a, float b, long long* cycles, float* sink) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float av = a + threadIdx.x, bv = b - threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = s;
}
template <int CHAINS> __global__ void k32(int reps, float a, float b, long long* cycles, float* sink) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float av = a + threadIdx.x, bv = b - threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = s;
}
template <class K> void run(const char* name, K kern, int chains) {
  long long* c; float* s; hipMalloc(&c, 8); hipMalloc(&s, 256);
  const int reps = 1000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, reps, 1e-3f, 1e-3f, c, s);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%s, %d chain(s): %.1f cycles per MFMA\n", name, chains, (double)h / (reps * 16.0 * chains));
}
int main() {
  run("v_mfma_f32_16x16x4_f32", k16<1>, 1); run("v_mfma_f32_16x16x4_f32", k16<2>, 2); run("v_mfma_f32_16x16x4_f32", k16<4>, 4);
  run("v_mfma_f32_32x32x2_f32", k32<1>, 1); run("v_mfma_f32_32x32x2_f32", k32<2>, 2);
  return 0;
}
