// Probe: what the matrix pipes sustain on the WHOLE chip for about a millisecond (the length of a tower launch), and at what
// clock: v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32, 256 CUs x 4 SIMDs x {1, 2} waves, 4 independent accumulators
// per wave; with and without LDS reads beside the MFMAs (the tower's A-operand stream).  The nominal peaks (2.5 PFLOP/s bf16,
// 157.3 TFLOP/s fp32) assume 2.4 GHz; under sustained matrix load the clock the chip holds is what bounds a real kernel.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_sustained mfma_sustained.hip && ./mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF16, bool LDS> __global__ void __launch_bounds__(512) k(int reps, long long* cyc, unsigned long long* wall, float* sink) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * i;
  __syncthreads();
  f32x4 acc[4];
  for (auto& a : acc) a = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(1e-3f * (threadIdx.x + i)); b8[i] = (__bf16)(1e-3f * (threadIdx.x - i)); }
  float af = 1e-3f * threadIdx.x, bf = 2e-3f * threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  int off = threadIdx.x * 4;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (LDS) {
        const float4 v = *(const float4*)&lds[(off + u * 2048) & 16380];
        af += v.x; bf += v.y;
        if (BF16) { a8[0] = (__bf16)v.z; b8[1] = (__bf16)v.w; }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (BF16) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[c], 0, 0, 0);
      }
    }
    off += 64;
  }
  const long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  float s = 0;
  for (auto& a : acc) s += a[0] + a[1] + a[2] + a[3];
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool BF16, bool LDS> void run(const char* name, int waves_per_simd, int reps) {
  const int blocks = 256, threads = 256 * waves_per_simd;
  long long* c; unsigned long long* w; float* s;
  hipMalloc(&c, 8 * blocks); hipMalloc(&w, 8 * blocks); hipMalloc(&s, 4 * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<BF16, LDS>), dim3(blocks), dim3(threads), 0, 0, reps / 8, c, w, s);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BF16, LDS>), dim3(blocks), dim3(threads), 0, 0, reps, c, w, s);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc; unsigned long long hw;
  hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); hipMemcpy(&hw, w, 8, hipMemcpyDeviceToHost);
  const double flop_per_mfma = BF16 ? 2.0 * 16 * 16 * 32 : 2.0 * 16 * 16 * 4;
  const double flops = (double)blocks * (threads / 64) * reps * 32.0 * flop_per_mfma;
  const double ghz = (double)hc / ((double)hw * 10.0);                 // wall_clock64 ticks at 100 MHz
  printf("%-44s %d wave(s)/SIMD%s: %.3f ms, %8.1f TFLOP/s, shader clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", name, waves_per_simd,
         LDS ? " + ds_read_b128 per 4 MFMAs" : "", ms, flops / (ms * 1e-3) / 1e12, ghz, (double)hc / (reps * 32.0 * waves_per_simd));
  hipFree(c); hipFree(w); hipFree(s);
}
int main() {
  run<true, false>("v_mfma_f32_16x16x32_bf16", 1, 6000); run<true, false>("v_mfma_f32_16x16x32_bf16", 2, 3000);
  run<true, true>("v_mfma_f32_16x16x32_bf16", 2, 3000);
  run<false, false>("v_mfma_f32_16x16x4_f32", 1, 3000); run<false, false>("v_mfma_f32_16x16x4_f32", 2, 1500);
  run<false, true>("v_mfma_f32_16x16x4_f32", 2, 1500);
  return 0;
}
