// Probe: LDS read throughput per instruction width (b32 / b64 / b128), linear addresses, 1..16 wavefronts of ONE
// workgroup (one CU); asm volatile loads, 8 in flight per wave; the clock stops after a workgroup barrier.
//   hipcc --offload-arch=gfx950 -O2 -o lds_width lds_width.hip && ./lds_width
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int W> __global__ void k_w(int reps, long long* cycles, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(lds) + (threadIdx.x & 63) * W * 4 + (threadIdx.x >> 6) * 64;
  float s = 0.f;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if constexpr (W == 4) {
      f4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(a), "n"(q * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q].x;
    } else if constexpr (W == 2) {
      f2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[q]) : "v"(a), "n"(q * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q].x;
    } else {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[q]) : "v"(a), "n"(q * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
  }
  __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = s;
}
template <int W> void run(const char* n) {
  long long* c; float* s; hipMalloc(&c, 8); hipMalloc(&s, 8192);
  for (int waves : {1, 4, 8, 16}) {
    const int reps = 2048;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_w<W>, dim3(1), dim3(64 * waves), 0, 0, reps, c, s);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (reps * 8.0);
    printf("%s %2d waves: %.1f cycles per instruction per wave -> %.0f B/clk per CU\n", n, waves, per, 64.0 * 4 * W * waves / per);
  }
}
int main() { run<1>("ds_read_b32 "); run<2>("ds_read_b64 "); run<4>("ds_read_b128"); return 0; }
