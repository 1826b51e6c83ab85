// Probe: bytes per clock a CU's vector memory path delivers when 8 wavefronts stream global_load_dwordx4 from a small region
// shared by ALL workgroups (weights: every CU reads the same 32 KB per tap): working sets of 16 KB (fits the 32 KB L1),
// 64 KB and 512 KB (L2).  Decides whether the bf16 tower may double its weight stream to halve its LDS reads.
//   hipcc --offload-arch=gfx950 -O2 -o l1_bw l1_bw.hip && ./l1_bw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(const float4* __restrict__ w, int words16, int reps, long long* cyc, float* sink) {
  float4 acc = make_float4(0, 0, 0, 0);
  const int tid = threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    int off = (tid + r * 512) % words16;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 v = w[(off + u * 64) % words16];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  float4* w; long long* c; float* s;
  (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20); (void)hipMalloc(&c, 8 * 256); (void)hipMalloc(&s, 4 * 256 * 512);
  for (int kb : {16, 32, 64, 512}) {
    const int words16 = kb * 1024 / 16, reps = 400;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, w, words16, reps, c, s);
    (void)hipDeviceSynchronize();
    long long h[256]; (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
    printf("working set %3d KB shared by all CUs: %.1f bytes per clock per CU (8 waves x %d x 8 x 1 KB in %.0f cycles)\n", kb, 512.0 * 16 * 8 * reps / mean, reps, mean);
  }
  return 0;
}
