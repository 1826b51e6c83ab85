// Probe: is v_mfma_f32_16x16x4_f32 a k-ordered fmaf chain (k = lane>>4 = 0,1,2,3), like 32x32x2?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D, int steps) {
  // A[steps][16 rows][4 k], B[steps][4 k][16 cols], C[16][16]
  int l = threadIdx.x;
  f32x4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) * 4 + r) * 16 + (l & 15)];
  for (int s = 0; s < steps; ++s) {
    float a = A[(s * 16 + (l & 15)) * 4 + (l >> 4)];
    float b = B[(s * 4 + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
int main() {
  const int steps = 37;
  size_t na = steps * 64, nb = steps * 64;
  float *hA = (float*)malloc(na * 4), *hB = (float*)malloc(nb * 4), hC[256], hD[256], ref[256];
  srand(1);
  for (size_t i = 0; i < na; ++i) hA[i] = (float)rand() / RAND_MAX * 2 - 1;
  for (size_t i = 0; i < nb; ++i) hB[i] = (float)rand() / RAND_MAX * 2 - 1;
  for (int i = 0; i < 256; ++i) hC[i] = (float)rand() / RAND_MAX;
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, na * 4); hipMalloc(&dB, nb * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, na * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, nb * 4, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, steps);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float acc = hC[i * 16 + j];
    for (int s = 0; s < steps; ++s) for (int kk = 0; kk < 4; ++kk) acc = fmaf(hA[(s * 16 + i) * 4 + kk], hB[(s * 4 + kk) * 16 + j], acc);
    ref[i * 16 + j] = acc;
  }
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) bad++;
  printf("16x16x4 vs k-ordered fmaf chain: %d / 256 differ\n", bad);
  return bad != 0;
}
