// Probe: which address patterns make ds_read_b128 conflict on gfx950?  16 wavefronts of one workgroup saturate the LDS;
// lane (lrow = lane & 15, g = lane >> 4) reads 16 bytes at row(lrow) * STRIDE + g * GOFF, the towers' A-operand pattern.
// Measured (MI355X): 4.2 cycles per wave instruction = 256 B/clk for a linear pattern and for 288- or 544-byte rows with
// the four k groups 16 bytes apart; 8 cycles for 272-byte rows (any group offset) and for 288-byte rows with groups 32+
// bytes apart; k rows equal mod 16 (272) cost k times more.  So a pass covers the lanes (lrow 0..7 | 8..15) x (g, g + 1)
// and wants { (STRIDE / 16) lrow + g } distinct mod 16.  T16 (fp32) and T16B (bf16) lay their rows out accordingly
// (resnet16.h posF, resnet16b.h SH): Mancala 21.5 -> 25 M sims/s, bf16 10x128 4.40 -> 4.85 M.
//   hipcc --offload-arch=gfx950 -O2 -o lds_conflict lds_conflict.hip && ./lds_conflict
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_c(const int* __restrict__ byteoff, int reps, long long* cycles, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(lds) + byteoff[threadIdx.x & 63];
  float s = 0.f;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    f4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(a), "n"(q * 64));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q].x;
  }
  __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = s;
}
static void run(const char* name, const std::vector<int>& rows, int goff, int stride = 272) {
  std::vector<int> off(64);
  for (int l = 0; l < 64; ++l) off[l] = rows[l & 15] * stride + (l >> 4) * goff;
  int* d; long long* c; float* s; hipMalloc(&d, 256); hipMalloc(&c, 8); hipMalloc(&s, 8192);
  hipMemcpy(d, off.data(), 256, hipMemcpyHostToDevice);
  const int reps = 2048, waves = 16;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_c, dim3(1), dim3(64 * waves), 0, 0, d, reps, c, s);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-58s row stride %4d B, g offset %3d B: %.2f LDS cycles per wave instruction\n", name, stride, goff, (double)h / (reps * 8.0 * waves));
  hipFree(d); hipFree(c); hipFree(s);
}
int main() {
  std::vector<int> nat(16), same16(16), pair(16), mod8(16), perm(16);
  for (int i = 0; i < 16; ++i) { nat[i] = i; same16[i] = 16 * (i % 8) ; pair[i] = (i / 2) + 16 * (i % 2); mod8[i] = (i % 8) + 16 * (i / 8) ; }
  const int p[16] = {37, 2, 19, 52, 5, 22, 39, 8, 41, 58, 11, 28, 45, 14, 31, 48};      // distinct mod 16, scattered
  for (int i = 0; i < 16; ++i) perm[i] = p[i];
  for (int goff : {64, 16}) {
    run("rows 0..15 (consecutive)", nat, goff);
    run("16 scattered rows, all distinct mod 16", perm, goff);
    run("rows equal mod 16 in pairs (2 lanes per residue)", pair, goff);
    run("rows r and r + 8 share ... (distinct mod 16, equal mod 8)", mod8, goff);
    run("rows 0,16,32,.. (8 residues-0 twice each: all equal mod 16)", same16, goff);
  }
  for (int stride : {256, 272, 288, 304, 320, 336, 400, 528, 544, 1040})
    for (int goff : {16, 32, 64, 128}) run("sweep: rows 0..15", nat, goff, stride);
  std::vector<int> lin(16); for (int i = 0; i < 16; ++i) lin[i] = i;
  run("linear: lane l at 16 l bytes", lin, 256, 16);
  return 0;
}
