// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for k_tree's access patterns (MI355X_MICROARCH.md: "FETCH_SIZE
// reports exactly half of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Four kernels over a 4 GB buffer (far beyond L2 + Infinity Cache), each touching
// a KNOWN number of bytes, every 128-byte line at most once:
//   k_stream      16 B per lane, consecutive (the guide's calibrated case)
//   k_line128     random 128-byte lines, 8 lanes x 16 B per line (a node record read by a slot's lane group)
//   k_sector32    random lines, ONE 32-byte sector of each (2 lanes x 16 B) (a side record / a slot-state word)
//   k_half64      random lines, the first 64 bytes of each (4 lanes x 16 B) (the packed 64-byte slot record)
//   k_write32     random lines, one 32-byte sector of each written (a backup's read-modify-write leaves one dirty sector)
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE`; the program prints the bytes each kernel asked for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// line number of group g: a bijection on [0, nlines) (nlines a power of two): odd multiplier + xor
__device__ __forceinline__ uint64_t line_of(uint64_t g, uint64_t nlines) { return ((g * 0x9E3779B97F4A7C15ULL) ^ (g >> 7)) & (nlines - 1); }

__global__ void k_stream(const uint4* __restrict__ b, uint4* __restrict__ sink, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 a = make_uint4(0, 0, 0, 0);
  if (i < n16) a = b[i];
  if (a.x == 0x12345678u && a.y == 0x9abcdef0u) sink[0] = a;
}
template <int LANES> __global__ void k_rand(const char* __restrict__ b, uint4* __restrict__ sink, uint64_t groups, uint64_t nlines) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, g = t / LANES;
  uint4 a = make_uint4(0, 0, 0, 0);
  if (g < groups) a = *(const uint4*)(b + line_of(g, nlines) * 128 + (t % LANES) * 16);
  if (a.x == 0x12345678u && a.y == 0x9abcdef0u) sink[0] = a;
}
__global__ void k_write32(char* __restrict__ b, uint64_t groups, uint64_t nlines) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, g = t / 2;
  if (g < groups) *(uint4*)(b + line_of(g, nlines) * 128 + (t % 2) * 16) = make_uint4((unsigned)t, 1, 2, 3);
}

int main() {
  const size_t bytes = (size_t)4 << 30;
  const uint64_t nlines = bytes / 128;                               // 2^25
  char* buf; uint4* sink;
  CHK(hipMalloc((void**)&buf, bytes)); CHK(hipMalloc((void**)&sink, 64));
  CHK(hipMemset(buf, 0, bytes));
  CHK(hipDeviceSynchronize());
  const uint64_t groups = 1 << 22;                                   // 4 M lines touched per random kernel (of 32 M)
  const size_t n16 = (size_t)1 << 26;                                // 1 GB streamed
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_stream, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const uint4*)buf, sink, n16);
    hipLaunchKernelGGL((k_rand<8>), dim3((unsigned)(groups * 8 / 256)), dim3(256), 0, 0, buf, sink, groups, nlines);
    hipLaunchKernelGGL((k_rand<2>), dim3((unsigned)(groups * 2 / 256)), dim3(256), 0, 0, buf, sink, groups, nlines);
    hipLaunchKernelGGL((k_rand<4>), dim3((unsigned)(groups * 4 / 256)), dim3(256), 0, 0, buf, sink, groups, nlines);
    hipLaunchKernelGGL(k_write32, dim3((unsigned)(groups * 2 / 256)), dim3(256), 0, 0, buf, groups, nlines);
    CHK(hipDeviceSynchronize());
  }
  printf("asked: k_stream %zu B | k_rand<8> (128-B lines) %llu B | k_rand<2> (32-B sectors) %llu B | k_rand<4> (64 B) %llu B | k_write32 %llu B\n",
         n16 * 16, (unsigned long long)(groups * 128), (unsigned long long)(groups * 32), (unsigned long long)(groups * 64), (unsigned long long)(groups * 32));
  return 0;
}
