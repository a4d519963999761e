// Micro-benchmark of z_q store shapes into an NCHW [B, 256, 32, 32] tensor (268 MB): what does the store
// INSTRUCTION count cost at equal bytes, and do 8-byte pieces of one 128-B line written by two different waves
// merge in L2 (same workgroup, with / without skew; different workgroups)?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define D 256
#define HW 1024
// (1) dword per lane, lane = (token c, half h), channels 16s+8h+j: the legacy epilogue (128 instr / wave)
__global__ __launch_bounds__(256, 2) void s_dword(float* __restrict__ zq, float v) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  long n = ((long)blockIdx.x * 4 + wave) * 32 + c;
  long b = n / HW; int hw = n % HW;
  float* zp = zq + ((size_t)b * D + 8 * h) * HW + hw;
#pragma unroll
  for (int q = 0; q < 128; ++q) zp[(size_t)(16 * (q >> 3) + (q & 7)) * HW] = v + q;
}
// (2) dwordx4 rows: wave-instr = 2 channel rows x 128 tokens (1 KiB), 32 instr / wave
__global__ __launch_bounds__(256, 2) void s_x4(float* __restrict__ zq, float v) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long tok0 = (long)blockIdx.x * 128;
  long b = tok0 / HW; int hw0 = tok0 % HW;
  float* zp = zq + ((size_t)b * D + 64 * wave + (lane >> 5)) * HW + hw0 + (lane & 31) * 4;
#pragma unroll
  for (int i = 0; i < 32; ++i) { f32x4 x = {v, v + i, v, v}; *(f32x4*)(zp + (size_t)2 * i * HW) = x; }
}
// (3) dwordx2: wave-instr = 1 channel row x 128 tokens (512 B), 64 instr / wave
__global__ __launch_bounds__(256, 2) void s_x2(float* __restrict__ zq, float v) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long tok0 = (long)blockIdx.x * 128;
  long b = tok0 / HW; int hw0 = tok0 % HW;
  float* zp = zq + ((size_t)b * D + 64 * wave) * HW + hw0 + lane * 2;
#pragma unroll
  for (int i = 0; i < 64; ++i) { f32x2 x = {v, v + i}; *(f32x2*)(zp + (size_t)i * HW) = x; }
}
// (4) 8-byte cells of every line split between two waves: MODE 0 partner = the other wave of a pair in the same
// workgroup, same instruction index; 1 = same, partner delayed by `skew` 100-MHz ticks; 2 = partner in a workgroup half
// a grid away.  Each wave covers 64 tokens x 256 channels but writes only its parity of the 8-byte cells
// (lane -> cell 2*lane + parity): 256 instr / wave of 8 B per lane.
template <int MODE>
__global__ __launch_bounds__(256, 2) void s_split(float* __restrict__ zq, float v, int skew, int nblk) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int parity, pairwave; long blk;
  if (MODE == 2) { parity = (blockIdx.x >= (unsigned)nblk / 2); blk = blockIdx.x % (nblk / 2); pairwave = wave; }
  else { parity = wave & 1; blk = blockIdx.x; pairwave = wave >> 1; }
  // MODE 0/1: block = 128 tokens, wave pair p covers tokens [64p, 64p+64)... keep 128 tokens per block: pair covers 64
  // MODE 2: two blocks share 256 tokens: block pair covers tokens [256 blk', +256), wave w covers 64 of them
  long tok0 = (MODE == 2) ? blk * 256 + pairwave * 64 : blk * 128 + pairwave * 64;
  long b = tok0 / HW; int hw0 = tok0 % HW;
  if (MODE == 1 && parity) { unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)skew) __builtin_amdgcn_s_sleep(8); }
  // 64 tokens = 32 cells of 8 B per channel row; this wave writes cells of its parity: 16 cells per row -> one
  // instruction covers 4 channel rows (64 lanes x 8 B)
  float* zp = zq + ((size_t)b * D + (lane >> 4)) * HW + hw0 + ((lane & 15) * 2 + parity) * 2;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) { f32x2 x = {v, v + i}; *(f32x2*)(zp + (size_t)4 * i * HW) = x; }
}
#define RUN(name, blocks, ...) { \
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(__VA_ARGS__); \
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(__VA_ARGS__); \
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
  printf("%-44s %8.1f us  %7.2f TB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e12); }
int main() {
  long B = 256, N = B * HW; size_t bytes = (size_t)N * D * 4;
  float *zq; hipMalloc(&zq, bytes); hipMemset(zq, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int nb = N / 128;
  RUN("dword/lane (128 instr/wave)", nb, s_dword, dim3(nb), dim3(256), 0, 0, zq, 1.0f);
  RUN("dwordx2 rows (64 instr/wave)", nb, s_x2, dim3(nb), dim3(256), 0, 0, zq, 1.0f);
  RUN("dwordx4 rows (32 instr/wave)", nb, s_x4, dim3(nb), dim3(256), 0, 0, zq, 1.0f);
  RUN("8B cells split, same WG, no skew", nb, s_split<0>, dim3(nb), dim3(256), 0, 0, zq, 1.0f, 0, nb);
  RUN("8B cells split, same WG, 1 us skew", nb, s_split<1>, dim3(nb), dim3(256), 0, 0, zq, 1.0f, 100, nb);
  RUN("8B cells split, same WG, 3 us skew", nb, s_split<1>, dim3(nb), dim3(256), 0, 0, zq, 1.0f, 300, nb);
  RUN("8B cells split, same WG, 10 us skew", nb, s_split<1>, dim3(nb), dim3(256), 0, 0, zq, 1.0f, 1000, nb);
  RUN("8B cells split, other WG half a grid away", nb, s_split<2>, dim3(nb), dim3(256), 0, 0, zq, 1.0f, 0, nb);
  return 0;
}
