// Micro-benchmark: how fast can a 4-wave workgroup stream its 128-token x 256-channel NCHW tile
// (row segments of 512 B, 4 KiB apart) with different load shapes?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D 256
#define HW 1024
// (a) per-lane dword loads, ALL in flight, lane=(token c, half h), channels 16s+8h+j
template <int WPS, int BATCH>
__global__ __launch_bounds__(256, WPS) void k_dword(const float* __restrict__ z, float* __restrict__ out, long N) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  long n = ((long)blockIdx.x * 4 + wave) * 32 + c;
  long b = n / HW; int hw = n % HW;
  const float* zp = z + ((size_t)b * D + 8 * h) * HW + hw;
  float acc = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < 128; s0 += BATCH) {
    float v[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) { int q = s0 + i; v[i] = zp[(size_t)(16 * (q >> 3) + (q & 7)) * HW]; }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) acc += v[i];
    if (BATCH < 128) { unsigned dep; asm volatile("v_mov_b32 %0, 0" : "=v"(dep) : "v"(acc)); zp += dep; }
  }
  if (acc == 12345.678f) out[n] = acc;
}
// (b) dwordx4 row loads: wave-instr = 2 channel rows x 128 tokens (1 KiB), 32 instr per wave per tile
template <int WPS, int BATCH>
__global__ __launch_bounds__(256, WPS) void k_x4(const float* __restrict__ z, float* __restrict__ out, long N) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long tok0 = (long)blockIdx.x * 128;
  long b = tok0 / HW; int hw0 = tok0 % HW;
  // wave w handles channels [64w, 64w+64): instr i covers channels 64w+2i, +1
  const float* zp = z + ((size_t)b * D + 64 * wave + (lane >> 5)) * HW + hw0 + (lane & 31) * 4;
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int i0 = 0; i0 < 32; i0 += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) v[i] = *(const f32x4*)(zp + (size_t)2 * (i0 + i) * HW);
#pragma unroll
    for (int i = 0; i < BATCH; ++i) acc += v[i];
    if (BATCH < 32) { unsigned dep; asm volatile("v_mov_b32 %0, 0" : "=v"(dep) : "v"(acc[0])); zp += dep; }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[tok0] = acc[0];
}
// (c) global->LDS DMA, 16 B per lane, whole 128 KiB tile in 8 chunks of 16 KiB through 2 buffers
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_dma(const float* __restrict__ z, float* __restrict__ out, long N) {
  __shared__ __attribute__((aligned(16))) float buf[2][4096];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long tok0 = (long)blockIdx.x * 128;
  long b = tok0 / HW; int hw0 = tok0 % HW;
  float acc = 0.f;
  auto issue = [&](int ch) {   // chunk ch = channels [32ch, 32ch+32): 16 KiB = 16 wave-instr, 4 per wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row2 = wave * 4 + i;      // pair of channel rows
      const float* src = z + ((size_t)b * D + 32 * ch + 2 * row2 + (lane >> 5)) * HW + hw0 + (lane & 31) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(&buf[ch & 1][row2 * 256]), 16, 0, 0);
    }
  };
  issue(0);
  for (int ch = 0; ch < 8; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ch + 1 < 8) issue(ch + 1);
    for (int i = 0; i < 16; ++i) acc += buf[ch & 1][i * 256 + threadIdx.x];
  }
  if (acc == 12345.678f) out[tok0] = acc;
}
#define RUN(name, kern, blocks) { \
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, z, out, N); \
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, z, out, N); \
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
  printf("%-28s %8.1f us  %7.2f TB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e12); }
int main() {
  long B = 256, N = B * HW; size_t bytes = (size_t)N * D * 4;
  float *z, *out; hipMalloc(&z, bytes); hipMalloc(&out, N * 4); hipMemset(z, 0x3c, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int nb = N / 128;
  RUN("dword all-in-flight 2w/simd", (k_dword<2, 128>), nb);
  RUN("dword batch64 4w/simd", (k_dword<4, 64>), nb);
  RUN("dword batch32 4w/simd", (k_dword<4, 32>), nb);
  RUN("dword batch32 8w/simd", (k_dword<8, 32>), nb);
  RUN("dword batch16 8w/simd", (k_dword<8, 16>), nb);
  RUN("x4 all 32 in flight 2w", (k_x4<2, 32>), nb);
  RUN("x4 batch8 4w/simd", (k_x4<4, 8>), nb);
  RUN("x4 batch8 8w/simd", (k_x4<8, 8>), nb);
  RUN("x4 batch4 8w/simd", (k_x4<8, 4>), nb);
  RUN("dma 2x16KiB 2 wg/cu", (k_dma<2>), nb);
  RUN("dma 2x16KiB 4 wg/cu", (k_dma<4>), nb);
  return 0;
}
