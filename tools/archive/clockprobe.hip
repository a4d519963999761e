// clockprobe.hip -- sustained shader clock under back-to-back fp16 MFMAs (and idle-ish VALU) on gfx950.
// s_memtime counts shader clocks, s_memrealtime a constant 100 MHz clock: their ratio over a long loop
// is the average clock the wave actually ran at.   hipcc --offload-arch=gfx950 -O3 clockprobe.hip -o clockprobe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, int iters)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc0 = {}, acc1 = {};
    float v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
            }
        } else if (KIND == 2) {                       // ONE dependent accumulator chain
#pragma unroll
            for (int u = 0; u < 16; ++u) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (acc0[0] + acc1[3] + v == 12345.678f) out[0] = 0;
}

#include <cstdlib>
int main(int argc, char **argv)
{
    const int blocks = 2048, iters = 20000;
    const int wgs = argc > 1 ? atoi(argv[1]) : 2048;   // kind 2: 256 = one wave per SIMD, 512 = two, ...
    unsigned long long *d, *h = new unsigned long long[2 * blocks];
    hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks);
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            else if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
            double sc = 0, rc = 0;
            for (int i = 0; i < (kind == 2 ? wgs : blocks); ++i) { sc += h[2 * i]; rc += h[2 * i + 1]; }
            double ghz = sc / rc * 0.1;
            double mfma_per_simd = (double)(kind == 2 ? wgs : blocks) * 4 /*waves*/ * iters * 16.0 / 1024.0;
            printf("%s rep %d: %.2f ms, memtime/memrealtime -> %.3f GHz (if memtime ticks at shader clock)", kind == 0 ? "mfma 2 chains" : (kind == 2 ? "mfma 1 chain " : "valu"), rep, ms, ghz);
            if (kind != 1) printf(", %.1f ns per MFMA per SIMD = %.2f GHz at 32 cycles/MFMA", ms * 1e6 / mfma_per_simd, 32.0 / (ms * 1e6 / mfma_per_simd));
            printf("\n");
        }
    }
    return 0;
}
