// What fp16 matrix rate does the part SUSTAIN?  Nothing but MFMAs on register operands (no LDS, no memory in the loop), every CU busy,
// for ~2 s per case so that the clock governor settles; operands random, or zero (less switching).  The pass-1 code loop, the gate GEMM's
// matrix phase and the K = 16384 pass 1 all run at 1.3 - 1.4 PFLOP/s; this says how far that is from what the silicon gives under its
// power cap, as opposed to the 2.5 PFLOP/s of the data sheet (2.4 GHz x 256 CUs x 4 SIMDs x 1024 flop/clk).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_sustained.hip -o /tmp/mfma_sustained && /tmp/mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long c0, c1, r0, r1; };

// SHAPE 0: v_mfma_f32_16x16x32_f16 (16 KFLOP), 8 independent accumulators; SHAPE 1: v_mfma_f32_32x32x16_f16 (32 KFLOP), 4 accumulators
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_kernel(const f16x8 *__restrict__ src, int iters, float *__restrict__ out, Stamp *__restrict__ stamps)
{
    const int tid = threadIdx.x;
    f16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = src[((size_t)blockIdx.x * 16 + i) * 256 + tid];
        b[i] = src[((size_t)blockIdx.x * 16 + 8 + i) * 256 + tid];
    }
    Stamp st;
    st.c0 = __builtin_amdgcn_s_memtime();
    st.r0 = __builtin_amdgcn_s_memrealtime();
    float o = 0.0f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + k) & 7], b[k], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 7], b[k], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o += acc[i][0] + acc[i][15];
    }
    st.c1 = __builtin_amdgcn_s_memtime();
    st.r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * 256 + tid] = o;
    if (tid == 0) stamps[blockIdx.x] = st;
}

template <int SHAPE>
static void run(const char *name, const f16x8 *src, float *out, Stamp *stamps, int ncu, int wg_per_cu, double seconds)
{
    const int grid = ncu * wg_per_cu;
    const int iters = 20000;
    const double flop_per_launch = (double)grid * 4 * iters * 64 * (SHAPE == 0 ? 16384.0 : 32768.0) * (SHAPE == 0 ? 1.0 : 0.5);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_kernel<SHAPE>, dim3(grid), dim3(256), 0, 0, src, iters, out, stamps);
    (void)hipDeviceSynchronize();
    // run back to back for `seconds`, report the LAST launch (settled clock)
    float ms = 0, total = 0;
    int n = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_kernel<SHAPE>, dim3(grid), dim3(256), 0, 0, src, iters, out, stamps);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms;
        ++n;
    }
    std::vector<Stamp> h(grid);
    (void)hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    double clk = 0;
    for (auto &s : h) clk += (double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) * 10.0);
    printf("\"%s\": {\"launches\": %d, \"last_launch_ms\": %.3f, \"PFLOP_per_s\": %.3f, \"counter_GHz_mean\": %.3f}", name, n, ms,
           flop_per_launch / (ms * 1e-3) * 1e-15, clk / grid);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t n = (size_t)2 * ncu * 16 * 256;
    std::vector<_Float16> h(n * 8);
    srand(3);
    for (auto &v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    f16x8 *src, *zero; float *out; Stamp *stamps;
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&zero, n * 16); (void)hipMalloc(&out, (size_t)2 * ncu * 256 * 4); (void)hipMalloc(&stamps, 2 * ncu * sizeof(Stamp));
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    (void)hipMemset(zero, 0, n * 16);
    printf("{\"cus\": %d, \"seconds_per_case\": %.1f, ", ncu, seconds);
    run<0>("16x16x32_random_1_wave_per_simd", src, out, stamps, ncu, 1, seconds); printf(", ");
    run<0>("16x16x32_random_2_waves_per_simd", src, out, stamps, ncu, 2, seconds); printf(", ");
    run<1>("32x32x16_random_1_wave_per_simd", src, out, stamps, ncu, 1, seconds); printf(", ");
    run<1>("32x32x16_random_2_waves_per_simd", src, out, stamps, ncu, 2, seconds); printf(", ");
    run<0>("16x16x32_zeros_2_waves_per_simd", zero, out, stamps, ncu, 2, seconds); printf(", ");
    run<1>("32x32x16_zeros_2_waves_per_simd", zero, out, stamps, ncu, 2, seconds);
    printf("}\n");
    return 0;
}
