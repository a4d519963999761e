// Microbenchmark: cycles per ds_add_f32 wave-instruction for three address patterns (one workgroup of NW waves per CU):
//   0: lane = column, two rows per instruction (conflict-free by construction)
//   1: 64 random rows, column fixed, row stride 33 words (random banks)
//   2: 64 random rows, row stride 32 words (all lanes of equal row parity on one bank)
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_rate.hip -o /tmp/lds_atomic_rate && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void k(const int *rows, int iters, float *out, long long *cyc)
{
    extern __shared__ float tab[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024 * 33; i += blockDim.x) tab[i] = 0.0f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int r = rows[(blockIdx.x * blockDim.x + threadIdx.x) & 65535];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            int addr;
            if (MODE == 0) addr = ((r + (lane >> 5)) & 1023) * 32 + (lane & 31);
            else if (MODE == 1) addr = r * 33 + c;
            else addr = r * 32 + c;
            atomicAdd(tab + addr, 1.0f);
        }
        r = (r * 1103515245 + 12345 + it) & 1023;
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x];
}

int main()
{
    std::vector<int> h(65536);
    for (auto &v : h) v = rand() & 1023;
    int *rows; float *out; long long *cyc;
    hipMalloc(&rows, 65536 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(rows, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 64;
    for (int nw : {4, 16}) {
        for (int mode = 0; mode < 3; ++mode) {
            auto fn = mode == 0 ? k<0> : (mode == 1 ? k<1> : k<2>);
            hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 33 * 4);
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(fn, dim3(256), dim3(nw * 64), 1024 * 33 * 4, 0, rows, iters, out, cyc);
            hipDeviceSynchronize();
            long long c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
            printf("{\"waves\": %d, \"mode\": %d, \"cycles_per_ds_add_f32_instruction_per_CU\": %.1f}\n", nw, mode, avg / (double)(iters * 32 * nw));
        }
    }
    return 0;
}
