// Energy per byte of pass 1's memory phases by ACCESS SHAPE.  Pass 1 runs at the socket power cap, so its time is its energy over
// the cap, and by the copy kernel's power (1176 W at 4.9 TB/s) its 643 MB of traffic is the largest single share.  Does that
// share depend on how the bytes are asked for?  Streaming kernels over the B = 256 latents (268 MB, random values), each run back to
// back for ~3 s while tools/micro/run_mem_energy.sh samples rocm-smi:
//   read  dword    pass 1's prologue: lane = (token, channel half), 128 x 4-B loads per lane, nt
//   read  x4       16 B per lane: a wave-instruction covers 2 channel rows x 128 tokens
//   read  dma      global_load_lds_dwordx4 into a double-buffered 16-KiB LDS tile, summed from LDS
//   write dword    pass 1's epilogue: 128 x 4-B nt stores per lane
//   write x4       16-B nt stores
//   copy  dword / x4   read + write
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mem_energy.hip -o tools/micro/mem_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D 256
#define HW 1024

#define LD(NT, p) ((NT) ? __builtin_nontemporal_load(p) : *(p))
template <int MODE, bool NT = true>   // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256, 2) void k_dword(const float *__restrict__ z, float *__restrict__ out, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const long n = ((long)blockIdx.x * 4 + wave) * 32 + c;
    const long b = n / HW;
    const int hw = (int)(n % HW);
    const size_t base = ((size_t)b * D + 8 * h) * HW + hw;
    float v[128];
    if (MODE != 1) {
#pragma unroll
        for (int q = 0; q < 128; ++q) v[q] = LD(NT, z + base + (size_t)(16 * (q >> 3) + (q & 7)) * HW);
    } else {
#pragma unroll
        for (int q = 0; q < 128; ++q) v[q] = (float)(lane + q);
    }
    if (MODE != 0) {
#pragma unroll
        for (int q = 0; q < 128; ++q) __builtin_nontemporal_store(v[q], out + base + (size_t)(16 * (q >> 3) + (q & 7)) * HW);
    } else {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 128; ++q) acc += v[q];
        if (acc == 12345.678f) sink[n] = acc;
    }
}

template <int MODE, bool NT = true>
__global__ __launch_bounds__(256, 2) void k_x4(const float *__restrict__ z, float *__restrict__ out, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tok0 = (long)blockIdx.x * 128;
    const long b = tok0 / HW;
    const int hw0 = (int)(tok0 % HW);
    const size_t base = ((size_t)b * D + 64 * wave + (lane >> 5)) * HW + hw0 + (lane & 31) * 4;
    f32x4 v[32];
    if (MODE != 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = LD(NT, (const f32x4 *)(z + base + (size_t)2 * i * HW));
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = f32x4{(float)lane, (float)i, 1.f, 2.f};
    }
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) __builtin_nontemporal_store(v[i], (f32x4 *)(out + base + (size_t)2 * i * HW));
    } else {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += v[i];
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[tok0] = acc[0];
    }
}

template <int AUX>
__global__ __launch_bounds__(256, 2) void k_dma(const float *__restrict__ z, float *__restrict__ out, float *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) float buf[2][4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tok0 = (long)blockIdx.x * 128;
    const long b = tok0 / HW;
    const int hw0 = (int)(tok0 % HW);
    float acc = 0.f;
    auto issue = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row2 = wave * 4 + i;
            const float *src = z + ((size_t)b * D + 32 * ch + 2 * row2 + (lane >> 5)) * HW + hw0 + (lane & 31) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(&buf[ch & 1][row2 * 256]), 16, 0, AUX);
        }
    };
    issue(0);
    for (int ch = 0; ch < 8; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ch + 1 < 8) issue(ch + 1);
        for (int i = 0; i < 16; ++i) acc += buf[ch & 1][i * 256 + threadIdx.x];
    }
    if (acc == 12345.678f) sink[tok0] = acc;
}

// read dma4: pass 1's lane <-> data mapping kept (lane = token, channel half), every access a 4-B-per-lane LDS-DMA into a wave-private,
// double-buffered 4-KiB chunk (16 channels of the lane's half ... one k-step), read back by the lane that asked for it: no barrier
__global__ __launch_bounds__(256, 2) void k_dma4(const float *__restrict__ z, float *__restrict__ out, float *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) float buf[4][2][16 * 64];      // [wave][buffer][instruction][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const long n = ((long)blockIdx.x * 4 + wave) * 32 + c;
    const long b = n / HW;
    const int hw = (int)(n % HW);
    const float *zp = z + ((size_t)b * D + 8 * h) * HW + hw;
    float acc = 0.f;
    auto issue = [&](int ch) {                                // k-steps 2 ch, 2 ch + 1: 16 accesses
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int s = 2 * ch + (q >> 3), j = q & 7;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(zp + (size_t)(16 * s + j) * HW),
                                             (__attribute__((address_space(3))) void *)(&buf[wave][ch & 1][q * 64]), 4, 0, 2);
        }
    };
    issue(0);
    for (int ch = 0; ch < 8; ++ch) {
        if (ch + 1 < 8) { issue(ch + 1); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += buf[wave][ch & 1][q * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the buffer is re-filled two chunks on
    }
    if (acc == 12345.678f) sink[n] = acc;
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "read_dword";
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    const long B = 256, N = B * HW;
    const size_t bytes = (size_t)N * D * 4;
    // NROT copies of the latents / outputs, used in rotation: the 256-MB memory-side cache cannot serve a launch from the previous one
    const int NROT = 4;
    float *zs[NROT], *outs[NROT], *sink;
    for (int r = 0; r < NROT; ++r) { (void)hipMalloc(&zs[r], bytes); (void)hipMalloc(&outs[r], bytes); }
    (void)hipMalloc(&sink, N * 4);
    {
        std::vector<float> h(bytes / 4);
        unsigned s = 12345u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        for (int r = 0; r < NROT; ++r) (void)hipMemcpy(zs[r], h.data(), bytes, hipMemcpyHostToDevice);
    }
    const int nb = (int)(N / 128);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int rot = 0;
    auto launch = [&]() {
        float *z = zs[rot], *out = outs[rot];
        rot = (rot + 1) % NROT;
        if (!strcmp(which, "read_dword")) hipLaunchKernelGGL(k_dword<0>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "write_dword")) hipLaunchKernelGGL(k_dword<1>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "copy_dword")) hipLaunchKernelGGL(k_dword<2>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "read_x4")) hipLaunchKernelGGL(k_x4<0>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "write_x4")) hipLaunchKernelGGL(k_x4<1>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "copy_x4")) hipLaunchKernelGGL(k_x4<2>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "read_dma4")) hipLaunchKernelGGL(k_dma4, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "read_dword_plain")) hipLaunchKernelGGL((k_dword<0, false>), dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "read_x4_plain")) hipLaunchKernelGGL((k_x4<0, false>), dim3(nb), dim3(256), 0, 0, z, out, sink);
        else if (!strcmp(which, "read_dma_nt")) hipLaunchKernelGGL(k_dma<2>, dim3(nb), dim3(256), 0, 0, z, out, sink);
        else hipLaunchKernelGGL(k_dma<0>, dim3(nb), dim3(256), 0, 0, z, out, sink);
    };
    for (int i = 0; i < 5; ++i) launch();
    (void)hipDeviceSynchronize();
    float total = 0, ms = 0;
    long n = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms;
        n += 200;
    }
    const double per = ms / 200 * 1e-3;
    const double moved = (strstr(which, "copy") ? 2.0 : 1.0) * (double)bytes;
    printf("{\"case\": \"%s\", \"us_per_launch\": %.2f, \"TB_per_s\": %.3f, \"bytes_per_launch\": %.0f}\n", which, per * 1e6, moved / per * 1e-12, moved);
    return 0;
}
