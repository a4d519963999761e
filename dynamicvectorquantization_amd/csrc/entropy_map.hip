// entropy_map.hip -- fused patch-entropy map for gfx950.
//
// Replaces Entropy.forward of the reference (models/stage1_dynamic/dqvae_dual_entropy.py:13-63):
// grayscale -> 16x16 unfold -> 32-bin Gaussian-KDE histogram over [0, 1] (sigma 0.01) -> normalise
// (+1e-40) -> -sum p ln p.  The reference materialises a [B*256, 256, 32] fp32 tensor (2.1 GB at
// B = 256) and makes several passes over it; here one wave owns one patch: its 256 gray values go to
// LDS once, lane (bin, half) accumulates the kernel values of 128 pixels for its bin, and the
// 32-bin reductions are wave shuffles.  The image is read from HBM exactly once (768 KiB / image);
// the 8192 exp per patch make the kernel VALU/transcendental-bound, not HBM-bound.
// fp32 subnormals stay enabled (hipcc default): the reference's epsilon 1e-40 is a subnormal.
// Transcendental math -> parity is to 1e-5, not bit-exact (tests/test_entropy.py).
#include "dvq_common.h"

__global__ __launch_bounds__(256) void entropy_map_kernel(const float *__restrict__ img, int B, int H, int W,
                                                          float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float gray[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gh = H / 16, gw = W / 16;
    const long patch = (long)blockIdx.x * 4 + wave;
    const long npatch = (long)B * gh * gw;
    if (patch < npatch) {
        const int b = (int)(patch / (gh * gw));
        const int pr = (int)(patch - (long)b * gh * gw);
        const int py = pr / gw, px = pr - py * gw;
        // lane -> row lane>>2, columns 4(lane&3) .. +3 of the patch (one 16-B load per channel)
        const size_t off = ((size_t)(py * 16 + (lane >> 2))) * W + px * 16 + (lane & 3) * 4;
        const size_t plane = (size_t)H * W;
        const float *p = img + (size_t)b * 3 * plane + off;
        const f32x4 r = *(const f32x4 *)p, g = *(const f32x4 *)(p + plane), bl = *(const f32x4 *)(p + 2 * plane);
        f32x4 gy;
#pragma unroll
        for (int j = 0; j < 4; ++j)      // 0.2989 R + 0.5870 G + 0.1140 B, left to right, fp32 (:51)
            gy[j] = __fadd_rn(__fadd_rn(__fmul_rn(0.2989f, r[j]), __fmul_rn(0.5870f, g[j])), __fmul_rn(0.1140f, bl[j]));
        *(f32x4 *)&gray[wave][lane * 4] = gy;
    }
    __syncthreads();
    if (patch >= npatch) return;
    const int bin = lane & 31, half = lane >> 5;
    const float center = (float)bin * (1.0f / 31.0f);            // torch.linspace(0, 1, 32)
    const float inv_sigma = 1.0f / 0.01f;
    float acc = 0.0f;
    const float *gp = &gray[wave][half * 128];
#pragma unroll 4
    for (int i = 0; i < 128; i += 4) {
        const f32x4 v = *(const f32x4 *)(gp + i);                // same address in all lanes of a half: broadcast
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = (v[j] - center) * inv_sigma;
            acc += expf(-0.5f * t * t);
        }
    }
    acc += __shfl_xor(acc, 32);
    float pdf = acc * (1.0f / 256.0f);                           // mean over the 256 pixels (:40)
    float norm = pdf;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) norm += __shfl_xor(norm, o);
    norm += 1e-40f;                                              // (:41) epsilon is an fp32 subnormal
    pdf = pdf / norm + 1e-40f;                                   // (:42)
    float term = pdf * logf(pdf);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) term += __shfl_xor(term, o);
    if (lane == 0) out[patch] = -term;                           // (:43)
}

int dvq_launch_entropy_map(const float *img, int B, int H, int W, float *out, hipStream_t st)
{
    const long npatch = (long)B * (H / 16) * (W / 16);
    hipLaunchKernelGGL(entropy_map_kernel, dim3((unsigned)((npatch + 3) / 4)), dim3(256), 0, st, img, B, H, W, out);
    return (int)hipGetLastError();
}
