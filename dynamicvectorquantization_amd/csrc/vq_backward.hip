// vq_backward.hip -- backward of the quantizer's straight-through forward with respect to its input (gfx950).
//
// What autograd derives from the reference graph (modules/vector_quantization/quantize2_mask.py:172-182,
// quantize_vqgan.py:290-298): identity through z + (z_q - z).detach(), plus the commitment term's
//   d loss / d z = g_loss * (2 c / numel) * (z - e) * m        (c = beta, or 1 with legacy = True; m = codebook_mask)
// with e = the codebook row chosen at FORWARD time (a 1-MiB snapshot of the codebook is kept for backward: the EMA update
// overwrites the weight in place between forward and backward).  The reference runs this as five or six element-wise
// passes over the [B, D, H, W] tensor plus a transposed copy of e; here it is one streaming pass: read z and g_zq, gather
// e from the snapshot (L2 resident), write g_z -- 3 x 4 bytes per element, HBM-bound.
// Same fp32 operation order as the torch expression it replaces: g_zq + (g_loss * fl(2 c / numel)) * ((z - e) * m).
//
// Mapping as pass 1 of the assign: a wave owns 32 consecutive tokens, lane = (token, half); per k-step of 16 channels a
// lane handles 8 (two 16-byte gathers of its code's row, 8 loads / stores whose wave instructions cover 128-byte runs).
#include "dvq_common.h"

__global__ __launch_bounds__(256) void vq_backward_z_kernel(
    const float *__restrict__ z, const float *__restrict__ E, const long long *__restrict__ codes,
    const float *__restrict__ mask, const float *__restrict__ g_zq, const float *__restrict__ g_loss, float coef_scale,
    int D, int HW, int K, long N, float *__restrict__ gz)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const long n = ((long)blockIdx.x * 4 + wave) * 32 + c;
    if (n >= N) return;
    const long b = n / HW;
    const int hw = (int)(n - b * HW);
    const size_t base = ((size_t)b * D + 8 * h) * HW + hw;
    long long cj = codes[n];
    const bool ok = cj >= 0 && cj < K;                       // (the forward writes valid codes; anything else: no loss term)
    const float *ep = E + (size_t)(ok ? cj : 0) * D + 8 * h;
    const float m = (mask != nullptr) ? mask[n] : 1.0f;
    const float c0 = (g_loss != nullptr && ok) ? __fmul_rn(g_loss[0], coef_scale) : 0.0f;
    const float *zp = z + base;
    const float *gp = (g_zq != nullptr) ? g_zq + base : nullptr;
    float *op = gz + base;
    const int S16 = D / 16;
#pragma unroll 2
    for (int s = 0; s < S16; ++s) {
        const f32x4 e0 = *(const f32x4 *)(ep + 16 * s), e1 = *(const f32x4 *)(ep + 16 * s + 4);
        float zz[8], gg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            zz[j] = __builtin_nontemporal_load(zp + (size_t)(16 * s + j) * HW);
            gg[j] = (gp != nullptr) ? __builtin_nontemporal_load(gp + (size_t)(16 * s + j) * HW) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = (j < 4) ? e0[j & 3] : e1[j & 3];
            const float diff = __fmul_rn(__fsub_rn(zz[j], e), m);
            __builtin_nontemporal_store(__fadd_rn(gg[j], __fmul_rn(c0, diff)), op + (size_t)(16 * s + j) * HW);
        }
    }
}

int dvq_launch_vq_backward_z(const float *z, const float *E, const long long *codes, const float *mask, const float *g_zq,
                             const float *g_loss, float coef_scale, int D, int HW, int K, long N, float *gz, hipStream_t st)
{
    hipLaunchKernelGGL(vq_backward_z_kernel, dim3((unsigned)((N + 127) / 128)), dim3(256), 0, st, z, E, codes, mask, g_zq, g_loss,
                       coef_scale, D, HW, K, N, gz);
    return (int)hipGetLastError();
}
