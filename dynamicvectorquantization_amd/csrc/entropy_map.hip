// entropy_map.hip -- fused patch-entropy map for gfx950.
//
// Replaces Entropy.forward of the reference (models/stage1_dynamic/dqvae_dual_entropy.py:13-63):
// grayscale -> 16x16 unfold -> 32-bin Gaussian-KDE histogram over [0, 1] (sigma 0.01) -> normalise
// (+1e-40) -> -sum p ln p.  The reference materialises a [B*256, 256, 32] fp32 tensor (2.1 GB at
// B = 256) and makes several passes over it; here one wave owns one patch and a lane owns four of its pixels.
// The 8192 Gaussian evaluations per patch collapse to four exp per pixel: with b0 the bin nearest to the pixel value
// v and d = v - c_b0 (|d| <= 1/62), the kernel value at bin b0 + k is
//     exp(-(d - k/31)^2 / 2 sigma^2) = exp(-d^2 / 2 sigma^2) * exp(d / (31 sigma^2))^k * exp(-k^2 / (2 (31 sigma)^2)),
// the last factor a constant G_k (5.5e-3, 9.2e-10, 4.6e-21, 7.0e-37 for |k| = 1..4; beyond |k| = 4 the value is below
// the smallest fp32 subnormal, i.e. exactly the 0 the reference computes).  The three nearest bins (which carry the
// entropy) are evaluated directly with the reference's own fp32 bin centres, the six beyond them by the recurrence
// v_{k+1} = v_k * exp(d / (31 sigma^2)) * G_{k+1} / G_k.  Every lane adds its pixels' <= 9 values per
// pixel into its OWN column of a per-wave LDS histogram (no atomics, fixed order -> deterministic), then lane (bin,
// half) sums that bin's row.  16 expf + ~100 LDS instructions per lane instead of 128 expf.
// The image is read from HBM exactly once (768 KiB / image).
// fp32 subnormals stay enabled (hipcc default): the reference's epsilon 1e-40 is a subnormal.
// Transcendental math -> parity is to 1e-5, not bit-exact (tests/test_entropy.py).
#include "dvq_common.h"

__global__ __launch_bounds__(256) void entropy_map_kernel(const float *__restrict__ img, int B, int H, int W,
                                                          float *__restrict__ out)
{
    constexpr int ROW = 65;                                      // histogram [32 bins][64 lanes + 1 pad]
    __shared__ float hist[4][32 * ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gh = H / 16, gw = W / 16;
    const long patch = (long)blockIdx.x * 4 + wave;
    const long npatch = (long)B * gh * gw;
    if (patch >= npatch) return;                                 // no workgroup barrier below: LDS use is per wave
    float *hw = hist[wave];
#pragma unroll
    for (int b2 = 0; b2 < 32; ++b2) hw[b2 * ROW + lane] = 0.0f;
    const int b = (int)(patch / (gh * gw));
    const int pr = (int)(patch - (long)b * gh * gw);
    const int py = pr / gw, px = pr - py * gw;
    // lane -> row lane>>2, columns 4(lane&3) .. +3 of the patch (one 16-B load per channel)
    const size_t off = ((size_t)(py * 16 + (lane >> 2))) * W + px * 16 + (lane & 3) * 4;
    const size_t plane = (size_t)H * W;
    const float *p = img + (size_t)b * 3 * plane + off;
    const f32x4 r = *(const f32x4 *)p, g = *(const f32x4 *)(p + plane), bl = *(const f32x4 *)(p + 2 * plane);
    bool has_nan = false;
    // G_k = exp(-k^2 / (2 (31 * 0.01)^2))
    const float G1 = 5.5005146e-03f, G2 = 9.1540500e-10f, G3 = 4.6092459e-21f, G4 = 7.0218754e-37f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // 0.2989 R + 0.5870 G + 0.1140 B, left to right, fp32 (:51)
        const float v = __fadd_rn(__fadd_rn(__fmul_rn(0.2989f, r[j]), __fmul_rn(0.5870f, g[j])), __fmul_rn(0.1140f, bl[j]));
        if (v != v) has_nan = true;
        const float x = v * 31.0f;
        if (x > -5.0f && x < 36.0f) {                            // (also false for NaN / Inf: an infinite pixel adds exp(-inf) = 0)
            const int b0 = (int)rintf(x);
            // the three nearest bins directly, with the reference's own bin centres (torch.linspace(0, 1, 32) in fp32:
            // i * step below the middle, 1 - (31 - i) * step above it); the far bins by the recurrence
            auto centre = [](int bb) -> float {
                const float step = 1.0f / 31.0f;
                return (bb < 16) ? __fmul_rn((float)bb, step) : __fsub_rn(1.0f, __fmul_rn((float)(31 - bb), step));
            };
            auto direct = [&](int bb) -> float {
                const float t = (v - centre(bb)) / 0.01f;        // (:38-39) residual / sigma
                return expf(-0.5f * (t * t));
            };
            const float e0 = direct(b0), ep = direct(b0 + 1), em = direct(b0 - 1);
            const float d = v - centre(b0);
            const float r1 = expf(d * (1.0f / (31.0f * 0.01f * 0.01f)));
            const float ri = 1.0f / r1;
            auto add = [&](int bb, float val) {
                if (bb >= 0 && bb < 32) hw[bb * ROW + lane] += val;
            };
            add(b0, e0); add(b0 + 1, ep); add(b0 - 1, em);
            float up = ep * r1 * (G2 / G1), dn = em * ri * (G2 / G1);
            add(b0 + 2, up); add(b0 - 2, dn);
            up *= r1 * (G3 / G2); dn *= ri * (G3 / G2);
            add(b0 + 3, up); add(b0 - 3, dn);
            up *= r1 * (G4 / G3); dn *= ri * (G4 / G3);
            add(b0 + 4, up); add(b0 - 4, dn);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int bin = lane & 31, half = lane >> 5;
    float acc = 0.0f;
    const float *hr = hw + bin * ROW + half * 32;                // bank (bin + i) mod 32: conflict-free across the 32 bins
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += hr[i];
    acc += __shfl_xor(acc, 32);
    if (__ballot(has_nan) != 0ull) acc = __builtin_nanf("");    // a NaN pixel makes every bin NaN in the reference
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
