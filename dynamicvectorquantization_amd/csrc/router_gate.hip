// router_gate.hip -- fused feature-router gate for gfx950.
//
// Replaces the forward of the reference feature routers (inference; training keeps the autograd path):
//   modules/dynamic_modules/RouterDual.py:35-43     GroupNorm x2, AvgPool2d(2) of the fine branch,
//                                                   channel concat, NHWC, Linear [-> SiLU -> Linear]
//   modules/dynamic_modules/RouterTriple.py:46-56   GroupNorm x3, AvgPool2d(4) fine / (2) median, concat,
//                                                   Linear [-> SiLU | ReLU -> Linear]
// which is 8-10 launches and ~5 passes over the branch features in the reference.  Here:
//   1. gn_stats_kernel   one pass over every branch: (mean, rstd) per (image, group).
//   2. w1_split_kernel   hidden-layer weight -> 32-row MFMA tile images, split w = hi + lo with
//                        hi = fp16(w), lo = fp16(w - hi) (22 significand bits between them).
//   3. router_gate_kernel  one workgroup per 32 coarse cells: pools the raw features (the average of
//                        normalised pixels is the normalised average), applies the GroupNorm affine,
//                        keeps the [32 x F] feature tile in LDS (also split hi + lo), multiplies it with
//                        the hidden layer on the fp16 matrix cores as hi*hi + hi*lo + lo*hi with fp32
//                        accumulation (3 MFMAs at 16x the fp32-MFMA rate; the dropped lo*lo term is
//                        2^-22 relative, i.e. fp32-grade products -- the logits decide an argmax
//                        downstream), applies the activation and contracts with the output layer in the
//                        MFMA epilogue.  The [cells x F] concat, its NHWC copy and the hidden activations
//                        never reach HBM.
// A different summation order than ATen/MKL and 2^-22 instead of 2^-24 products: tolerance parity
// (logits within 1e-4 of the reference, ~1e-6 in practice), by design.
#include "dvq_common.h"

struct DvqGateArgs {
    const float *h[3];        // branches, coarse -> fine
    const float *gn_w[3];     // GroupNorm affine per branch (nullptr with groups == 0)
    const float *gn_b[3];
    int scale[3];             // fine pixels per coarse cell edge of the branch (1, 2[, 4])
    int nb;                   // branches (2 dual, 3 triple)
    int B, C, hc, wc;
    int groups;               // 0 = no normalisation
    float eps;
};

// ---- 1. GroupNorm statistics: stats[(br * B + b) * groups + g] = (mean, rstd)
__global__ __launch_bounds__(256) void gn_stats_kernel(DvqGateArgs a, float2 *__restrict__ stats)
{
    const int br = blockIdx.y;
    const int bg = blockIdx.x;                              // b * groups + g
    const int sc = a.scale[br];
    const size_t n = (size_t)(a.C / a.groups) * (a.hc * sc) * (a.wc * sc);   // contiguous in NCHW
    const float *p = a.h[br] + (size_t)bg * n;
    double s = 0.0, ss = 0.0;
    const size_t n4 = n / 4;
    for (size_t i = threadIdx.x; i < n4; i += 256) {
        f32x4 v = *(const f32x4 *)(p + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s += v[j]; ss += (double)v[j] * v[j]; }
    }
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) { s += p[i]; ss += (double)p[i] * p[i]; }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean = red[0][0] / (double)n;
        double var = red[1][0] / (double)n - mean * mean;    // biased, as GroupNorm
        if (var < 0.0) var = 0.0;
        stats[(size_t)br * gridDim.x + bg] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)a.eps)));
    }
}

// ---- 2. hidden-layer weight W1 [Hid, F] -> split fp16 tile images (Fp = F rounded up to 16, S = Fp/16):
//   imgH / imgL [t][s][lane = 32h + c][j < 8] = hi / lo of W1[32t + c][16s + 8h + j]   (zero padded)
__global__ __launch_bounds__(256) void w1_split_kernel(const float *__restrict__ W1, int Hid, int F, int Fp,
                                                       _Float16 *__restrict__ imgH, _Float16 *__restrict__ imgL)
{
    const size_t per_tile = (size_t)32 * Fp;
    const size_t total = (size_t)((Hid + 31) / 32) * per_tile;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / per_tile);
        const int r = (int)(i - (size_t)t * per_tile);
        const int s16 = r >> 9, lane = (r >> 3) & 63, j = r & 7;
        const int k = 16 * s16 + 8 * (lane >> 5) + j;
        const int row = t * 32 + (lane & 31);
        const float w = (row < Hid && k < F) ? W1[(size_t)row * F + k] : 0.0f;
        const _Float16 hi = (_Float16)w;
        imgH[i] = hi;
        imgL[i] = (_Float16)(w - (float)hi);
    }
}

// ---- 3. the gate
// ACT: 0 = single Linear (no hidden layer), 1 = SiLU, 2 = ReLU.  G = logits per cell (2 / 3).
template <int G>
__global__ __launch_bounds__(256) void router_gate_kernel(
    DvqGateArgs a, const float2 *__restrict__ stats, const _Float16 *__restrict__ imgH,
    const _Float16 *__restrict__ imgL, const float *__restrict__ b1, const float *__restrict__ W2,
    const float *__restrict__ b2, int Hid, int act, float *__restrict__ gate)
{
    // LDS: XH | XL halves [Fp/16][64 lanes = 32h + cell][8] each (B operands), then bias / output rows
    extern __shared__ __attribute__((aligned(16))) float X[];
    const int F = a.nb * a.C;
    const int Fp = (F + 15) & ~15;
    _Float16 *XH = (_Float16 *)X, *XL = XH + 32 * Fp;
    __shared__ unsigned s_amax;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ncell = (long)a.B * a.hc * a.wc;
    const long cell0 = (long)blockIdx.x * 32;

    // ---- feature tile: pooled + normalised, split into fp16 hi + lo, in MFMA B-operand order.  A thread
    // keeps one cell (tid & 31) and walks channels (tid >> 5) + 8i, eight independent loads in flight.
    auto build = [&](float xscale) -> float {
        float vmax = 0.0f;
        const int cell = tid & 31;
        const long cg = cell0 + cell;
        const bool live = cg < ncell;
        const long cgl = live ? cg : ncell - 1;
        const int b = (int)(cgl / (a.hc * a.wc));
        const int rem = (int)(cgl - (long)b * a.hc * a.wc);
        const int y = rem / a.wc, x = rem - y * a.wc;
        const int cpg = a.groups > 0 ? a.C / a.groups : 1;
        for (int k = F + (tid >> 5); k < Fp; k += 8) {      // zero the k padding
            const int idx = (((k >> 4) * 64 + ((k >> 3) & 1) * 32 + cell) * 8) + (k & 7);
            XH[idx] = (_Float16)0.0f;
            XL[idx] = (_Float16)0.0f;
        }
        for (int br = 0; br < a.nb; ++br) {
            const int sc = a.scale[br];
            const int Wb = a.wc * sc;
            const size_t plane = (size_t)(a.hc * sc) * Wb;
            const float *p0 = a.h[br] + (size_t)b * a.C * plane + (size_t)sc * y * Wb + sc * x;
            const float2 *st = stats + ((size_t)br * a.B + b) * (a.groups > 0 ? a.groups : 1);
            const float *gw = a.gn_w[br], *gb = a.gn_b[br];
            auto pooled = [&](int ch) -> float {
                const float *p = p0 + (size_t)ch * plane;
                if (sc == 1) return p[0];
                if (sc == 2) {
                    f32x2 r0 = *(const f32x2 *)p, r1 = *(const f32x2 *)(p + Wb);
                    return ((r0[0] + r0[1]) + (r1[0] + r1[1])) * 0.25f;
                }
                float s4 = 0.0f;                        // sc == 4
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 r = *(const f32x4 *)(p + (size_t)i * Wb);
                    s4 += (r[0] + r[1]) + (r[2] + r[3]);
                }
                return s4 * 0.0625f;
            };
            auto put = [&](int ch, float v) {
                if (a.groups > 0) {
                    const float2 ms = st[ch / cpg];
                    v = (v - ms.x) * ms.y * gw[ch] + gb[ch];
                }
                v = live ? v * xscale : 0.0f;
                vmax = fmaxf(vmax, fabsf(v));
                const int k = br * a.C + ch;
                const int idx = (((k >> 4) * 64 + ((k >> 3) & 1) * 32 + cell) * 8) + (k & 7);
                const _Float16 hi = (_Float16)v;
                XH[idx] = hi;
                XL[idx] = (_Float16)(v - (float)hi);
            };
            int ch = tid >> 5;
            for (; ch + 56 < a.C; ch += 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pooled(ch + 8 * u);
#pragma unroll
                for (int u = 0; u < 8; ++u) put(ch + 8 * u, v[u]);
            }
            for (; ch < a.C; ch += 8) put(ch, pooled(ch));
        }
        return vmax;
    };
    // features beyond the fp16 range (possible with normalization_type "none") are handled by an exact
    // power-of-two rescale of the whole tile, undone on the fp32 accumulators
    if (tid == 0) s_amax = 0u;
    __syncthreads();
    float inv_scale = 1.0f;
    {
        float vmax = build(1.0f);
        atomicMax(&s_amax, __float_as_uint(vmax));          // non-negative floats order like their bits
        __syncthreads();
        const float wgmax = __uint_as_float(s_amax);
        if (!(wgmax < 16384.0f) && wgmax < __builtin_inff()) {           // workgroup-uniform, rare
            int e;
            (void)frexpf(wgmax, &e);                                      // wgmax = m * 2^e, m in [0.5, 1)
            const float xs = ldexpf(1.0f, 10 - e);                        // brings the maximum to ~2^10
            inv_scale = ldexpf(1.0f, e - 10);
            __syncthreads();
            (void)build(xs);
        }
    }
    __syncthreads();

    float part[G];
#pragma unroll
    for (int g = 0; g < G; ++g) part[g] = 0.0f;
    float *PB = X + 32 * Fp;                           // hidden bias + output-layer rows: [1 + G][Hid]
    if (act != 0) {
        for (int i = tid; i < Hid; i += 256) {
            PB[i] = b1[i];
#pragma unroll
            for (int g = 0; g < G; ++g) PB[(1 + g) * Hid + i] = W2[(size_t)g * Hid + i];
        }
    }

    if (act == 0) {
        // single Linear: gate[cell][g] = W2[g][:] . x + b2[g]; 8 threads per cell split k
        const int cell = tid >> 3, sub = tid & 7;
        for (int k = sub; k < F; k += 8) {
            const int idx = (((k >> 4) * 64 + ((k >> 3) & 1) * 32 + cell) * 8) + (k & 7);
            const float xv = ((float)XH[idx] + (float)XL[idx]) * inv_scale;
#pragma unroll
            for (int g = 0; g < G; ++g) part[g] = __builtin_fmaf(W2[(size_t)g * F + k], xv, part[g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            part[g] += __shfl_xor(part[g], 1);
            part[g] += __shfl_xor(part[g], 2);
            part[g] += __shfl_xor(part[g], 4);
        }
        const long cg = cell0 + cell;
        if (sub == 0 && cg < ncell) {
#pragma unroll
            for (int g = 0; g < G; ++g) gate[cg * G + g] = part[g] + b2[g];
        }
        return;
    }

    __syncthreads();
    // ---- hidden layer on the fp16 matrix cores (split operands): wave w takes hidden-row tiles w, w + 4, ...
    const int T = (Hid + 31) / 32;
    const int S = Fp / 16;
    const f16x8 *xh = (const f16x8 *)XH + lane, *xl = (const f16x8 *)XL + lane;      // + s * 64 per k-step
    // The A fragments (hidden-layer weight tiles) come straight from L2, one 16-B load per lane per MFMA triple; a
    // workgroup has one wave per SIMD (the feature tile fills the LDS), so the L2 latency is hidden by software
    // pipelining: the fragments of the next two groups of four k-steps (2 x 8 loads = 64 VGPRs) are in flight while
    // the current group's twelve MFMAs run.  Groups are numbered through this wave's tiles: g -> (tile wave + 4 (g / GPT),
    // k-steps 4 (g % GPT) ..).
    auto epilogue = [&](int t, const f32x16 &acc) {    // bias, activation, contraction with the output layer
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j < Hid) {
                const float yv = acc[r] * inv_scale + PB[j];
                const float hv = (act == 1) ? yv / (1.0f + expf(-yv)) : (yv > 0.0f ? yv : 0.0f);
#pragma unroll
                for (int g = 0; g < G; ++g) part[g] = __builtin_fmaf(PB[(1 + g) * Hid + j], hv, part[g]);
            }
        }
    };
    if ((S & 3) == 0) {
        const int GPT = S >> 2;                        // groups per tile
        const int ntile = (T - wave + 3) / 4;          // tiles of this wave
        const int NG = ntile * GPT;
        auto fetch = [&](int g, f16x8 (&vh)[4], f16x8 (&vl)[4]) {
            const int gg = g < NG ? g : NG - 1;        // past the end: harmless repeat
            const int t = wave + 4 * (gg / GPT), s0 = 4 * (gg % GPT);
            const f16x8 *ah = (const f16x8 *)imgH + (size_t)t * S * 64 + lane;
            const f16x8 *al = (const f16x8 *)imgL + (size_t)t * S * 64 + lane;
#pragma unroll
            for (int u = 0; u < 4; ++u) { vh[u] = ah[(s0 + u) * 64]; vl[u] = al[(s0 + u) * 64]; }
        };
        f32x16 acc;
        auto step = [&](int g, const f16x8 (&vh)[4], const f16x8 (&vl)[4]) {
            if (g >= NG) return;
            const int t = wave + 4 * (g / GPT), s0 = 4 * (g % GPT);
            if (s0 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            }
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { bh[u] = xh[(s0 + u) * 64]; bl[u] = xl[(s0 + u) * 64]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[u], bh[u], acc, 0, 0, 0);   // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u], bl[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u], bh[u], acc, 0, 0, 0);
            }
            if (s0 + 4 == S) epilogue(t, acc);
        };
        if (NG > 0) {
            f16x8 h0[4], l0[4], h1[4], l1[4], h2[4], l2[4];
            fetch(0, h0, l0);
            fetch(1, h1, l1);
            fetch(2, h2, l2);
            for (int g = 0; g < NG; g += 3) {
                step(g, h0, l0);
                fetch(g + 3, h0, l0);
                step(g + 1, h1, l1);
                fetch(g + 4, h1, l1);
                step(g + 2, h2, l2);
                fetch(g + 5, h2, l2);
            }
        }
    } else {
    for (int t = wave; t < T; t += 4) {
        const f16x8 *ah = (const f16x8 *)imgH + (size_t)t * S * 64 + lane;
        const f16x8 *al = (const f16x8 *)imgL + (size_t)t * S * 64 + lane;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int s0 = 0; s0 < S; ++s0) {
            const f16x8 vh = ah[s0 * 64], vl = al[s0 * 64], bh = xh[s0 * 64], bl = xl[s0 * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, bh, acc, 0, 0, 0);
        }
        epilogue(t, acc);
    }
    }
    __syncthreads();                                   // everyone is done reading X
    float *red = X;                                    // [4 waves][G][32 cells]
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float v = part[g] + __shfl_xor(part[g], 32);
        if (h == 0) red[(wave * G + g) * 32 + c] = v;
    }
    __syncthreads();
    if (tid < 32 * G) {
        const int cell = tid / G, g = tid - cell * G;
        const long cg = cell0 + cell;
        if (cg < ncell) {
            float v = (red[(0 * G + g) * 32 + cell] + red[(1 * G + g) * 32 + cell]) +
                      (red[(2 * G + g) * 32 + cell] + red[(3 * G + g) * 32 + cell]);
            gate[cg * G + g] = v + b2[g];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static size_t align256r(size_t x) { return (x + 255) / 256 * 256; }

// ws: [stats nb*B*groups float2][W1 hi image | W1 lo image: ceil(Hid/32)*32*Fp halves each]
size_t dvq_router_gate_ws_bytes(int nb, int B, int C, int groups, int Hid)
{
    const size_t Fp = ((size_t)nb * C + 15) / 16 * 16;
    return align256r((size_t)nb * B * (groups > 0 ? groups : 1) * sizeof(float2)) +
           align256r((size_t)((Hid + 31) / 32) * 32 * Fp * 2 * sizeof(_Float16)) + 256;
}

int dvq_launch_router_gate(int nb, const float *const *h, const float *const *gn_w, const float *const *gn_b,
                           int B, int C, int hc, int wc, int groups, float eps,
                           const float *W1, const float *b1, const float *W2, const float *b2,
                           int Hid, int act, float *gate, void *ws, hipStream_t st)
{
    DvqGateArgs a;
    for (int i = 0; i < 3; ++i) {
        a.h[i] = i < nb ? h[i] : nullptr;
        a.gn_w[i] = (i < nb && groups > 0) ? gn_w[i] : nullptr;
        a.gn_b[i] = (i < nb && groups > 0) ? gn_b[i] : nullptr;
        a.scale[i] = 1 << i;
    }
    a.nb = nb; a.B = B; a.C = C; a.hc = hc; a.wc = wc; a.groups = groups; a.eps = eps;
    const int F = nb * C, Fp = (F + 15) & ~15;
    float2 *stats = (float2 *)ws;
    _Float16 *imgH = (_Float16 *)((char *)ws + align256r((size_t)nb * B * (groups > 0 ? groups : 1) * sizeof(float2)));
    _Float16 *imgL = imgH + (size_t)((Hid + 31) / 32) * 32 * Fp;
    if (groups > 0)
        hipLaunchKernelGGL(gn_stats_kernel, dim3(B * groups, nb), dim3(256), 0, st, a, stats);
    if (act != 0) {
        size_t total = (size_t)((Hid + 31) / 32) * 32 * Fp;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(w1_split_kernel, dim3(blocks), dim3(256), 0, st, W1, Hid, F, Fp, imgH, imgL);
    }
    const long ncell = (long)B * hc * wc;
    const unsigned grid = (unsigned)((ncell + 31) / 32);
    const size_t shmem = ((size_t)32 * Fp + (size_t)(1 + nb) * Hid) * sizeof(float);
    if (nb == 2) {
        static unsigned long long done2 = 0;
        { int rc = dvq_allow_dynamic_lds((const void *)router_gate_kernel<2>, 160 * 1024 - 256, &done2); if (rc) return rc; }
        hipLaunchKernelGGL(router_gate_kernel<2>, dim3(grid), dim3(256), shmem, st, a, stats, imgH, imgL, b1, W2, b2, Hid, act, gate);
    } else {
        static unsigned long long done3 = 0;
        { int rc = dvq_allow_dynamic_lds((const void *)router_gate_kernel<3>, 160 * 1024 - 256, &done3); if (rc) return rc; }
        hipLaunchKernelGGL(router_gate_kernel<3>, dim3(grid), dim3(256), shmem, st, a, stats, imgH, imgL, b1, W2, b2, Hid, act, gate);
    }
    return (int)hipGetLastError();
}
