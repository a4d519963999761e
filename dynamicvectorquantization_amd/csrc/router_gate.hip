// router_gate.hip -- fused feature-router gate for gfx950.
//
// Replaces the forward of the reference feature routers (inference; training keeps the autograd path):
//   modules/dynamic_modules/RouterDual.py:35-43     GroupNorm x2, AvgPool2d(2) of the fine branch,
//                                                   channel concat, NHWC, Linear [-> SiLU -> Linear]
//   modules/dynamic_modules/RouterTriple.py:46-56   GroupNorm x3, AvgPool2d(4) fine / (2) median, concat,
//                                                   Linear [-> SiLU | ReLU -> Linear]
// which is 8-10 launches and ~5 passes over the branch features in the reference.  Here:
//   1. gate_pool_kernel  THE one pass over every branch (one workgroup per image and channel group, all
//                        branches): (mean, rstd) per (image, group) and the raw per-cell averages
//                        pooled[b][k = branch * C + c][cell] (the average of normalised pixels is the normalised
//                        average, so the GroupNorm affine is applied to the averages later).  Streaming, high
//                        occupancy: bound by the feature read.
//   2. w1_split_kernel   hidden-layer weight -> 32-row MFMA tile images, split w = hi + lo with
//                        hi = fp16(w), lo = fp16(w - hi) (22 significand bits between them); done once per
//                        weight version when the caller keeps the images (dvq_router_gate_prepare_f32).
//   3. router_gate_kernel  one workgroup per 32 coarse cells: reads their pooled averages (coalesced over cells),
//                        applies the GroupNorm affine,
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
#include <cstdlib>
#include <type_traits>

struct DvqGateArgs {
    const float *h[3];        // branches, coarse -> fine
    const float *gn_w[3];     // GroupNorm affine per branch (nullptr with groups == 0)
    const float *gn_b[3];
    int scale[3];             // fine pixels per coarse cell edge of the branch (1, 2[, 4])
    int nb;                   // branches (2 dual, 3 triple)
    int B, C, hc, wc;
    int groups;               // 0 = no normalisation
    float eps;
    int vec;                  // pooling pass: 16-B loads where the rows allow (0: tuning aid DVQ_GATE_POOL_SCALAR=1)
};

// ---- 1. one pass over the branch features: the GroupNorm of every (image, group) folded with its affine into
// ab[b * F + k] = (rstd * w, bias - mean * rstd * w), k = br * C + ch (normalised = raw * ab.x + ab.y), and
// the raw per-cell averages pool[(b * F + br * C + ch) * ncell + cell].  Workgroup = (image b, channel group g); a
// thread owns whole (channel, cell) pairs, so a pooled value is summed in a fixed order by one thread (deterministic,
// and the same expression as before the split) and neighbouring threads read neighbouring cells: every load
// instruction covers whole rows of cells.  groups == 0 (no normalisation): pseudo-groups of 8 channels, no stats.
// IMG (the GEMM form of the gate, gate_gemm_kernel): the pooled averages of the workgroup's channels stay in LDS until the group's
// statistics are known, and are written ONCE, normalised and split hi + lo, straight into the MFMA B-operand images
//   ximg[block = cell / 32][k-step s][hi 1 KiB | lo 1 KiB][lane = 32 h + cell % 32][j < 8] = feature k = 16 s + 8 h + j of the cell
// (cell = global cell index b * hc * wc + y * wc + x), so that the matrix kernel streams them with plain LDS-DMA and no
// workgroup rebuilds a tile.  fp16 range: every workgroup derives the SAME power-of-two scale from the GroupNorm parameters
// alone -- |normalised value| <= |w| sqrt(n) + |b| for every element of a group of n values, and an average of such values
// obeys the same bound -- so no workgroup needs another's data; xs[0] = that scale's inverse for the matrix kernel.
#define GATE_LD(p) (NT ? __builtin_nontemporal_load(p) : *(p))    // NT: see DVQ_CACHED_MAX_BYTES (dvq_common.h)
#ifndef DVQ_POOL_WPE
#define DVQ_POOL_WPE 8           // workgroups per CU the pooling pass is compiled for (a streaming pass lives on occupancy)
#endif
#define DVQ_GATE_NORM_MAGIC 0x44563531      // tail of the weight prep: [magic][max |gn_w|, max |gn_b| per branch] (dvq_router_gate_prepare_norm_f32)
template <bool IMG, bool NT>
__global__ __launch_bounds__(256, DVQ_POOL_WPE) void gate_pool_kernel(DvqGateArgs a, float2 *__restrict__ ab,
                                                        float *__restrict__ pool, char *__restrict__ ximg,
                                                        float *__restrict__ xs, const float *__restrict__ nbound)
{
    const int G = a.groups > 0 ? a.groups : a.C / 8;
    const int cpg = a.C / G;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int ncell = a.hc * a.wc, F = a.nb * a.C;
    const int npair = cpg * ncell;
    __shared__ double red[3][2][4];                          // [branch (IMG; else 0)][sum | sum of squares][wave]
    __shared__ float s_bound[4];
    __shared__ float2 s_aff[3 * 64];                         // IMG: (scale, shift) of the group's channels, per branch
    extern __shared__ float lp[];                            // IMG: [branch][cpg][ncell] pooled averages
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float xscale = 1.0f;
    if (IMG) {
        // the fp16 range bound max_k (|w_k| sqrt(n_br) + |b_k|).  With the per-branch maxima of the GroupNorm parameters prepared
        // once per parameter version (nbound; a slightly larger bound: max |w| and max |b| need not sit on one channel) it is six
        // scalars; without them every workgroup scans all F parameters: a dependent load chain and a barrier of its own, 5 of
        // 46 us at B = 128 and 39 of 314 us at B = 1024 (profiles/r05_gate_pool.json)
        float bound = 0.0f;
        if (nbound != nullptr && __builtin_bit_cast(int, nbound[0]) == DVQ_GATE_NORM_MAGIC) {
            for (int br = 0; br < a.nb; ++br) {
                const float n = (float)cpg * (float)(a.hc * a.scale[br]) * (float)(a.wc * a.scale[br]);
                bound = fmaxf(bound, nbound[1 + 2 * br] * sqrtf(n) * 1.0001f + nbound[2 + 2 * br]);
            }
        } else {
        for (int k = tid; k < F; k += 256) {
            const int br = k / a.C, ch = k - br * a.C;
            const float n = (float)cpg * (float)(a.hc * a.scale[br]) * (float)(a.wc * a.scale[br]);
            bound = fmaxf(bound, fabsf(a.gn_w[br][ch]) * sqrtf(n) * 1.0001f + fabsf(a.gn_b[br][ch]));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off));
        if (lane == 0) s_bound[wave] = bound;
        __syncthreads();
        bound = fmaxf(fmaxf(s_bound[0], s_bound[1]), fmaxf(s_bound[2], s_bound[3]));
        }
        float inv = 1.0f;
        if (bound > 0.0f && bound < __builtin_inff()) {
            int e;
            (void)frexpf(bound, &e);                         // bound = m 2^e, m in [0.5, 1)
            const int sh = 13 - e < 100 ? 13 - e : 100;      // brings the bound into [2^12, 2^13): up as well as down -- the bound is
            xscale = ldexpf(1.0f, sh);                       // ~sqrt(n) above typical values, whose lo halves would otherwise be fp16
            inv = ldexpf(1.0f, -sh);                         // subnormals (an absolute 2^-25 instead of 22 significand bits)
        }
        if (blockIdx.x == 0 && tid == 0) xs[0] = inv;
    }
    for (int br = 0; br < a.nb; ++br) {
        const int sc = a.scale[br];
        const int Wb = a.wc * sc;
        const size_t plane = (size_t)(a.hc * sc) * Wb;
        const float *p0 = a.h[br] + ((size_t)b * a.C + (size_t)g * cpg) * plane;
        float *o0 = IMG ? lp + (size_t)br * npair : pool + ((size_t)b * F + (size_t)br * a.C + (size_t)g * cpg) * ncell;
        double s = 0.0, ss = 0.0;
        if (a.vec && (Wb & 3) == 0) {
            // rows are whole float4s: a thread takes 4 consecutive source columns = 4 / sc cells of one row of cells
            // (16-B loads for every branch), sc source rows deep
            const int cu = 4 / sc;                           // cells per unit (sc = 1, 2, 4 -> 4, 2, 1)
            const int upr = a.wc / cu;                       // units per row of cells
            const int upc = a.hc * upr;                      // units per channel
            const int nunit = cpg * upc;
            for (int u = tid; u < nunit; u += 256) {
                const int ch = u / upc, r = u - ch * upc;
                const int y = r / upr, xu = r - y * upr;
                const float *p = p0 + (size_t)ch * plane + (size_t)sc * y * Wb + 4 * xu;
                float *o = o0 + (size_t)ch * ncell + y * a.wc + xu * cu;
                auto acc4 = [&](const f32x4 &r) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s += r[j]; ss += (double)r[j] * r[j]; }
                };
                if (sc == 1) {
                    const f32x4 r0 = GATE_LD((const f32x4 *)p);
                    acc4(r0);
                    *(f32x4 *)o = r0;
                } else if (sc == 2) {
                    const f32x4 r0 = GATE_LD((const f32x4 *)p);
                    const f32x4 r1 = GATE_LD((const f32x4 *)(p + Wb));
                    acc4(r0); acc4(r1);
                    f32x2 v;
                    v[0] = ((r0[0] + r0[1]) + (r1[0] + r1[1])) * 0.25f;
                    v[1] = ((r0[2] + r0[3]) + (r1[2] + r1[3])) * 0.25f;
                    *(f32x2 *)o = v;
                } else {
                    f32x4 rw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) rw[i] = GATE_LD((const f32x4 *)(p + (size_t)i * Wb));
                    float s4 = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        s4 += (rw[i][0] + rw[i][1]) + (rw[i][2] + rw[i][3]);
                        acc4(rw[i]);
                    }
                    *o = s4 * 0.0625f;
                }
            }
        } else
        for (int pr = tid; pr < npair; pr += 256) {
            const int ch = pr / ncell, cell = pr - ch * ncell;
            const int y = cell / a.wc, x = cell - y * a.wc;
            const float *p = p0 + (size_t)ch * plane + (size_t)sc * y * Wb + sc * x;
            float v;
            if (sc == 1) {
                const float r = GATE_LD(p);
                s += r; ss += (double)r * r;
                v = r;
            } else if (sc == 2) {
                const f32x2 r0 = GATE_LD((const f32x2 *)p);
                const f32x2 r1 = GATE_LD((const f32x2 *)(p + Wb));
#pragma unroll
                for (int j = 0; j < 2; ++j) { s += r0[j]; ss += (double)r0[j] * r0[j]; s += r1[j]; ss += (double)r1[j] * r1[j]; }
                v = ((r0[0] + r0[1]) + (r1[0] + r1[1])) * 0.25f;
            } else {
                f32x4 r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = GATE_LD((const f32x4 *)(p + (size_t)i * Wb));
                float s4 = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s4 += (r[i][0] + r[i][1]) + (r[i][2] + r[i][3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s += r[i][j]; ss += (double)r[i][j] * r[i][j]; }
                }
                v = s4 * 0.0625f;
            }
            o0[pr] = v;
        }
        auto affine = [&](int brr, int chl) -> float2 {      // the group's statistics of branch brr folded with channel chl's affine
            const double n = (double)cpg * (double)(a.hc * a.scale[brr]) * (double)(a.wc * a.scale[brr]);
            const int rb = IMG ? brr : 0;
            const double mean = ((red[rb][0][0] + red[rb][0][1]) + (red[rb][0][2] + red[rb][0][3])) / n;
            double var = ((red[rb][1][0] + red[rb][1][1]) + (red[rb][1][2] + red[rb][1][3])) / n - mean * mean;   // biased, as GroupNorm
            if (var < 0.0) var = 0.0;
            const float mf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
            const int ch = g * cpg + chl;
            const float sc_ = rstd * a.gn_w[brr][ch];
            return make_float2(sc_, a.gn_b[brr][ch] - mf * sc_);
        };
        if (a.groups > 0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); ss += __shfl_xor(ss, off); }
            if (IMG) {                                       // one barrier for all branches, after the loop
                if (lane == 0) { red[br][0][wave] = s; red[br][1][wave] = ss; }
            } else {
                __syncthreads();                             // red[] of the previous branch consumed
                if (lane == 0) { red[0][0][wave] = s; red[0][1][wave] = ss; }
                __syncthreads();
                if (tid < cpg) ab[(size_t)b * F + (size_t)br * a.C + g * cpg + tid] = affine(br, tid);
            }
        }
        if (!IMG || br + 1 < a.nb) continue;
        // IMG, after the last branch: every branch's pooled averages are in LDS and its sums in red[]
        __syncthreads();
        for (int i = tid; i < a.nb * cpg; i += 256) s_aff[(i / cpg) * 64 + (i % cpg)] = affine(i / cpg, i % cpg);
        __syncthreads();
        // one thread per (branch, cell, octet of channels): 8 LDS reads, normalise, scale, split, two 16-B stores
        const int noct = cpg / 8;
        for (int u = tid; u < a.nb * ncell * noct; u += 256) {
            const int brr = u / (ncell * noct), r2 = u - brr * (ncell * noct);
            const int o = r2 / ncell, cell = r2 - o * ncell;
            f16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2 af = s_aff[brr * 64 + 8 * o + j];
                const float x = (lp[(size_t)brr * npair + (8 * o + j) * ncell + cell] * af.x + af.y) * xscale;
                const _Float16 hh = (_Float16)x;
                hi[j] = hh;
                lo[j] = (_Float16)(x - (float)hh);
            }
            const int kk = brr * a.C + g * cpg + 8 * o;      // first feature of the octet
            const long cg = (long)b * ncell + cell;
            char *dst = ximg + ((size_t)(cg >> 5) * (F / 16) + (kk >> 4)) * 2048 + (((kk >> 3) & 1) * 32 + (int)(cg & 31)) * 16;
            *(f16x8 *)dst = hi;
            *(f16x8 *)(dst + 1024) = lo;
        }
    }
}

// ---- 2. hidden-layer weight W1 [Hid, F] -> split fp16 tile images (Fp = F rounded up to 16, S = Fp/16):
//   imgH / imgL [t][s][lane = 32h + c][j < 8] = hi / lo of W1[32t + c][16s + 8h + j]   (zero padded)
//   The images hold w * 2^sh with the power of two that brings max |W1| into [2^14, 2^15): fp16 keeps 11 bits below 2^-14 no
//   longer (subnormals), so without it lo = fp16(w - hi) of a default-initialised layer (|w| ~ 0.04, w - hi ~ 2e-5) is already a
//   subnormal and a layer of small weights (|w| ~ 1e-5, e.g. behind GroupNorm weights of a few thousand) would lose hi as well --
//   a golden of that shape is off by 2e-3 without the scale.  tail[DVQ_GATE_WMAX] = max |W1| (w1_max_kernel),
//   tail[DVQ_GATE_WINV] = 2^-sh for the matrix kernels' epilogues (exact: applied to the fp32 accumulators).
#define DVQ_GATE_WMAX 8
#define DVQ_GATE_WINV 9
__global__ __launch_bounds__(256) void w1_max_kernel(const float *__restrict__ W1, size_t n, unsigned *__restrict__ wmax)
{
    __shared__ unsigned red[4];
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = max(m, __float_as_uint(fabsf(W1[i])));           // non-negative floats order like their bits (NaN above infinity)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(wmax, max(max(red[0], red[1]), max(red[2], red[3])));
}
__global__ __launch_bounds__(256) void w1_split_kernel(const float *__restrict__ W1, int Hid, int F, int Fp,
                                                       _Float16 *__restrict__ imgH, _Float16 *__restrict__ imgL,
                                                       float *__restrict__ tail)
{
    const float wmax = tail[DVQ_GATE_WMAX];
    float wscale = 1.0f, winv = 1.0f;
    if (wmax > 0.0f && wmax < __builtin_inff()) {            // zero, infinite or NaN weights: as they are
        int e;
        (void)frexpf(wmax, &e);                              // wmax = m 2^e, m in [0.5, 1)
        const int sh = 15 - e < 100 ? 15 - e : 100;
        wscale = ldexpf(1.0f, sh);
        winv = ldexpf(1.0f, -sh);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        tail[0] = 0.0f;                                      // no GroupNorm maxima (yet): gate_pool_kernel, nbound
        tail[DVQ_GATE_WINV] = winv;
    }
    const size_t per_tile = (size_t)32 * Fp;
    const size_t total = (size_t)((Hid + 31) / 32) * per_tile;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / per_tile);
        const int r = (int)(i - (size_t)t * per_tile);
        const int s16 = r >> 9, lane = (r >> 3) & 63, j = r & 7;
        const int k = 16 * s16 + 8 * (lane >> 5) + j;
        const int row = t * 32 + (lane & 31);
        const float w = (row < Hid && k < F) ? W1[(size_t)row * F + k] * wscale : 0.0f;
        const _Float16 hi = (_Float16)w;
        imgH[i] = hi;
        imgL[i] = (_Float16)(w - (float)hi);
    }
}

// ---- 3. the gate
// ACT: 0 = single Linear (no hidden layer), 1 = SiLU, 2 = ReLU.  G = logits per cell (2 / 3).
#define GATE_NW 8            // waves per workgroup of the gate kernel: two per SIMD (the feature tile fills the LDS: one workgroup per CU)
// CB = blocks of 32 cells per workgroup: every weight fragment fetched from L2 feeds CB x 3 MFMAs (the stream of
// weight fragments through one CU's vector-memory path is what bounds the matrix phase); CB = 2 needs the split tile of
// 64 cells in LDS: F <= 512 (the dual routers).
template <int G, int CB>
__global__ __launch_bounds__(GATE_NW * 64) void router_gate_kernel(
    DvqGateArgs a, const float2 *__restrict__ ab, const float *__restrict__ pool,
    const _Float16 *__restrict__ imgH, const _Float16 *__restrict__ imgL, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, int Hid, int act, float *__restrict__ gate,
    const float *__restrict__ wtail)
{
    // LDS: per cell block XH | XL halves [Fp/16][64 lanes = 32h + cell][8] each (B operands), then bias / output rows
    extern __shared__ __attribute__((aligned(16))) float X[];
    const int F = a.nb * a.C;
    const int Fp = (F + 15) & ~15;
    __shared__ unsigned s_amax;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ncell = (long)a.B * a.hc * a.wc;
    const long cell0 = (long)blockIdx.x * 32 * CB;

    // ---- feature tile: normalised averages, split into fp16 hi + lo, in MFMA B-operand order.  A thread keeps one
    // cell (tid & 31) and walks the octets of features (tid >> 5) + 2 GATE_NW i: eight loads (one 128-B line per half wave
    // each: the cells of a workgroup are consecutive in pool[b][k][cell]), one 16-B LDS store per image.
    auto build = [&](int cb, float xscale) -> float {
        float vmax = 0.0f;
        _Float16 *XH = (_Float16 *)X + (size_t)cb * 64 * Fp, *XL = XH + 32 * Fp;
        const int cell = tid & 31;
        const long cg = cell0 + 32 * cb + cell;
        const bool live = cg < ncell;
        const long cgl = live ? cg : ncell - 1;
        const int nci = a.hc * a.wc;
        const int b = (int)(cgl / nci);
        const int rem = (int)(cgl - (long)b * nci);
        const float *pb = pool + (size_t)b * F * nci + rem;
        const float2 *abb = ab + (size_t)b * F;
        auto load8 = [&](int o, float (&v)[8], f32x4 (&q)[4]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (8 * o < F) ? pb[(size_t)(8 * o + j) * nci] : 0.0f;
            if (a.groups > 0 && 8 * o < F) {
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = *(const f32x4 *)(abb + 8 * o + 2 * j);   // (scale, shift) of two channels
            }
        };
        auto put8 = [&](int o, float (&v)[8], const f32x4 (&q)[4]) {
            const int k0 = 8 * o;
            f16x8 hi, lo;
            if (k0 < F) {
                if (a.groups > 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] * q[j >> 1][2 * (j & 1)] + q[j >> 1][2 * (j & 1) + 1];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = live ? v[j] * xscale : 0.0f;
                    vmax = fmaxf(vmax, fabsf(x));
                    const _Float16 hh = (_Float16)x;
                    hi[j] = hh;
                    lo[j] = (_Float16)(x - (float)hh);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.0f; lo[j] = (_Float16)0.0f; }
            }
            const int slot = (o >> 1) * 64 + (o & 1) * 32 + cell;          // f16x8 index: k-step o / 2, half o & 1
            ((f16x8 *)XH)[slot] = hi;
            ((f16x8 *)XL)[slot] = lo;
        };
        const int NO = Fp / 8;
        constexpr int OL = GATE_NW * 2;                      // octet lanes (threads per cell)
        int o = tid >> 5;
        for (; o + 3 * OL < NO; o += 4 * OL) {               // four octets (32 loads) in flight per thread
            float v0[8], v1[8], v2[8], v3[8];
            f32x4 q0[4], q1[4], q2[4], q3[4];
            load8(o, v0, q0); load8(o + OL, v1, q1); load8(o + 2 * OL, v2, q2); load8(o + 3 * OL, v3, q3);
            put8(o, v0, q0); put8(o + OL, v1, q1); put8(o + 2 * OL, v2, q2); put8(o + 3 * OL, v3, q3);
        }
        if (o + OL < NO) {                                   // two octets left (triple at GATE_NW = 8: 6 per thread)
            float v0[8], v1[8];
            f32x4 q0[4], q1[4];
            load8(o, v0, q0); load8(o + OL, v1, q1);
            put8(o, v0, q0); put8(o + OL, v1, q1);
            o += 2 * OL;
        }
        for (; o < NO; o += OL) {
            float v0[8];
            f32x4 q0[4];
            load8(o, v0, q0);
            put8(o, v0, q0);
        }
        return vmax;
    };
    // features beyond the fp16 range (possible with normalization_type "none") are handled by an exact
    // power-of-two rescale of the whole tile, undone on the fp32 accumulators
    if (tid == 0) s_amax = 0u;
    __syncthreads();
    float inv_scale = 1.0f;
#ifdef GATE_PROBE_NO_BUILD
    if (false)
#endif
    {
        float vmax = 0.0f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) vmax = fmaxf(vmax, build(cb, 1.0f));
        atomicMax(&s_amax, __float_as_uint(vmax));          // non-negative floats order like their bits
        __syncthreads();
        const float wgmax = __uint_as_float(s_amax);
        if (!(wgmax < 16384.0f) && wgmax < __builtin_inff()) {           // workgroup-uniform, rare
            int e;
            (void)frexpf(wgmax, &e);                                      // wgmax = m * 2^e, m in [0.5, 1)
            const float xs = ldexpf(1.0f, 10 - e);                        // brings the maximum to ~2^10
            inv_scale = ldexpf(1.0f, e - 10);
            __syncthreads();
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) (void)build(cb, xs);
        }
    }
    __syncthreads();

    float part[CB][G];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < G; ++g) part[cb][g] = 0.0f;
    float *PB = X + CB * 32 * Fp;                        // hidden bias + output-layer rows: [1 + G][Hid]
    if (act != 0) {
        for (int i = tid; i < Hid; i += GATE_NW * 64) {
            PB[i] = b1[i];
#pragma unroll
            for (int g = 0; g < G; ++g) PB[(1 + g) * Hid + i] = W2[(size_t)g * Hid + i];
        }
    }

    if (act == 0) {
        // single Linear: gate[cell][g] = W2[g][:] . x + b2[g]; 2 GATE_NW threads per cell split k
        constexpr int SUB = GATE_NW * 2;
        const int cell = tid / SUB, sub = tid % SUB;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const _Float16 *XH = (const _Float16 *)X + (size_t)cb * 64 * Fp, *XL = XH + 32 * Fp;
            for (int k = sub; k < F; k += SUB) {
                const int idx = (((k >> 4) * 64 + ((k >> 3) & 1) * 32 + cell) * 8) + (k & 7);
                const float xv = ((float)XH[idx] + (float)XL[idx]) * inv_scale;
#pragma unroll
                for (int g = 0; g < G; ++g) part[cb][g] = __builtin_fmaf(W2[(size_t)g * F + k], xv, part[cb][g]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int off = 1; off < SUB; off <<= 1) part[cb][g] += __shfl_xor(part[cb][g], off);
            }
            const long cg = cell0 + 32 * cb + cell;
            if (sub == 0 && cg < ncell) {
#pragma unroll
                for (int g = 0; g < G; ++g) gate[cg * G + g] = part[cb][g] + b2[g];
            }
        }
        return;
    }

    __syncthreads();
#ifdef GATE_PROBE_NO_MFMA
    if (gate != nullptr) return;
#endif
    // ---- hidden layer on the fp16 matrix cores (split operands): wave w takes hidden-row tiles w, w + GATE_NW, ...
    const int T = (Hid + 31) / 32;
    const int S = Fp / 16;
    const f16x8 *xh = (const f16x8 *)X + lane;         // cell block cb: + cb * 8 * Fp; lo half: + 4 * Fp; k-step s: + s * 64
    const int XLO = 4 * Fp, XCB = 8 * Fp;              // in f16x8 units (32 * Fp halves = 4 * Fp fragments)
    // The A fragments (hidden-layer weight tiles) come straight from L2, one 16-B load per lane per CB MFMA triples; the
    // feature tile fills the LDS, so there is one workgroup per CU and the L2 latency is hidden by its own two waves per
    // SIMD plus software pipelining: the fragments of the next three groups of four k-steps (3 x 8 loads = 96 VGPRs) are
    // in flight while the current group's MFMAs run.  Groups are numbered through this wave's tiles:
    // g -> (tile wave + GATE_NW (g / GPT), k-steps 4 (g % GPT) ..).
    const float hid_scale = inv_scale * wtail[DVQ_GATE_WINV];         // feature scale and weight scale (w1_split_kernel), both powers of two
    auto epilogue = [&](int t, const f32x16 &acc, float (&pt)[G]) {    // bias, activation, contraction with the output layer
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j < Hid) {
                const float yv = acc[r] * hid_scale + PB[j];
                // SiLU with the fast exp and division (v_exp_f32 / v_rcp_f32 based, ~1-2 ulp): the library expf + IEEE division were
                // a fifth of the kernel; the logits' tolerance is 1e-4 (measured error unchanged at ~1e-6)
                const float hv = (act == 1) ? __fdividef(yv, 1.0f + __expf(-yv)) : (yv > 0.0f ? yv : 0.0f);
#pragma unroll
                for (int g = 0; g < G; ++g) pt[g] = __builtin_fmaf(PB[(1 + g) * Hid + j], hv, pt[g]);
            }
        }
    };
    if ((S & 3) == 0) {
        const int GPT = S >> 2;                        // groups per tile
        const int ntile = (T - wave + GATE_NW - 1) / GATE_NW;   // tiles of this wave
        const int NG = ntile * GPT;
        auto fetch = [&](int g, f16x8 (&vh)[4], f16x8 (&vl)[4]) {
            const int gg = g < NG ? g : NG - 1;        // past the end: harmless repeat
            const int t = wave + GATE_NW * (gg / GPT), s0 = 4 * (gg % GPT);
            const f16x8 *ah = (const f16x8 *)imgH + (size_t)t * S * 64 + lane;
            const f16x8 *al = (const f16x8 *)imgL + (size_t)t * S * 64 + lane;
#ifdef GATE_PROBE_NO_LOADS
#pragma unroll
            for (int u = 0; u < 4; ++u) { vh[u] = xh[u * 64]; vl[u] = xh[XLO + u * 64]; (void)ah; (void)al; }
#else
#pragma unroll
            for (int u = 0; u < 4; ++u) { vh[u] = ah[(s0 + u) * 64]; vl[u] = al[(s0 + u) * 64]; }
#endif
        };
        f32x16 acc[CB];
        auto step = [&](int g, const f16x8 (&vh)[4], const f16x8 (&vl)[4]) {
            if (g >= NG) return;
            const int t = wave + GATE_NW * (g / GPT), s0 = 4 * (g % GPT);
            if (s0 == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const f16x8 bh = xh[cb * XCB + (s0 + u) * 64], bl = xh[cb * XCB + XLO + (s0 + u) * 64];
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[u], bh, acc[cb], 0, 0, 0);   // small terms first
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u], bl, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u], bh, acc[cb], 0, 0, 0);
                }
            }
            if (s0 + 4 == S) {
#ifdef GATE_PROBE_NO_EPI
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) part[cb][0] += acc[cb][0];
#else
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) epilogue(t, acc[cb], part[cb]);
#endif
            }
        };
        if (NG > 0) {
            f16x8 h0[4], l0[4], h1[4], l1[4], h2[4], l2[4], h3[4], l3[4];
            fetch(0, h0, l0);
            fetch(1, h1, l1);
            fetch(2, h2, l2);
            fetch(3, h3, l3);
            for (int g = 0; g < NG; g += 4) {
                step(g, h0, l0);
                fetch(g + 4, h0, l0);
                step(g + 1, h1, l1);
                fetch(g + 5, h1, l1);
                step(g + 2, h2, l2);
                fetch(g + 6, h2, l2);
                step(g + 3, h3, l3);
                fetch(g + 7, h3, l3);
            }
        }
    } else {
    for (int t = wave; t < T; t += GATE_NW) {
        const f16x8 *ah = (const f16x8 *)imgH + (size_t)t * S * 64 + lane;
        const f16x8 *al = (const f16x8 *)imgL + (size_t)t * S * 64 + lane;
        f32x16 acc[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
        for (int s0 = 0; s0 < S; ++s0) {
            const f16x8 vh = ah[s0 * 64], vl = al[s0 * 64];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f16x8 bh = xh[cb * XCB + s0 * 64], bl = xh[cb * XCB + XLO + s0 * 64];
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, bh, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, bl, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, bh, acc[cb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) epilogue(t, acc[cb], part[cb]);
    }
    }
    __syncthreads();                                   // everyone is done reading X
    float *red = X;                                    // [GATE_NW waves][CB][G][32 cells]
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float v = part[cb][g] + __shfl_xor(part[cb][g], 32);
            if (h == 0) red[((wave * CB + cb) * G + g) * 32 + c] = v;
        }
    __syncthreads();
    if (tid < 32 * G * CB) {
        const int cb = tid / (32 * G), rr = tid - cb * 32 * G;
        const int cell = rr / G, g = rr - cell * G;
        const long cg = cell0 + 32 * cb + cell;
        if (cg < ncell) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < GATE_NW; w += 2)
                v += red[((w * CB + cb) * G + g) * 32 + cell] + red[(((w + 1) * CB + cb) * G + g) * 32 + cell];
            gate[cg * G + g] = v + b2[g];
        }
    }
}

// ---- 3b. the gate's hidden layer as a tiled GEMM: cells x hidden units, the WEIGHTS resident in registers.
// router_gate_kernel above keeps a 32- or 64-cell feature tile in LDS and streams the whole hidden layer past it: every
// workgroup re-reads the 1 - 2.4 MB of hi / lo weight images from L2 (604 MB at B = 128, triple) and rebuilds its tile from
// the pooled averages.  Here a wave owns ONE 32-row tile of the hidden layer for the whole launch -- its hi / lo A fragments
// (S k-steps x 2 x 4 registers: 256 VGPRs dual, 384 triple; one wave per SIMD, 512 registers) are loaded once -- and the cells
// stream past it: the feature images the pooling pass wrote (gate_pool_kernel<true>) move through an LDS ring of 16-KiB chunks
// (8 k-steps, hi | lo) by LDS-DMA, shared by the workgroup's four waves (four row tiles = 128 hidden units).  Per k-step a
// wave issues two ds_read_b128 and three MFMAs; nothing else is in the loop.  grid = (cell-block groups, hidden groups of 128);
// a workgroup walks its cell blocks with stride gridDim.x.  The output layer is contracted in the epilogue (bias, activation,
// W2 rows from LDS); the partial logits of the hidden groups go to part[hg][cell][g] and are summed in a fixed order by
// gate_finalize_kernel (deterministic; no float atomics).
#ifndef DVQ_GEMM_ABL
#define DVQ_GEMM_ABL 0           // timing experiments of the tuning build only (results WRONG): 1 no ring DMA in the loop, 2 no MFMAs,
#endif                           // 4 no epilogue, 8 no B-fragment reads
#ifndef DVQ_GEMM_CH16
#define DVQ_GEMM_CH16 1          // ring chunks of 16 k-steps (4 slots) instead of 8 (8 slots): half the workgroup barriers, the kernel -2 ... -4 %
#endif                           // (same box; profiles/r05_gate_counters.json)
template <int G, int S>
__global__ __launch_bounds__(256, 1) void gate_gemm_kernel(
    const char *__restrict__ ximg, const _Float16 *__restrict__ imgH, const _Float16 *__restrict__ imgL,
    const float *__restrict__ b1, const float *__restrict__ W2, int Hid, int act, long ncell, int nblocks,
    const float *__restrict__ xs, float *__restrict__ part, const float *__restrict__ wtail)
{
    constexpr int CH = (DVQ_GEMM_CH16 && S % 16 == 0) ? 16 : ((S % 8 == 0) ? 8 : 4);   // k-steps per ring chunk
    constexpr int NCH = S / CH;                              // chunks per cell block
    constexpr int CHB = CH * 2048;                           // bytes per chunk (hi + lo)
    constexpr int RING = (CH == 16) ? 4 : 8;                 // 128 KiB of ring either way
    constexpr int PPW = CH * 2 / 4;                          // 1-KiB DMA pieces per wave and chunk
    static_assert(S % CH == 0 && PPW >= 1, "S is a multiple of 4");
    extern __shared__ __attribute__((aligned(16))) char glds[];       // [RING][CHB] | PB [4 waves][1 + G][32] | red [4][G][32]
    float *PB = (float *)(glds + RING * CHB);
    float *red = PB + 4 * (1 + G) * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int T = (Hid + 31) / 32;
    const int hg = blockIdx.y;
    const int t_raw = hg * 4 + wave;
    const bool tile_ok = t_raw < T;
    const int t = tile_ok ? t_raw : T - 1;
    const int nmine = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // cell blocks of this workgroup
    const int total = nmine * NCH;                           // chunks it streams

    const char *isrc = ximg;                                 // source / ring address of the chunk being issued (this wave's pieces)
    char *idst = glds;
    auto issue_begin = [&](int q) {                          // chunk q of this workgroup's stream; past the end: harmless repeat
        const int qq = q < total ? q : total - 1;
        const int bi = (int)blockIdx.x + (qq / NCH) * (int)gridDim.x;
        isrc = ximg + ((size_t)bi * S + (size_t)(qq % NCH) * CH) * 2048 + (wave * PPW) * 1024 + lane * 16;
        idst = glds + (q & (RING - 1)) * CHB + (wave * PPW) * 1024;
    };
    auto issue_piece = [&](int p_) { glds16(isrc + p_ * 1024, idst + p_ * 1024); };
    auto issue = [&](int q) {
        issue_begin(q);
#pragma unroll
        for (int p_ = 0; p_ < PPW; ++p_) issue_piece(p_);
    };
    if (total <= 0) return;
#pragma unroll
    for (int q = 0; q < RING - 2; ++q) issue(q);
    // this wave's rows of the hidden layer: A fragments, once
    f16x8 ah[S], al[S];
    {
        const f16x8 *gh = (const f16x8 *)imgH + (size_t)t * S * 64 + lane;
        const f16x8 *gl = (const f16x8 *)imgL + (size_t)t * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) { ah[s] = gh[s * 64]; al[s] = gl[s * 64]; }
    }
    // bias and output-layer rows of this wave's 32 hidden units (rows past Hid: zeros -> contribute nothing)
    for (int i = lane; i < (1 + G) * 32; i += 64) {
        const int row = i >> 5, j = t * 32 + (i & 31);
        float v = 0.0f;
        if (j < Hid && tile_ok) v = (row == 0) ? b1[j] : W2[(size_t)(row - 1) * Hid + j];
        PB[wave * (1 + G) * 32 + i] = v;
    }
    const float inv_scale = xs[0] * wtail[DVQ_GATE_WINV];    // feature scale (gate_pool_kernel) and weight scale (w1_split_kernel)
    const float *pb = PB + wave * (1 + G) * 32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // A fragments and the first RING - 2 chunks (mine) are here

    // The k-steps of all of this workgroup's cell blocks form ONE stream.  B fragments (hi, lo) are read from the ring four
    // k-steps ahead of their MFMAs with hand-placed ds_read_b128 behind a counted lgkmcnt (a lone wave per SIMD has nobody to
    // hide an LDS round trip behind); when the read-ahead enters a new chunk, the chunk's arrival is waited for (counted vmcnt
    // + barrier) and the chunk RING - 2 further on is issued into the slot everybody left two chunks ago.
    const unsigned lds_lane = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)glds + lane * 16;
    f16x8 bh[4], bl[4];
#define GG_RD(U, KOFF)                                                                                                  \
    asm volatile("ds_read_b128 %0, %2 offset:%c3\n\tds_read_b128 %1, %2 offset:%c4"                                      \
                 : "=v"(bh[(U) & 3]), "=v"(bl[(U) & 3]) : "v"(rd_base), "i"((KOFF) * 2048), "i"((KOFF) * 2048 + 1024))
    unsigned rd_base;                                        // LDS address of the chunk the read-ahead is in (+ lane * 16)
    auto enter_chunk = [&](int qn) {                         // the read-ahead moves into chunk qn
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (RING - 3)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_begin(qn + RING - 2);                          // its pieces ride between the MFMAs of the next k-steps
        if (!(DVQ_GEMM_ABL & 1)) issue_piece(0);
        rd_base = lds_lane + (unsigned)((qn & (RING - 1)) * CHB);
    };
    __builtin_amdgcn_s_barrier();                            // everybody's pieces of the first chunks (and PB) are in LDS
    asm volatile("" ::: "memory");
    if (!(DVQ_GEMM_ABL & 1)) issue(RING - 2);
    rd_base = lds_lane;
    GG_RD(0, 0); GG_RD(1, 1); GG_RD(2, 2); GG_RD(3, 3);
    // Epilogue of a cell block = bias, activation, contraction with the output layer: register r of lane half h is hidden row
    // (r & 3) + 8 (r >> 2) + 4 h of the tile, cell c.  SiLU = y / (1 + 2^(-y log2 e)) on v_exp_f32 / v_rcp_f32 (~2 ulp; the logits'
    // tolerance is 1e-4): the IEEE division sequence and a branch per element were a quarter of the kernel.  PIPE (dual router,
    // >= 20 k-steps): the ~150 vector instructions of block i's epilogue ride in the MFMA shadows of block i + 1's first 17
    // k-steps, one accumulator row per k-step, with the bias / output-layer rows resident in registers (a lone wave per SIMD
    // has nothing else to fill the matrix pipe with meanwhile: 0.6 us per block otherwise).
    constexpr bool PIPE = (G == 2) && (S >= 20);
    float cb[16], cw[G][16];
    if (PIPE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
            cb[r] = pb[j];
#pragma unroll
            for (int g = 0; g < G; ++g) cw[g][r] = pb[(1 + g) * 32 + j];
        }
    }
    auto store_part = [&](int ib, const float (&pt)[G]) {    // this wave's (row tile's) slice part[4 hg + wave][cell][g]: no workgroup
        const long cgl = ((long)blockIdx.x + (long)ib * gridDim.x) * 32 + c;      // barrier; gate_finalize_kernel adds the slices in order
        float *po = part + ((size_t)(hg * 4 + wave) * ncell + cgl) * G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float v = pt[g] + __shfl_xor(pt[g], 32);
            if (h == 0 && cgl < ncell) po[g] = v;
        }
    };
    auto run = [&](auto silu_tag) {
        constexpr bool SILU = decltype(silu_tag)::value;
        auto activ = [&](float yv) -> float {
            return SILU ? yv * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * yv)) : (yv > 0.0f ? yv : 0.0f);
        };
        f32x16 accp;                                         // PIPE: the previous block's accumulators
        float ptp[G];
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) ptp[g] = 0.0f;
        auto epi_row = [&](int r) {
            const float hv = activ(__builtin_fmaf(accp[r], inv_scale, cb[r]));
#pragma unroll
            for (int g = 0; g < G; ++g) ptp[g] = __builtin_fmaf(cw[g][r], hv, ptp[g]);
        };
        auto block = [&](int i, auto prev_tag) {
            constexpr bool PREV = decltype(prev_tag)::value;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int sk = 0; sk < S; ++sk) {
                const int u = sk + 4;                        // read-ahead: k-step sk + 4 of this block = k-step (sk + 4) % S of block i or i + 1
                // four pairs (k-steps sk .. sk + 3) are in flight: the oldest has landed when at most six reads are outstanding
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bh[sk & 3]), "+v"(bl[sk & 3]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (!(DVQ_GEMM_ABL & 2)) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sk], bh[sk & 3], acc, 0, 0, 0);     // small terms first
                    __builtin_amdgcn_sched_barrier(0);
                    // one ring piece per k-step, behind an MFMA that keeps the pipe busy while the DMA instruction issues
                    if (u % CH > 0 && u % CH < PPW && !(DVQ_GEMM_ABL & 1)) issue_piece(u % CH);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sk], bl[sk & 3], acc, 0, 0, 0);
                    if (PIPE && PREV && sk < 16 && !(DVQ_GEMM_ABL & 4)) { __builtin_amdgcn_sched_barrier(0); epi_row(sk); __builtin_amdgcn_sched_barrier(0); }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sk], bh[sk & 3], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (PIPE && PREV && sk == 16 && !(DVQ_GEMM_ABL & 4)) {
                    store_part(i - 1, ptp);
#pragma unroll
                    for (int g = 0; g < G; ++g) ptp[g] = 0.0f;
                    __builtin_amdgcn_sched_barrier(0);
                }
                // k-step sk + 4 takes over the pair's registers (an output of these reads must always have a later use -- here
                // the counted wait four k-steps on, at the end of a block the drain below: hipcc reuses the registers of a dead
                // asm output while the read is still in flight)
                if (u % CH == 0) enter_chunk(i * NCH + u / CH);  // (past the last chunk: the ring repeats the last one)
                GG_RD(u, u % CH);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DVQ_GEMM_ABL & 4) { if (acc[0] == 12345.678f) part[0] = acc[1]; return; }
            if (PIPE) {
                accp = acc;                                  // consumed during the next block (or after the loop)
                return;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]), "+v"(bh[2]), "+v"(bl[2]),
                         "+v"(bh[3]), "+v"(bl[3]) :: "memory");  // the next block's first four pairs: landed before other LDS traffic
            float pt[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pt[g] = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 bb = *(const f32x4 *)(pb + 8 * q4 + 4 * h);
                f32x4 ww[G];
#pragma unroll
                for (int g = 0; g < G; ++g) ww[g] = *(const f32x4 *)(pb + (1 + g) * 32 + 8 * q4 + 4 * h);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float hv = activ(acc[4 * q4 + r4] * inv_scale + bb[r4]);
#pragma unroll
                    for (int g = 0; g < G; ++g) pt[g] = __builtin_fmaf(ww[g][r4], hv, pt[g]);
                }
            }
            store_part(i, pt);
        };
        if constexpr (PIPE) {
            block(0, std::false_type{});
            for (int i = 1; i < nmine; ++i) block(i, std::true_type{});
        } else {
            for (int i = 0; i < nmine; ++i) block(i, std::false_type{});
        }
        // drain the read-ahead (its last four pairs are never consumed) and, PIPE, the last block's epilogue
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]), "+v"(bh[2]), "+v"(bl[2]),
                     "+v"(bh[3]), "+v"(bl[3]) :: "memory");
        if (PIPE && !(DVQ_GEMM_ABL & 4)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) epi_row(r);
            store_part(nmine - 1, ptp);
        }
    };
    if constexpr (PIPE) {
        if (act == 1) run(std::true_type{});
        else run(std::false_type{});
    } else {
        // one copy of the k-loop (the triple router's 384 fragment registers leave no room for a second live one); the
        // activation is chosen inside the epilogue
        for (int i = 0; i < nmine; ++i) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int sk = 0; sk < S; ++sk) {
                const int u = sk + 4;
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bh[sk & 3]), "+v"(bl[sk & 3]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (!(DVQ_GEMM_ABL & 2)) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sk], bh[sk & 3], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u % CH > 0 && u % CH < PPW && !(DVQ_GEMM_ABL & 1)) issue_piece(u % CH);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sk], bl[sk & 3], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sk], bh[sk & 3], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (u % CH == 0) enter_chunk(i * NCH + u / CH);
                GG_RD(u, u % CH);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DVQ_GEMM_ABL & 4) { if (acc[0] == 12345.678f) part[0] = acc[1]; continue; }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]), "+v"(bh[2]), "+v"(bl[2]),
                         "+v"(bh[3]), "+v"(bl[3]) :: "memory");
            float pt[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pt[g] = 0.0f;
            auto contract = [&](auto silu_tag) {
                constexpr bool SILU = decltype(silu_tag)::value;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 bb = *(const f32x4 *)(pb + 8 * q4 + 4 * h);
                    f32x4 ww[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) ww[g] = *(const f32x4 *)(pb + (1 + g) * 32 + 8 * q4 + 4 * h);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float yv = acc[4 * q4 + r4] * inv_scale + bb[r4];
                        const float hv = SILU ? yv * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * yv))
                                              : (yv > 0.0f ? yv : 0.0f);
#pragma unroll
                        for (int g = 0; g < G; ++g) pt[g] = __builtin_fmaf(ww[g][r4], hv, pt[g]);
                    }
                }
            };
            if (act == 1) contract(std::true_type{});
            else contract(std::false_type{});
            store_part(i, pt);
        }
    }
    (void)red;
#undef GG_RD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // surplus ring DMA
}

// gate[cell][g] = b2[g] + the row tiles' partial logits in tile order (slices past the hidden layer's last tile hold zeros)
__global__ __launch_bounds__(256) void gate_finalize_kernel(const float *__restrict__ part, int HG, long n, int G,
                                                            const float *__restrict__ b2, float *__restrict__ gate)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.0f;
    for (int h0 = 0; h0 < HG; h0 += 24) {                    // 24 independent loads at a time (dual 16 slices, triple 24: ONE round
        float pv[24];                                        // trip to the memory the other XCDs' workgroups wrote), added in slice order
#pragma unroll
        for (int k = 0; k < 24; ++k) pv[k] = (h0 + k < HG) ? part[(size_t)(h0 + k) * n + i] : 0.0f;
#pragma unroll
        for (int k = 0; k < 24; ++k) v += pv[k];
    }
    gate[i] = v + b2[i % G];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static size_t align256r(size_t x) { return (x + 255) / 256 * 256; }

// ws: [folded GroupNorm affine B*F float2][pooled averages B*F*hc*wc floats][W1 hi image | W1 lo image: ceil(Hid/32)*32*Fp halves
// each -- used only when the caller passes no prepared images]
static size_t gate_img_bytes(int nb, int C, int Hid)
{
    const size_t Fp = ((size_t)nb * C + 15) / 16 * 16;
    return align256r((size_t)((Hid + 31) / 32) * 32 * Fp * 2 * sizeof(_Float16));
}
size_t dvq_router_gate_prep_bytes_impl(int nb, int C, int Hid) { return gate_img_bytes(nb, C, Hid) + 256; }
// the GEMM form: feature images (as many bytes as the pooled averages they replace, rounded up to whole 32-cell blocks) and the
// hidden groups' partial logits
static size_t gate_ximg_bytes(int nb, int B, int C, int hc, int wc)
{
    const size_t nblocks = ((size_t)B * hc * wc + 31) / 32;
    return align256r(nblocks * (size_t)((nb * C + 15) / 16) * 2048);
}
static size_t gate_part_bytes(int nb, int B, int hc, int wc, int Hid)
{
    return align256r((size_t)((Hid + 127) / 128) * 4 * B * hc * wc * nb * sizeof(float));       // one slice per row tile
}
size_t dvq_router_gate_ws_bytes(int nb, int B, int C, int hc, int wc, int groups, int Hid)
{
    (void)groups;
    const size_t pool = align256r((size_t)B * nb * C * hc * wc * sizeof(float));
    const size_t ximg = gate_ximg_bytes(nb, B, C, hc, wc);
    return align256r((size_t)B * nb * C * sizeof(float2)) + (pool > ximg ? pool : ximg) + gate_img_bytes(nb, C, Hid) + 256 +
           gate_part_bytes(nb, B, hc, wc, Hid) + 512;
}

// per-branch maxima of the GroupNorm parameters -> the tail of the weight prep (see gate_pool_kernel: nbound)
__global__ __launch_bounds__(256) void gate_norm_bound_kernel(const float *w0, const float *b0, const float *w1, const float *b1,
                                                              const float *w2, const float *b2, int nb, int C, float *out)
{
    __shared__ float red[6][4];
    const float *ws[3] = {w0, w1, w2}, *bs[3] = {b0, b1, b2};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int br = 0; br < 3; ++br) {
        float mw = 0.0f, mb = 0.0f;
        if (br < nb)
            for (int i = threadIdx.x; i < C; i += 256) { mw = fmaxf(mw, fabsf(ws[br][i])); mb = fmaxf(mb, fabsf(bs[br][i])); }
        // NaN parameters: fmaxf drops them -- a NaN weight poisons the logits whatever the scale, nothing to protect
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, off)); mb = fmaxf(mb, __shfl_xor(mb, off)); }
        if (lane == 0) { red[2 * br][wave] = mw; red[2 * br + 1][wave] = mb; }
    }
    __syncthreads();
    if (threadIdx.x < 6) out[1 + threadIdx.x] = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
    if (threadIdx.x == 0) out[0] = __builtin_bit_cast(float, (int)DVQ_GATE_NORM_MAGIC);
}

int dvq_launch_router_gate_prepare_norm(const float *const *gn_w, const float *const *gn_b, int nb, int C, int Hid, void *prep,
                                        hipStream_t st)
{
    float *tail = (float *)((char *)prep + gate_img_bytes(nb, C, Hid));
    hipLaunchKernelGGL(gate_norm_bound_kernel, dim3(1), dim3(256), 0, st, gn_w[0], gn_b[0], gn_w[1], gn_b[1],
                       nb == 3 ? gn_w[2] : nullptr, nb == 3 ? gn_b[2] : nullptr, nb, C, tail);
    return (int)hipGetLastError();
}

// hidden-layer weight -> split fp16 tile images (kept by the caller across calls while the weight is unchanged)
// (prep: images + the 256-byte tail, always)
int dvq_launch_router_gate_prepare(const float *W1, int nb, int C, int Hid, void *prep, hipStream_t st)
{
    const int F = nb * C, Fp = (F + 15) & ~15;
    _Float16 *imgH = (_Float16 *)prep;
    _Float16 *imgL = imgH + (size_t)((Hid + 31) / 32) * 32 * Fp;
    size_t total = (size_t)((Hid + 31) / 32) * 32 * Fp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    float *tail = (float *)((char *)prep + gate_img_bytes(nb, C, Hid));
    (void)hipMemsetAsync(tail + DVQ_GATE_WMAX, 0, sizeof(float), st);
    const size_t nw = (size_t)Hid * F;
    int mblocks = (int)((nw + 256 * 16 - 1) / (256 * 16));
    if (mblocks > 512) mblocks = 512;
    if (mblocks < 1) mblocks = 1;
    hipLaunchKernelGGL(w1_max_kernel, dim3(mblocks), dim3(256), 0, st, W1, nw, (unsigned *)(tail + DVQ_GATE_WMAX));
    hipLaunchKernelGGL(w1_split_kernel, dim3(blocks), dim3(256), 0, st, W1, Hid, F, Fp, imgH, imgL, tail);
    return (int)hipGetLastError();
}

int dvq_launch_router_gate(int nb, const float *const *h, const float *const *gn_w, const float *const *gn_b,
                           int B, int C, int hc, int wc, int groups, float eps,
                           const float *W1, const float *b1, const float *W2, const float *b2,
                           int Hid, int act, const void *w1_prep, float *gate, void *ws, hipStream_t st)
{
    DvqGateArgs a;
    for (int i = 0; i < 3; ++i) {
        a.h[i] = i < nb ? h[i] : nullptr;
        a.gn_w[i] = (i < nb && groups > 0) ? gn_w[i] : nullptr;
        a.gn_b[i] = (i < nb && groups > 0) ? gn_b[i] : nullptr;
        a.scale[i] = 1 << i;
    }
    a.nb = nb; a.B = B; a.C = C; a.hc = hc; a.wc = wc; a.groups = groups; a.eps = eps;
    a.vec = 1;
#ifdef DVQ_TUNING
    {
        const char *e = getenv("DVQ_GATE_POOL_SCALAR");       // tuning build only: the scalar pooling form for A/B
        if (e && e[0] == '1') a.vec = 0;
    }
#endif
    const int F = nb * C, Fp = (F + 15) & ~15;
    // features that fit the memory-side cache are read with plain loads (pass 1 reads them again right after)
    const int fs = 1 << (nb - 1);
    const bool cached = (size_t)B * C * hc * fs * wc * fs * sizeof(float) <= DVQ_CACHED_MAX_BYTES;
    float2 *stats = (float2 *)ws;                            // (scale, shift) per (image, feature)
    float *pool = (float *)((char *)ws + align256r((size_t)B * F * sizeof(float2)));
    // ws: [folded GroupNorm affine][pooled averages (direct form) OR feature images (GEMM form)][W1 images, when the caller
    //      passes none][partial logits of the GEMM form][its scale word]
    const long ncell = (long)B * hc * wc;
    const size_t pool_area = align256r((size_t)B * F * hc * wc * sizeof(float));
    const size_t ximg_area = gate_ximg_bytes(nb, B, C, hc, wc);
    char *after = (char *)pool + (pool_area > ximg_area ? pool_area : ximg_area);
    const _Float16 *imgH = (const _Float16 *)w1_prep;
    if (act != 0 && imgH == nullptr) {
        int rc = dvq_launch_router_gate_prepare(W1, nb, C, Hid, after, st);
        if (rc) return rc;
        imgH = (const _Float16 *)after;
    }
    const _Float16 *imgL = imgH ? imgH + (size_t)((Hid + 31) / 32) * 32 * Fp : nullptr;
    const float *wtail = imgH ? (const float *)((const char *)imgH + gate_img_bytes(nb, C, Hid)) : nullptr;   // act == 0: unused
    // the GEMM form (gate_gemm_kernel): a hidden layer with GroupNorm'd inputs whose groups are whole octets of channels that fit the
    // pooling workgroup's LDS, F a multiple of 16 with an instantiated number of k-steps
    const int cpg = groups > 0 ? C / groups : 0;
    const int S16 = F / 16;
    const bool gemm_form = act != 0 && groups > 0 && cpg % 8 == 0 && cpg <= 64 && (size_t)nb * cpg * hc * wc * 4 <= 48 * 1024 &&
                           F % 16 == 0 && (S16 == 8 || S16 == 12 || S16 == 16 || S16 == 24 || S16 == 32 || S16 == 48);
    if (gemm_form) {
        // (prepared images carry the GroupNorm maxima in their tail when dvq_router_gate_prepare_norm_f32 ran; images rebuilt inside
        // this call do not)
        const float *nbound = (w1_prep != nullptr) ? (const float *)((const char *)w1_prep + gate_img_bytes(nb, C, Hid)) : nullptr;
        char *ximg = (char *)pool;
        const _Float16 *imgLg = imgL;
        float *part = (float *)(after + gate_img_bytes(nb, C, Hid) + 256);
        float *xs = (float *)((char *)part + gate_part_bytes(nb, B, hc, wc, Hid));
        const int nblocks = (int)((ncell + 31) / 32);
        if ((ncell & 31) != 0)                               // the last block's unused cell columns must hold finite values
            (void)hipMemsetAsync(ximg + (size_t)(nblocks - 1) * S16 * 2048, 0, (size_t)S16 * 2048, st);
        if (cached)
            hipLaunchKernelGGL((gate_pool_kernel<true, false>), dim3(B * groups), dim3(256), (size_t)nb * cpg * hc * wc * sizeof(float), st, a,
                               stats, pool, ximg, xs, nbound);
        else
            hipLaunchKernelGGL((gate_pool_kernel<true, true>), dim3(B * groups), dim3(256), (size_t)nb * cpg * hc * wc * sizeof(float), st, a,
                               stats, pool, ximg, xs, nbound);
        int ncu = 256;
        {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            ncu = n > 0 ? n : 256;
        }
        const int HG = (Hid + 127) / 128;
        int CG = ncu / HG;
        if (CG < 1) CG = 1;
        if (CG > nblocks) CG = nblocks;
#define DVQ_GEMM_LAUNCH(GG, SS)                                                                                        \
        do {                                                                                                           \
            static unsigned long long done_ = 0;                                                                       \
            constexpr int CH_ = (DVQ_GEMM_CH16 && SS % 16 == 0) ? 16 : ((SS % 8 == 0) ? 8 : 4);                         \
            constexpr int CHB_ = CH_ * 2048;                                                                           \
            const size_t shm = (CH_ == 16 ? 4 : 8) * (size_t)CHB_ + (4 * (1 + GG) * 32 + 4 * GG * 32) * sizeof(float); \
            int rc = dvq_allow_dynamic_lds((const void *)gate_gemm_kernel<GG, SS>, (int)shm, &done_);                  \
            if (rc) return rc;                                                                                         \
            hipLaunchKernelGGL((gate_gemm_kernel<GG, SS>), dim3(CG, HG), dim3(256), shm, st, ximg, imgH, imgLg, b1, W2, \
                               Hid, act, ncell, nblocks, xs, part, wtail);                                             \
        } while (0)
        if (nb == 2) {
            switch (S16) {
            case 8:  DVQ_GEMM_LAUNCH(2, 8); break;
            case 16: DVQ_GEMM_LAUNCH(2, 16); break;
            case 24: DVQ_GEMM_LAUNCH(2, 24); break;
            case 32: DVQ_GEMM_LAUNCH(2, 32); break;
            default: return -1000;
            }
        } else {
            switch (S16) {
            case 12: DVQ_GEMM_LAUNCH(3, 12); break;
            case 24: DVQ_GEMM_LAUNCH(3, 24); break;
            case 48: DVQ_GEMM_LAUNCH(3, 48); break;
            default: return -1000;
            }
        }
#undef DVQ_GEMM_LAUNCH
        const long nout = ncell * nb;
        hipLaunchKernelGGL(gate_finalize_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, part, HG * 4, nout, nb, b2, gate);
        return (int)hipGetLastError();
    }
    if (cached)
        hipLaunchKernelGGL((gate_pool_kernel<false, false>), dim3(B * (groups > 0 ? groups : C / 8)), dim3(256), 0, st, a, stats, pool,
                           nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((gate_pool_kernel<false, true>), dim3(B * (groups > 0 ? groups : C / 8)), dim3(256), 0, st, a, stats, pool,
                           nullptr, nullptr, nullptr);
    // two blocks of 32 cells per workgroup when the split tile of 64 cells fits the LDS and there are enough cells to keep
    // every CU busy that way
    const size_t shmem2 = ((size_t)2 * 32 * Fp + (size_t)(1 + nb) * Hid) * sizeof(float);
    int ncu = 256;                                           // of the CURRENT device (no process-wide cache)
    {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        ncu = n > 0 ? n : 256;
    }
    const bool cb2 = act != 0 && shmem2 <= 160 * 1024 - 512 && ncell >= 64L * ncu;
    const int CBv = cb2 ? 2 : 1;
    const unsigned grid = (unsigned)((ncell + 32 * CBv - 1) / (32 * CBv));
    const size_t shmem = cb2 ? shmem2 : ((size_t)32 * Fp + (size_t)(1 + nb) * Hid) * sizeof(float);
#define DVQ_GATE_LAUNCH(GG, CC)                                                                                        \
    do {                                                                                                               \
        static unsigned long long done_ = 0;                                                                           \
        int rc = dvq_allow_dynamic_lds((const void *)router_gate_kernel<GG, CC>, 160 * 1024 - 256, &done_);            \
        if (rc) return rc;                                                                                             \
        hipLaunchKernelGGL((router_gate_kernel<GG, CC>), dim3(grid), dim3(GATE_NW * 64), shmem, st, a, stats, pool,    \
                           imgH, imgL, b1, W2, b2, Hid, act, gate, wtail);                                             \
    } while (0)
    if (nb == 2) { if (cb2) DVQ_GATE_LAUNCH(2, 2); else DVQ_GATE_LAUNCH(2, 1); }
    else         { if (cb2) DVQ_GATE_LAUNCH(3, 2); else DVQ_GATE_LAUNCH(3, 1); }
#undef DVQ_GATE_LAUNCH
    return (int)hipGetLastError();
}
