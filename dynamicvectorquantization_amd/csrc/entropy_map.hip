// entropy_map.hip -- fused patch-entropy map for gfx950.
//
// Replaces Entropy.forward of the reference (models/stage1_dynamic/dqvae_dual_entropy.py:13-63):
// grayscale -> 16x16 unfold -> 32-bin Gaussian-KDE histogram over [0, 1] (sigma 0.01) -> normalise
// (+1e-40) -> -sum p ln p.  The reference materialises a [B*256, 256, 32] fp32 tensor (2.1 GB at
// B = 256) and makes several passes over it.  Here the image is read from HBM exactly once (768 KiB / image)
// and the 8192 Gaussian evaluations per patch collapse to four exponentials per pixel: with b0 the bin nearest to
// the pixel value v and d = v - c_b0 (|d| <= 1/62), the kernel value at bin b0 + k is
//     exp(-(d - k/31)^2 / 2 sigma^2) = exp(-d^2 / 2 sigma^2) * exp(d / (31 sigma^2))^k * exp(-k^2 / (2 (31 sigma)^2)),
// the last factor a constant G_k (5.5e-3, 9.2e-10, 4.6e-21, 7.0e-37 for |k| = 1..4; beyond |k| = 4 the value is below
// the smallest fp32 subnormal, i.e. exactly the 0 the reference computes).  The three nearest bins (which carry the
// entropy) are evaluated directly with the reference's own fp32 bin centres, the six beyond them by the recurrence
// v_{k+1} = v_k * exp(d / (31 sigma^2)) * G_{k+1} / G_k.
//
// Round 6 form (the round-3 form -- one wave per patch, 36 dependent LDS read-modify-writes per lane
// behind bounds checks, fp32 divisions, libm expf, 1070 vector instructions per patch -- took 104 us at B = 256):
//   * a HALF-wave owns a patch (a wave = two consecutive patches, so a load instruction moves whole 128-byte lines), a
//     lane eight of its pixels; waves are PERSISTENT (at most four 40-KiB workgroups per CU) and fetch the next pair's
//     pixels before they work on the current one: the kernel is bound by vector-instruction issue + LDS round trips, and
//     with all waves of a launch in the same phase a generation of one-shot waves first waited for HBM together and then
//     computed together (in-kernel stamps: 11 k cycles of load wait in a 28 k-cycle wave);
//   * per-lane histogram COLUMNS in LDS, row stride 64 floats: lane l only ever touches bank l mod 32 -- no conflicts,
//     no atomics, a fixed summation order (deterministic).  Rows cover bins -4 .. 35, so the five bins a pixel feeds
//     (b0 - 2 .. b0 + 2) are one base address + immediate offsets: five reads in flight, five adds, five writes per
//     pixel, no bounds checks (bins outside 0 .. 31 land in rows nobody reads);
//   * the bins at distance 3 and 4 hold at most 7.6e-15 per pixel: they can only matter when the patch has (almost) no
//     pixel inside the bins' range -- then the normaliser itself is that small.  They are added in a second, wave-uniform
//     phase taken only when a patch's total mass is below 2^-10 (neglected otherwise: < 1e-10 of the mass, < 1e-8 in H)
//     and some pixel of it is near the bins' range at all (rare: the pixels' values are recomputed, not kept);
//   * pixels are processed in pairs on fp32 pairs (v_pk_mul / v_pk_add), v_exp_f32 / v_rcp_f32 on arguments whose results
//     are normal numbers (the direct bins sit within 1.5 bin widths of the pixel: >= 8e-6), a multiplication by 100 for the
//     reference's division by sigma = 0.01f (2e-8 apart); the far values come from multiplications, which keep fp32
//     subnormals (the reference's epsilon 1e-40 is one: subnormals stay enabled, hipcc's default);
//   * the nearest bin is CLAMPED to -4 .. 35 instead of range-tested: the values are those of the true gray value at the
//     bins around the clamped one, i.e. what the reference computes for them (0 far outside); the bin centres come from a
//     table held one entry per lane (ds_bpermute: the LDS crossbar, no LDS memory);
//   * the column sums are read with ds_read_b128 in a per-bin rotated order (2-way instead of 16-way bank conflicts) and the
//     three reductions of the tail are DPP row operations + one cross-row shuffle instead of five shuffles each.
// What was measured on the way (profiles/r06_entropy.json): one-shot waves of this form 60-64 us whatever the vector-
// instruction count (850 -> 640 per pair) or the workgroup size; four patches per wave 62-65; a start stagger of the four
// workgroups of a CU +1-4 %; the far bins as four more histogram rows for the patches that need them 61 (a flagged pair took
// three times as long as the others and the unluckiest waves set the kernel's time), as a second loop after the main one 71;
// this form 49 us at B = 256 in the steady state (0.51 of 8 TB/s; 13.5 us at B = 64).
// Transcendental math -> parity is to 1e-5, not bit-exact (tests/test_entropy.py).
#include "dvq_common.h"
#include <type_traits>

namespace {
constexpr int ENT_PAD = 4;                      // rows for bins -4 .. -1 in front of bin 0
constexpr int ENT_ROWS = 32 + 2 * ENT_PAD;     // bins -4 .. 35
#ifndef ENT_WAVES_N
#define ENT_WAVES_N 4
#endif
constexpr int ENT_WAVES = ENT_WAVES_N;                    // 40 rows x 256 B x 4 waves = 40 KiB per workgroup: four workgroups per CU
// G_k = exp(-k^2 / (2 (31 * 0.01)^2))
constexpr float G1 = 5.5005146e-03f, G2 = 9.1540500e-10f, G3 = 4.6092459e-21f, G4 = 7.0218754e-37f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int ENT_CT0 = 5;                      // lane k + ENT_CT0 of the table register = centre of bin k, k = -5 .. 36
}   // namespace

struct EntPixels { f32x4 r[2], g[2], b[2]; };

// the eight pixels of this lane for patch `patch` (clamped to the last one): rows ll>>2 and 8 + (ll>>2) of the patch, columns
// 4 (ll&3) .. +3 -- with the two patches of a wave side by side a load instruction covers eight image rows x 128 contiguous bytes
__device__ __forceinline__ EntPixels ent_load(const float *__restrict__ img, int patch, int npatch, int gh, int gw, int H, int W, int ll)
{
    const int pc = patch < npatch ? patch : npatch - 1;         // (an odd patch count: the last wave's second half repeats its first)
    const int b = pc / (gh * gw);
    const int pr = pc - b * gh * gw;
    const int py = pr / gw, px = pr - py * gw;
    const size_t plane = (size_t)H * W;
    const float *p = img + (size_t)b * 3 * plane + ((size_t)(py * 16 + (ll >> 2))) * W + px * 16 + (ll & 3) * 4;
    const size_t down = (size_t)8 * W;
    EntPixels x;
    x.r[0] = __builtin_nontemporal_load((const f32x4 *)p);
    x.g[0] = __builtin_nontemporal_load((const f32x4 *)(p + plane));
    x.b[0] = __builtin_nontemporal_load((const f32x4 *)(p + 2 * plane));
    x.r[1] = __builtin_nontemporal_load((const f32x4 *)(p + down));
    x.g[1] = __builtin_nontemporal_load((const f32x4 *)(p + plane + down));
    x.b[1] = __builtin_nontemporal_load((const f32x4 *)(p + 2 * plane + down));
    return x;
}
// sum over the 16 lanes of a DPP row (= a patch's lanes), every lane gets it: vector instructions only
__device__ __forceinline__ float row_sum(float x)
{
    auto dpp = [](float v, auto ctrl) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});            // quad_perm [1, 0, 3, 2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});            // quad_perm [2, 3, 0, 1]
    x += dpp(x, std::integral_constant<int, 0x141>{});           // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});           // row_mirror
    return x;
}

// two pixels of a lane at once (fp32 pairs: the multiplies / adds / subtracts issue as v_pk_* at twice the scalar rate, and the
// kernel is bound by vector-instruction issue -- 4.2 cycles per wave instruction measured): gray values, nearest bins, the five /
// nine kernel values per pixel (see the header).  `ctab` = a table held one entry per LANE: lane k + ENT_CT0 has the reference's fp32 centre of bin k, k = -5 .. 36.
struct EntPix2 { f32x2 e0, ep, em, up, dn, r1, ri, w; int b0[2]; bool nan, far; };
__device__ __forceinline__ EntPix2 ent_pixel2(f32x2 R, f32x2 G, f32x2 Bl, float ctab)
{
    EntPix2 o;
    // 0.2989 R + 0.5870 G + 0.1140 B, left to right, fp32, every product and sum rounded (:51; -ffp-contract=off)
    const f32x2 v = (R * 0.2989f + G * 0.5870f) + Bl * 0.1140f;
    o.nan = (v[0] != v[0]) || (v[1] != v[1]);
    const f32x2 x = v * 31.0f;
    // within distance 4.5 of a bin of 0 .. 31 (x in (-4.5, 35.5)) but not within 2.5: only such a pixel has anything above the
    // smallest fp32 subnormal to give to the bins at distance 3 and 4 that is not dwarfed by its own bins at distance <= 2
    // (the reference's exp(-(4.5 / 0.31)^2 / 2) = exp(-105) underflows to 0)
    const f32x2 xc = x - 15.5f;
    o.far = (fabsf(xc[0]) < 20.0f && fabsf(xc[0]) > 18.0f) || (fabsf(xc[1]) < 20.0f && fabsf(xc[1]) > 18.0f);
    // nearest bin, clamped to -4 .. 35 (a NaN gives -4): the values below are those of the TRUE gray value at the bins around
    // the clamped one, so a pixel outside the window contributes what the reference computes for it -- e.g. exp(-110) = 0 in
    // fp32 at distance 4.6 -- and nothing needs a range test (v_med3 instead of two compares, an and and a select)
    f32x2 fb, c0, cp, cm;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fb[i] = rintf(__builtin_amdgcn_fmed3f(x[i], -4.0f, 35.0f));
        o.b0[i] = (int)fb[i];
        const int a = (o.b0[i] + ENT_CT0) << 2;                   // lane b0 + 5 of the table register (a wave-wide gather, no LDS memory)
        cm[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a - 4, __builtin_bit_cast(int, ctab)));
        c0[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, ctab)));
        cp[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a + 4, __builtin_bit_cast(int, ctab)));
    }
    auto direct = [&](f32x2 d) -> f32x2 {                        // (:38-39) exp(-((v - c) / sigma)^2 / 2), d = v - c
        const f32x2 t = d * 100.0f;
        const f32x2 a = (t * t) * (-0.5f * LOG2E);
        f32x2 e;
        e[0] = __builtin_amdgcn_exp2f(a[0]); e[1] = __builtin_amdgcn_exp2f(a[1]);
        return e;
    };
    const f32x2 d = v - c0;
    o.e0 = direct(d); o.ep = direct(v - cp); o.em = direct(v - cm);
    const f32x2 ra = d * ((1.0f / (31.0f * 0.01f * 0.01f)) * LOG2E);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // |d| <= 1/62 inside the window (|ra| <= 7.6); far outside it every direct value is 0 and r1, 1 / r1 only have to stay finite
        o.r1[i] = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(ra[i], -100.0f, 100.0f));
        o.ri[i] = __builtin_amdgcn_rcpf(o.r1[i]);
        // the five near bins b0 - 2 .. b0 + 2 have rows only for b0 in -2 .. 33 (beyond, none of them is a bin of 0 .. 31)
        o.w[i] = (o.b0[i] >= -2 && o.b0[i] <= 33) ? 1.0f : 0.0f;
    }
    o.up = (o.ep * o.r1) * (G2 / G1); o.dn = (o.em * o.ri) * (G2 / G1);
    return o;
}


__global__ __launch_bounds__(64 * ENT_WAVES) void entropy_map_kernel(const float *__restrict__ img, int B, int H, int W,
                                                                    float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float hist[ENT_WAVES][ENT_ROWS * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the pair arithmetic stays in SGPRs)
    const int half = lane >> 5, ll = lane & 31;
    const int gh = H / 16, gw = W / 16;
    const int npatch = B * gh * gw;                              // (dvq_entropy_map_f32 checks that this fits)
    const int npair = (npatch + 1) / 2;
    const int stride = (int)gridDim.x * ENT_WAVES;               // persistent waves: pair, pair + stride, ...
    int pair = (int)blockIdx.x * ENT_WAVES + wave;
    if (pair >= npair) return;                                   // no workgroup barrier below: LDS use is per wave
    float *hw = hist[wave];
    float *col = hw + lane;                                      // this lane's column: bank lane mod 32 whatever the row
    const int bin = ll;                                          // lane (bin, half): the column sums of its half's patch
    float ctab;
    {
        // the reference's own bin centres (torch.linspace(0, 1, 32) in fp32: i * step below the middle, 1 - (31 - i) * step above
        // it), continued on both sides for the pixels outside [0, 1]; one entry per LANE, read by ds_bpermute
        const float f = (float)(lane - ENT_CT0), step = 1.0f / 31.0f;
        const float lo = __fmul_rn(f, step), hi = __fsub_rn(1.0f, __fmul_rn(__fsub_rn(31.0f, f), step));
        ctab = (f < 16.0f) ? lo : hi;
    }
    auto half_sum = [](float x) -> float {                       // over the 32 lanes of a half-wave (two DPP rows), every lane gets it
        x = row_sum(x);
        return x + __shfl_xor(x, 16);
    };
    EntPixels cur = ent_load(img, pair * 2 + half, npatch, gh, gw, H, W, ll);
    for (;;) {
        const int next = pair + stride;
        const bool more = next < npair;                          // wave-uniform
        EntPixels nxt = cur;
        if (more) nxt = ent_load(img, next * 2 + half, npatch, gh, gw, H, W, ll);     // in flight under this pair's arithmetic
        const int patch = pair * 2 + half;
#pragma unroll
        for (int r = 0; r < 32; ++r) col[(ENT_PAD + r) * 64] = 0.0f;                   // rows of bins 0 .. 31 (the others are never read)
        bool has_nan = false;
        bool edge = false;                                       // some pixel of this lane has its nearest bin at -4, -3, 34 or 35
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = j >> 1, e = (j & 1) * 2;
            const f32x2 R = {cur.r[q][e], cur.r[q][e + 1]}, G = {cur.g[q][e], cur.g[q][e + 1]}, Bl = {cur.b[q][e], cur.b[q][e + 1]};
            const EntPix2 x = ent_pixel2(R, G, Bl, ctab);
            has_nan = has_nan || x.nan;
            const f32x2 e0 = x.e0 * x.w, ep = x.ep * x.w, em = x.em * x.w, up = x.up * x.w, dn = x.dn * x.w;
            edge = edge || x.far;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int b0n = min(max(x.b0[i], -2), 33);
                float *row = col + (b0n + ENT_PAD - 2) * 64;     // bin b0 - 2 -> rows 0 .. 39
                const float h0 = row[0], h1 = row[64], h2 = row[128], h3 = row[192], h4 = row[256];
                row[0] = h0 + dn[i]; row[64] = h1 + em[i]; row[128] = h2 + e0[i]; row[192] = h3 + ep[i]; row[256] = h4 + up[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float *hr = hw + (ENT_PAD + bin) * 64 + half * 32;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 t = *(const f32x4 *)(hr + (((i + bin) & 7) << 2));              // rotated start: 2-way bank conflicts, not 16-way
            acc += (t[0] + t[1]) + (t[2] + t[3]);
        }
        float pdf = acc * (1.0f / 256.0f);                       // mean over the 256 pixels (:40)
        float norm = half_sum(pdf);
        // The bins at distance 3 and 4.  A pixel's value there is below 1e-9 of its value at distance 2, so they only matter for a
        // pixel with NO bin of 0 .. 31 within distance 2 -- nearest bin -4 or -3 (it feeds bins 0 and 1) or 34 or 35 (bins 31 and
        // 30) -- and only in a patch with (almost) no other mass, where the normaliser is that small.  Rare (2.7 % of the patches of
        // the section-8d images): those four sums are made in registers from recomputed pixel values -- no LDS, no second pass over
        // the image; adding them to the wave's other patch too is exact as well.  (The first form of this -- nine bins in the
        // histogram for such patches -- made a flagged pair three times as long as the others, and the waves that met three of them
        // set the kernel's time: 49 -> 61 us at B = 256.)
        const unsigned long long small = __ballot(!(norm >= 0x1p-10f)), edgel = __ballot(edge);
        if (((small & 1ull) && (edgel & 0xffffffffull)) || (((small >> 32) & 1ull) && (edgel >> 32))) {
            float f0 = 0.0f, f1 = 0.0f, f30 = 0.0f, f31 = 0.0f;
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                f32x2 R = {cur.r[0][0], cur.r[0][1]}, G = {cur.g[0][0], cur.g[0][1]}, Bl = {cur.b[0][0], cur.b[0][1]};
#pragma unroll
                for (int t = 1; t < 4; ++t)
                    if (t == j) {
                        const int q = t >> 1, e = (t & 1) * 2;
                        R = f32x2{cur.r[q][e], cur.r[q][e + 1]}; G = f32x2{cur.g[q][e], cur.g[q][e + 1]}; Bl = f32x2{cur.b[q][e], cur.b[q][e + 1]};
                    }
                const EntPix2 x = ent_pixel2(R, G, Bl, ctab);
                const f32x2 u3 = (x.up * x.r1) * (G3 / G2), d3 = (x.dn * x.ri) * (G3 / G2);
                const f32x2 u4 = (u3 * x.r1) * (G4 / G3), d4 = (d3 * x.ri) * (G4 / G3);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int b0 = x.b0[i];                      // (clamped to -4 .. 35; a pixel beyond has values that underflow to 0, as in the reference)
                    f0 += (b0 == -3 ? u3[i] : 0.0f) + (b0 == -4 ? u4[i] : 0.0f);
                    f1 += (b0 == -3 ? u4[i] : 0.0f);
                    f31 += (b0 == 34 ? d3[i] : 0.0f) + (b0 == 35 ? d4[i] : 0.0f);
                    f30 += (b0 == 34 ? d4[i] : 0.0f);
                }
            }
            f0 = half_sum(f0); f1 = half_sum(f1); f30 = half_sum(f30); f31 = half_sum(f31);
            pdf += (bin == 0 ? f0 : bin == 1 ? f1 : bin == 30 ? f30 : bin == 31 ? f31 : 0.0f) * (1.0f / 256.0f);
            norm = half_sum(pdf);
        }
        if ((__ballot(has_nan) >> (32 * half)) & 0xffffffffull) pdf = norm = __builtin_nanf("");   // a NaN pixel makes every bin NaN in the reference
        norm += 1e-40f;                                          // (:41) epsilon is an fp32 subnormal
        pdf = pdf / norm + 1e-40f;                               // (:42)
        const float term = half_sum(pdf * logf(pdf));
        if (ll == 0 && patch < npatch) out[patch] = -term;       // (:43)
        if (!more) break;
        cur = nxt;
        pair = next;
        __builtin_amdgcn_wave_barrier();                         // (the column sums were read before the next pair's zeroes are written: LDS is in order per wave)
    }
}

int dvq_launch_entropy_map(const float *img, int B, int H, int W, float *out, hipStream_t st)
{
    const long npair = ((long)B * (H / 16) * (W / 16) + 1) / 2;
    long nblk = (npair + ENT_WAVES - 1) / ENT_WAVES;
    if (nblk > 1024) nblk = 1024;                                // four 40-KiB workgroups per CU are resident: beyond that, waves loop
    hipLaunchKernelGGL(entropy_map_kernel, dim3((unsigned)nblk), dim3(64 * ENT_WAVES), 0, st, img, B, H, W, out);
    return (int)hipGetLastError();
}
