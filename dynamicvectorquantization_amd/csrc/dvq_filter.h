// dvq_filter.h -- declarations shared by the fp16-filter assign kernels (vq_assign_filter.hip: pass 1,
// resolver), the conv-folded codebook (vq_fold.hip), the routing prepass of the exact routed mode (vq_assign_routed.hip), the 1x1 conv with the select
// fused in (qconv.hip) and the exact kernel's routed list mode (vq_assign_exact.hip).
#pragma once
#include "dvq_common.h"

struct DvqF16Meta {
    int ok;         // 1: codebook finite and representable; 0: every token goes to the exact list
    int b_exp;      // eh = fp16(2^b e),  2^b max|e| in [2^14, 2^15)
    float scale_b;  // 2^b
    float emax;     // >= max_j ||e_j||
    float enmax;    // max_j en_j
    float etamax;   // >= max_j ||2^b e_j - eh_j||
    float pad[10];
};

static constexpr float GAMMA_P = 1.2207031e-4f;   // 2^-13
static constexpr float PACK_E = 1.93e-6f;         // 2^-19 (1 + margin)
static constexpr float REF_XN = 1.2e-7f;          // 2u
static constexpr float REF_RE = 1.6e-5f;          // u + gamma_256 (D <= 256)
static constexpr float DVQ_SEED_PAD = -3.0e38f;
static constexpr int RES_SLOTS = 32;              // resolver: queued tokens per workgroup
static constexpr int RES_CAND = 512;              // resolver: candidate pairs per workgroup

// record of one queued token (written by pass 1, read by the resolver)
//   [zf: D*4 B in channel order][meta 32 B]   (the resolver re-derives the fp16 fragments: same RNE conversion)
__host__ __device__ inline size_t rec_bytes(int D) { return (size_t)D * 4 + 32; }
//   n:    output position of the token (b*HW + hw), also what the exact list carries for it
//   m:    the token's loss weight (codebook_mask value; the resolver must not read the mask tensor: in the routed op pass 1
//         writes it in the same launch)
//   best: merged (distance, code) key of the sliced resolver (large K), ~0 = none yet; written ~0 by pass 1
//   rep:  the token stands for rep x rep output positions (n = the first; rows Wout apart): the copies of one coarse-cell vector
//         in the routed op, which all get the resolver's correction
struct RecMeta { int n; float xn; float thr; float m; unsigned long long best; int prov; int rep; };   // 32 B

// The bound W on |G - truth| (derivation: header of vq_assign_filter.hip);
// returns 2 W (1 + margin), NaN for a token the fp16 path cannot score.
__device__ __forceinline__ float dvq_filter_threshold(float xn, float amax, float zeta2, float sB,
                                                      const DvqF16Meta *__restrict__ meta)
{
    const float emax = meta->emax, enmax = meta->enmax, etamax = meta->etamax;
    const bool bad = !(xn < __builtin_inff()) || !(amax < 60000.0f) || !meta->ok
                     || !((0.5f * sB * enmax) < 1.0e37f);
    const float zeta = sqrtf(zeta2) * 1.001f;
    const float Rh = sqrtf(xn) * 1.00001f;
    const float zn_ = Rh + zeta;
    const float ehn = sB * emax + etamax;
    const float Wv = zeta * ehn + zn_ * etamax
                     + GAMMA_P * (zn_ * ehn + 0.5f * sB * enmax)
                     + PACK_E * sB * (Rh * emax + 0.5f * enmax)
                     + sB * (REF_XN * (xn + enmax) + REF_RE * Rh * emax);
    return bad ? __builtin_nanf("") : 2.0f * Wv * 1.001f;
}

// ---------------------------------------------------------------------------------------------
// FOLD: the model's 1x1 quant_conv folded into the codebook (vq_fold.hip; opt-in, loss-free inference / stage-2 tokenisation).
// With h = W x + bias,  h.e_j - en_j/2  =  x.(W^T e_j) + (bias.e_j - en_j/2): pass 1 scores the conv's INPUT x against the
// image of E' = E W with accumulator seeds 2^b' (bias.e_j - en_j/2) and never computes h; only tokens that are not provably
// decided get their h (resolver / exact-list kernel: the split-fp16 conv arithmetic of qconv.hip, then the reference chain).
// A token is final when best - second > 2 W', where W' bounds |G_j - truth_j| for EVERY h within the conv's tolerance of
// the real-number conv (|h - (W x + bias)|_o <= 1e-5 sum_k |W_ok||x_k|, the contract of dvq_qconv_f32), truth_j being the
// reference's fp32 distance chain evaluated on that h:
//   fp16 rounding of x and of 2^b' e'_j (actual residual norms)       zeta (2^b' emax' + eta') + (||x|| + zeta) eta'
//   MFMA fp32 accumulation                                             gamma' ((||x|| + zeta)(2^b' emax' + eta') + seedmax)
//   index bits in the mantissa                                         2^-19 (2^b' ||x|| emax' + seedmax)
//   seed rounded to fp32 (E' and the seeds are computed in fp64)       u seedmax
//   conv tolerance: |delta.e_j| <= 1e-5 (||x|| || |W|^T |e_j| || + ||bias|| ||e_j||)      2^b' 1e-5 (||x|| qmax + bnorm emax)
//   reference side on h, ||h|| <= sigma ||x|| (1 + 2e-4) + ||bias||    2^b' [2u (||h||^2 + enmax) + (u + gamma_D) ||h|| emax]
// (DvqFoldMeta shares its first six fields with DvqF16Meta, so pass 1 / the resolver read the scale the same way.)
// ---------------------------------------------------------------------------------------------
struct DvqFoldMeta {
    int ok;         // 1: codebook, conv weight, bias and E W finite and representable; 0: every token goes to the exact list
    int b_exp;      // eh' = fp16(2^b' e'),  2^b' max|e'| in [2^14, 2^15)
    float scale_b;  // 2^b'
    float emax;     // >= max_j ||e'_j||,  e'_j = W^T e_j
    float enmax;    // max_j en_j (the codebook's own)
    float etamax;   // >= max_j ||2^b' e'_j - eh'_j||
    float seedmax;  // >= max_j |2^b' (bias.e_j - en_j/2)|
    float qmax;     // >= max_j || |W|^T |e_j| ||
    float sigma;    // >= ||W||_2: min(||W||_F, sqrt(||W^T W||_inf))
    float bnorm;    // >= ||bias||_2
    float emax0;    // >= max_j ||e_j||
    float pad[5];
};
static_assert(sizeof(DvqFoldMeta) == sizeof(DvqF16Meta), "pass 1 reads either through the same pointer");
static constexpr float CONV_TOL = 1.0e-5f;        // contract of the conv's output, relative to sum |w||x| (dvq.h: dvq_qconv_f32)

__device__ __forceinline__ float dvq_fold_threshold(float xn, float amax, float zeta2, float sB,
                                                    const DvqFoldMeta *__restrict__ meta)
{
    const float emaxp = meta->emax, enmax = meta->enmax, etamax = meta->etamax, seedmax = meta->seedmax;
    const bool bad = !(xn < __builtin_inff()) || !(amax < 60000.0f) || !meta->ok || !(seedmax < 1.0e37f);
    const float zeta = sqrtf(zeta2) * 1.001f;
    const float Rx = sqrtf(xn) * 1.00001f;
    const float xn_ = Rx + zeta;
    const float ehn = sB * emaxp + etamax;
    const float Hn = meta->sigma * Rx * 1.0002f + meta->bnorm;
    const float Wv = zeta * ehn + xn_ * etamax
                     + GAMMA_P * (xn_ * ehn + seedmax)
                     + PACK_E * (sB * Rx * emaxp + seedmax)
                     + 6.1e-8f * seedmax
                     + sB * (CONV_TOL * 1.001f) * (Rx * meta->qmax + meta->bnorm * meta->emax0)
                     + sB * (REF_XN * (Hn * Hn * 1.00001f + enmax) + REF_RE * Hn * meta->emax0);
    return bad ? __builtin_nanf("") : 2.0f * Wv * 1.001f;
}

__device__ __forceinline__ float vmax_raw(float a, float b)
{
    float r;     // plain v_max_f32: no canonicalising pre-ops (fmaxf() adds two per call)
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float vmax_abs(float a, float b)
{
    float r;     // max(a, |b|) in one instruction (source modifier instead of a separate v_and)
    asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float vmax3_raw(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---------------------------------------------------------------------------------------------
// Routed token view: a dual / triple granularity batch addressed straight in the encoder branches (no
// h_dual / h_triple tensor exists).  One token per OUTPUT POSITION: the select is fused into a dense assign,
// every wave owns whole output rows, and the grain of a position's cell comes straight from the gate
// (pass 1) or from `indices` (exact-list kernel, resolver).
//
// Grain type g (0 coarse, [1 median,] G-1 fine) owns a source tensor src[g] [B, D, sub_g*hc, sub_g*wc]
// (sub = tokens per coarse-cell edge: dual 1 / 2, triple 1 / 2 / 4); an element of it is the value of
// rep_g x rep_g output positions (rep = SC / sub, SC = sub of the fine type).
// (Scoring only the unique tokens -- 640 of 1024 per image at fine ratio 0.5 -- was built three times in round 2 and
// measured slower every time: the z_q lines are then assembled from 8-byte pieces by different instructions; see
// DESIGN.md section 4.3 and profiles/archive/r02_*dedup*.)
// ---------------------------------------------------------------------------------------------
#define DVQ_ROUTE_MAX_CELLS 1024

struct DvqRouted {
    const float *src[3];
    int sub[3];               // tokens per coarse-cell edge
    int rep[3];               // output positions per token edge
    int G, B, D, hc, wc;
    int Wout, HWout;          // output grid: SC*wc, (SC*hc)*(SC*wc)
    const long long *indices; // [B, hc, wc] grain index per coarse cell (what the list kernel re-derives a token's source from)
    const void *gate;         // router output the grain is derived from (pass 1)
    int gate_mode;            // 0 f32 logits [.., G], 1 int64 [.., G], 2 f32 entropy map + thr
    float thr;
    long long *indices_out;   // outputs pass 1 writes itself (null: a prepass wrote them)
    float *cmask_out;
    long long *gate_out;
};

// The G gate values of one cell, fetched (fetch) and reduced to the grain index (reduce) separately so that a kernel
// can put other memory operations between the two.  Reduce = argmax with torch semantics (first maximal value wins,
// NaN counts as the maximum); mode 2: entropy > thr (NaN compares false -> 0).
// ---- 1x1 quant_conv (qconv.hip) shared with the CONV form of pass 1 ----------------------------------------------------
struct QconvMeta {
    int ok;            // weight finite
    int b_exp;
    float scale_w;     // 2^bw, 2^bw max|W| in [2^13, 2^14)
    float inv_scale_w;
    float pad[12];
};
__host__ __device__ inline size_t qconv_tile_bytes(int D) { return (size_t)2 * (D / 16) * 1024 + 256; }
// MFMA row rho (0..31) of a weight tile -> output channel inside the tile: row = (r & 3) + 8 (r >> 2) + 4 h holds
// channel 16 (r >> 3) + 8 h + (r & 7)  (r = accumulator register 0..15, h = lane half)
__host__ __device__ inline int qconv_row_channel(int rho)
{
    const int r = (rho & 3) | ((rho >> 3) << 2), h = (rho >> 2) & 1;
    return 16 * (r >> 3) + 8 * h + (r & 7);
}
// the conv fused into pass 1: weight images + meta of dvq_qconv_prepare_f32, and where the conv's output goes when it is
// needed outside the kernel: h_buf [B, D, HW] receives the rows of the tokens pass 1 hands to the exact-list kernel (all
// tokens with h_all != 0: tests)
struct DvqConv {
    const char *wimg;
    const QconvMeta *meta;
    const float *bias;     // channel order
    float *h_buf;
    int h_all;
};

// the conv folded into the codebook (vq_fold.hip): the buffer of dvq_fold_prepare_f32 (meta + the two images of E W) and
// the conv itself, which the resolver and the exact-list kernel run on the few tokens they handle
struct DvqFold {
    const char *fprep;
    DvqConv cv;            // h_buf unused
};

struct DvqGateRaw { float f[3]; long long i[3]; };

__device__ __forceinline__ DvqGateRaw dvq_gate_fetch(const void *gate, int mode, int G, size_t cell)
{
    DvqGateRaw r;
    r.f[0] = r.f[1] = r.f[2] = 0.0f;
    r.i[0] = r.i[1] = r.i[2] = 0;
    if (mode == 2) {
        r.f[0] = ((const float *)gate)[cell];
    } else if (mode == 1) {
        const long long *g = (const long long *)gate + cell * G;
        r.i[0] = g[0];
        r.i[1] = g[1];
        if (G == 3) r.i[2] = g[2];
    } else {
        const float *g = (const float *)gate + cell * G;
        r.f[0] = g[0];
        r.f[1] = g[1];
        if (G == 3) r.f[2] = g[2];
    }
    return r;
}

__device__ __forceinline__ int dvq_gate_reduce(const DvqGateRaw &r, int mode, int G, float thr)
{
    if (mode == 2) return (r.f[0] > thr) ? 1 : 0;
    int bi = 0;
    if (mode == 1) {
        long long best = r.i[0];
        if (r.i[1] > best) { best = r.i[1]; bi = 1; }
        if (G == 3 && r.i[2] > best) { best = r.i[2]; bi = 2; }
    } else {
        float best = r.f[0];
#pragma unroll
        for (int i = 1; i < 3; ++i) {
            const float v = r.f[i];
            if (i < G && ((v > best) || (v != v && best == best))) { best = v; bi = i; }
        }
    }
    return bi;
}

__device__ __forceinline__ int dvq_gate_argmax(const void *gate, int mode, int G, size_t cell, float thr)
{
    return dvq_gate_reduce(dvq_gate_fetch(gate, mode, G, cell), mode, G, thr);
}

// source of output position (y, x) of image b whose cell has grain g
__device__ __forceinline__ const float *dvq_dense_source(const DvqRouted &rv, int b, int y, int x, int g, int &stride)
{
    const int sub = rv.sub[g], rep = rv.rep[g];
    const int gwid = rv.wc * sub, plane = rv.hc * sub * gwid;
    stride = plane;
    return rv.src[g] + (size_t)b * rv.D * plane + (size_t)(y / rep) * gwid + x / rep;
}

struct DvqTok {
    const float *src;   // channel 0 of the token
    int stride;         // elements between channels
    long n;             // output position (b*HWout + y*Wout + x); code / mask index
    bool valid;
};

// token t = output position t of the batch (any value; out of range -> invalid, clamped to position 0's addresses)
__device__ __forceinline__ DvqTok dvq_routed_lookup(const DvqRouted &rv, long t)
{
    DvqTok k;
    const long total = (long)rv.B * rv.HWout;
    k.valid = t >= 0 && t < total;
    const int tt = k.valid ? (int)t : 0;
    const int b = tt / rv.HWout, pos = tt - b * rv.HWout;
    const int y = pos / rv.Wout, x = pos - y * rv.Wout;
    const int SC = rv.sub[rv.G - 1];
    const int g = (int)rv.indices[(size_t)b * rv.hc * rv.wc + (y / SC) * rv.wc + x / SC];
    k.src = dvq_dense_source(rv, b, y, x, g, k.stride);
    k.n = tt;
    return k;
}
