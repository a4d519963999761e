// dvq_filter.h -- declarations shared by the fp16-filter assign kernels (vq_assign_filter.hip: dense
// pass 1, resolver; vq_assign_routed.hip: routing prepass + routed / low-register pass 1) and by the
// exact kernel's routed list mode (vq_assign_exact.hip).
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
//   n:    output position of the token (b*HW + hw); a routed token covers the rep x rep block whose top-left
//         corner is n (rows Wout apart)
//   best: merged (distance, code) key of the sliced resolver (large K), ~0 = none yet; written ~0 by pass 1
//   tokid: what the exact list carries for the token (dense: n; routed: its unique-token id, slot*32 + lane)
struct RecMeta { int n; float xn; float thr; int tokid; unsigned long long best; int prov; int rep; };   // 32 B

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
// Routed token view: the unique tokens of a dual / triple granularity batch, addressed straight in
// the encoder branches (no h_dual / h_triple tensor exists).
//
// Grain type g (0 coarse, [1 median,] G-1 fine) owns a source tensor src[g] [B, D, sub_g*hc, sub_g*wc]
// (sub = tokens per coarse-cell edge: dual 1 / 2, triple 1 / 2 / 4) and covers rep_g x rep_g output
// positions per token (rep = SC / sub, SC = sub of the fine type).
//
// Token order: inside an image, row-major by the token's TOP-LEFT output position (y0, x0); images in
// batch order; no padding anywhere, types mix freely (source stride and replication are per lane).  So
// the 32 tokens of a wave (a "slot") cover one or two whole rows of the output grid: a row of a fine
// region takes its fine tokens left to right, interleaved with the coarse / median tokens whose block
// starts in that row, and every 128-B line of z_q is written completely by one wave or by two adjacent
// waves of one workgroup within a microsecond -- the L2 merges them.  (With the types in separate waves
// the 8-byte pieces of a line arrived from different CUs at different times and every line went to HBM
// as read-modify-write: measured 2.2x slower.)
// The routing prepass (routed_prepass_kernel) writes, per image, the table rank -> token (`tok`) and the
// image's token count; imgstart[b] = number of tokens before image b, imgstart[B] = all of them.
// ---------------------------------------------------------------------------------------------
#define DVQ_ROUTE_MAX_CELLS 1024

// tok entry: [15:14] grain type, [13:12] sy, [11:10] sx (position inside the coarse cell, in the type's grid),
// [9:0] coarse cell index
__host__ __device__ inline unsigned short dvq_tok_pack(int g, int sy, int sx, int cell)
{
    return (unsigned short)((g << 14) | (sy << 12) | (sx << 10) | cell);
}

struct DvqRouted {
    const float *src[3];
    int sub[3];               // tokens per coarse-cell edge
    int rep[3];               // output positions per token edge
    int G, B, D, hc, wc;
    int Wout, HWout;          // output grid: SC*wc, (SC*hc)*(SC*wc)
    const int *imgstart;      // [B + 1]
    const unsigned short *tok;   // [B][HWout]
    int dense;                // 1: one token per OUTPUT POSITION (no de-duplication; rank = position, rep = 1):
                              //    the select fused into a dense assign, every wave owns whole output rows.
                              //    No tables: the grain of a position's cell comes from `indices` (written by the
                              //    prepass or by pass 1 itself), or straight from the gate (dvq_gate_argmax)
    const long long *indices; // [B, hc, wc] grain index per coarse cell (dense form)
    const void *gate;         // router output the grain is derived from (pass 1 of the dense form)
    int gate_mode;            // 0 f32 logits [.., G], 1 int64 [.., G], 2 f32 entropy map + thr
    float thr;
    long long *indices_out;   // outputs the dense pass 1 writes itself (null: the prepass wrote them)
    float *cmask_out;
    long long *gate_out;
    const int *wgd;           // row-complete de-duplicated form (non-null): per workgroup slot [B][HWout / 128] four ints
                              // {first row of cells, rows of cells, unique tokens, 0}, written by the prepass; rows = 0:
                              // slot unused.  Downstream (resolver, list kernel) sees the dense view (dense = 1).
};

// row-complete de-duplication packs whole rows of cells into one pass-1 workgroup: at most this many unique tokens
// (its 4 waves x 32 token lanes) and this many output positions (the z_q staging buffer in LDS)
#define DVQ_RD_MAX_TOKENS 128
#define DVQ_RD_MAX_POS 512

// argmax over the G gate values of one cell with torch semantics (first maximal value wins, NaN counts as the
// maximum); mode 2: entropy > thr (NaN compares false -> 0)
__device__ __forceinline__ int dvq_gate_argmax(const void *gate, int mode, int G, size_t cell, float thr)
{
    if (mode == 2) return (((const float *)gate)[cell] > thr) ? 1 : 0;
    int bi = 0;
    if (mode == 1) {
        const long long *g = (const long long *)gate + cell * G;
        long long best = g[0];
        for (int i = 1; i < G; ++i) {
            const long long v = g[i];
            if (v > best) { best = v; bi = i; }
        }
    } else {
        const float *g = (const float *)gate + cell * G;
        float best = g[0];
        for (int i = 1; i < G; ++i) {
            const float v = g[i];
            if ((v > best) || (v != v && best == best)) { best = v; bi = i; }
        }
    }
    return bi;
}

// dense form: source of output position (y, x) of image b whose cell has grain g
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
    long n;             // top-left output position (b*HWout + y0*Wout + x0); code / mask index
    int rep;
    bool valid;
};

// image of token t, starting the walk at image b_hint (imgstart[b_hint] <= t must hold)
__device__ __forceinline__ int dvq_routed_image(const DvqRouted &rv, int t, int b_hint)
{
    int b = b_hint;
    while (b + 1 < rv.B && t >= rv.imgstart[b + 1]) ++b;
    return b;
}

// last image whose first token is <= t (binary search; t < imgstart[B])
__device__ __forceinline__ int dvq_routed_image_search(const DvqRouted &rv, int t)
{
    int lo = 0, hi = rv.B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rv.imgstart[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// token t of the batch (any value; beyond the end -> invalid, clamped to token 0's addresses)
__device__ __forceinline__ DvqTok dvq_routed_lookup(const DvqRouted &rv, int t, int b_hint)
{
    DvqTok k;
    if (rv.dense) {                                              // rank = output position
        const long total = (long)rv.B * rv.HWout;
        k.valid = t >= 0 && t < total;
        const int tt = k.valid ? t : 0;
        const int b = tt / rv.HWout, pos = tt - b * rv.HWout;
        const int y = pos / rv.Wout, x = pos - y * rv.Wout;
        const int SC = rv.sub[rv.G - 1];
        const int g = (int)rv.indices[(size_t)b * rv.hc * rv.wc + (y / SC) * rv.wc + x / SC];
        k.src = dvq_dense_source(rv, b, y, x, g, k.stride);
        k.n = tt;
        k.rep = 1;
        return k;
    }
    const int total = rv.imgstart[rv.B];
    k.valid = t >= 0 && t < total;
    const int tt = k.valid ? t : 0;
    const int b = dvq_routed_image(rv, tt, k.valid ? b_hint : 0);
    const unsigned e = (total > 0) ? rv.tok[(size_t)b * rv.HWout + (tt - rv.imgstart[b])] : 0u;
    const int g = (int)(e >> 14), sy = (int)((e >> 12) & 3u), sx = (int)((e >> 10) & 3u), cell = (int)(e & 1023u);
    const int cy = cell / rv.wc, cx = cell - cy * rv.wc;
    const int sub = rv.sub[g], rep = rv.rep[g];
    const int gy = cy * sub + sy, gx = cx * sub + sx;             // position in the type's own grid
    const int gwid = rv.wc * sub, plane = rv.hc * sub * gwid;
    k.src = rv.src[g] + (size_t)b * rv.D * plane + (size_t)gy * gwid + gx;
    k.stride = plane;
    k.n = (long)b * rv.HWout + (long)(gy * rep) * rv.Wout + gx * rep;
    k.rep = rep;
    return k;
}

// number of tokens of the batch
__device__ __forceinline__ int dvq_routed_total(const DvqRouted &rv)
{
    return rv.dense ? rv.B * rv.HWout : rv.imgstart[rv.B];
}

// arguments of the low-register pass-1 kernel (vq_assign_routed.hip: vq_pass1_kernel)
struct P1Args {
    const float *z;            // dense source [B, D, HW] (ROUTED = false)
    int HW;
    long N;
    DvqRouted rv;              // routed source (ROUTED = true)
    const char *img;
    const DvqF16Meta *meta;
    const float *E;
    const float *mask;         // [B, HWout] or null
    int K;
    float *zq;                 // [B, D, HWout] or null
    long long *codes;          // [B, HWout]
    double *partials;          // one per workgroup of the launch, or null
    int *counters;
    int *exact_list;
    char *records;
    int rec_cap;               // per shard
    int stagger_ticks;         // > 0: the odd "layers" of the first generation of workgroups start this many
                               // 100-MHz ticks late, so co-resident workgroups alternate HBM and matrix phases
    int stagger_blocks;        // workgroups per layer (= CUs) and first-generation size
    int stagger_first;
    int debug;                 // timing experiments only (DVQ_P1_DEBUG): 1 no second-row store of coarse tokens, 2 no coarse stores, 4 no fine stores
};
