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
// positions per token (rep = SC / sub, SC = sub of the fine type).  The routing prepass
// (routed_prepass_kernel) writes, per image and type, the list of coarse cells of that type in
// row-major order (`cells`) and the unique-token counts; tokens of a cell are consecutive
// (cell-major, row-major inside the cell), so the 32 tokens of a wave sit in one or two rows of coarse
// cells: a few 128-B lines per load instruction.
//
// Virtual token order: images in groups of DVQ_ROUTE_GROUP; inside a group all coarse tokens, then all
// median, then all fine ones (a group's outputs share L2 lines, its segments are processed close
// together in time); every (group, type) segment starts at a multiple of 32 tokens (a "slot" = the 32
// tokens of one wave), so a wave never mixes types: source stride and replication are wave-uniform.
// seg_base[grp*G + g] = first slot of the segment, seg_base[nseg] = total number of slots.
// ---------------------------------------------------------------------------------------------
#define DVQ_ROUTE_GROUP 8
#define DVQ_ROUTE_MAX_CELLS 1024

struct DvqRouted {
    const float *src[3];
    float mval[3];            // codebook_mask value of each type (1 / rep^2)
    int sub[3];               // tokens per coarse-cell edge
    int rep[3];               // output positions per token edge
    int G, B, D, hc, wc;
    int Wout, HWout;          // output grid: SC*wc, (SC*hc)*(SC*wc)
    const int *counts;        // [G][B] unique tokens
    const int *seg_base;      // [nseg + 1] slots
    const unsigned short *cells;   // [G][B][hc*wc]
    int nseg;
};

struct DvqTok {
    const float *src;   // channel 0 of the token
    int stride;         // elements between channels
    long n;             // top-left output position (b*HWout + y0*Wout + x0); code / mask index
    long zq0;           // element offset of channel 0 at that position in z_q [B, D, HWout]
    int rep;
    bool valid;
};

// token `c` of slot `slot` (both any value; slot beyond the end -> invalid).  g_out: the slot's type
// (wave-uniform if slot is).
__device__ __forceinline__ DvqTok dvq_routed_lookup(const DvqRouted &rv, int slot, int c, int &g_out)
{
    DvqTok t;
    const int nslots = rv.seg_base[rv.nseg];
    const bool in_range = slot < nslots;
    const int s = in_range ? slot : (nslots > 0 ? nslots - 1 : 0);
    int lo = 0, hi = rv.nseg;                       // last seg with seg_base[seg] <= s
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rv.seg_base[mid] <= s) lo = mid; else hi = mid;
    }
    const int seg = lo, grp = seg / rv.G, g = seg - grp * rv.G;
    g_out = g;
    int r = 32 * (s - rv.seg_base[seg]) + c;
    int b = grp * DVQ_ROUTE_GROUP;
    const int bend = (b + DVQ_ROUTE_GROUP < rv.B) ? b + DVQ_ROUTE_GROUP : rv.B;
    const int *cnt = rv.counts + (size_t)g * rv.B;
    while (b < bend) {
        const int cb = cnt[b];
        if (r < cb) break;
        r -= cb;
        ++b;
    }
    t.valid = in_range && b < bend && nslots > 0;
    if (b >= bend) { b = bend - 1; r = 0; }
    const int sub = rv.sub[g], rep = rv.rep[g];
    const int per = sub * sub;
    const int k = r / per, w = r - k * per;
    const int ncell = rv.hc * rv.wc;
    int cell = rv.cells[((size_t)g * rv.B + b) * ncell + (t.valid ? k : 0)];
    if (!t.valid) cell = 0;
    const int cy = cell / rv.wc, cx = cell - cy * rv.wc;
    const int sy = w / sub, sx = w - sy * sub;
    const int gy = cy * sub + sy, gx = cx * sub + sx;             // position in the type's own grid
    const int gwid = rv.wc * sub, plane = rv.hc * sub * gwid;
    t.src = rv.src[g] + (size_t)b * rv.D * plane + (size_t)gy * gwid + gx;
    t.stride = plane;
    const long pos = (long)(gy * rep) * rv.Wout + gx * rep;
    t.n = (long)b * rv.HWout + pos;
    t.zq0 = (long)b * rv.D * rv.HWout + pos;
    t.rep = rep;
    return t;
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
};
