// vq_assign_routed.hip -- routing prepass + the low-register pass-1 kernel of the filter path.
//
// Routed assign (dvq_vq_assign_routed_{dual,triple}_f32): the routing tail of the reference encoders
// (EncoderDual.py:134-149, EncoderTriple.py:148-176) and VectorQuantize2.forward
// (quantize2_mask.py:157-191) as ONE op that never materialises h_dual / h_triple.  A coarse cell's
// 2x2 (4x4) output positions are bit-identical copies of one source vector (repeat_interleave), so
// their codes, z_q values and loss terms are identical too: pass 1 scores each UNIQUE token once,
// straight from h_coarse / h_median / h_fine, and writes codes / z_q to every position it covers.
// Per-token arithmetic is the dense kernel's (same fp16 scores, same bound, same resolver, same exact
// chain), hence the same bits.  At fine ratio 0.5 that is 640 instead of 1024 tokens per image, and
// the 2 x 268 MB of h_dual traffic (select writes it, assign reads it back) disappear.
//
//   routed_prepass_kernel   one workgroup per image: argmax of the gate (or entropy > threshold) per
//                           coarse cell -> indices, codebook_mask, the router's int64 gate, and the
//                           image's rank -> token table in row-major order of the tokens' top-left output
//                           positions (dvq_filter.h: DvqRouted); the last workgroup to finish turns the
//                           per-image token counts into the prefix imgstart[]
//   vq_pass1_kernel         pass 1 with NO fp32 copy of z in registers: z is read once for the fp16
//                           fragments (64 VGPRs) and again in the epilogue, where it is still cache
//                           resident (L2 / Infinity Cache) -> <= 128 VGPRs, 3-4 waves per SIMD, so the
//                           HBM phases of some workgroups overlap the matrix-core phase of others.
//                           ROUTED = false is the same kernel on a dense [B, D, HW] tensor.
#include "dvq_filter.h"
#include <type_traits>

// MODE 0: f32 gate logits, 1: int64 gate, 2: f32 entropy map + threshold (route_select.hip has the same)
template <int G, int MODE>
__device__ __forceinline__ int routed_gate_argmax(const void *gate, size_t cell, float thr)
{
    if (MODE == 2) {
        return (((const float *)gate)[cell] > thr) ? 1 : 0;
    } else if (MODE == 1) {
        const long long *g = (const long long *)gate + cell * G;
        long long best = g[0];
        int bi = 0;
#pragma unroll
        for (int i = 1; i < G; ++i) {
            long long v = g[i];
            if (v > best) { best = v; bi = i; }
        }
        return bi;
    } else {
        const float *g = (const float *)gate + cell * G;
        float best = g[0];
        int bi = 0;
#pragma unroll
        for (int i = 1; i < G; ++i) {
            float v = g[i];
            if ((v > best) || (v != v && best == best)) { best = v; bi = i; }
        }
        return bi;
    }
}

template <int G, int MODE>
__global__ __launch_bounds__(256) void routed_prepass_kernel(
    const void *__restrict__ gate, float thr, int B, int hc, int wc,
    long long *__restrict__ indices, float *__restrict__ cmask, long long *__restrict__ gate_out,
    int *__restrict__ imgcount, unsigned short *__restrict__ tok, int *__restrict__ imgstart,
    int *__restrict__ ticket, int dense)
{
    constexpr int SC = (G == 2) ? 2 : 4;
    constexpr int MAXH = SC * DVQ_ROUTE_MAX_CELLS;             // hc <= ncell
    __shared__ unsigned char grain[DVQ_ROUTE_MAX_CELLS];
    __shared__ int rowstart[MAXH + 1];
    __shared__ int scan[256];
    __shared__ int flag;
    const int tid = threadIdx.x;
    const int b = blockIdx.x, ncell = hc * wc;
    const int H = SC * hc, W = SC * wc, W4 = W / 4, HW = H * W;
    auto sub_of = [](int g) { return (g == G - 1) ? SC : (g == 0 ? 1 : 2); };
    // tokens of a cell of type g whose top-left output row is the cell's sub-row ry (0 .. SC-1)
    auto contrib = [&](int g, int ry) { const int sub = sub_of(g), rep = SC / sub; return (ry % rep == 0) ? sub : 0; };

    for (int cell = tid; cell < ncell; cell += 256) {
        const int g = routed_gate_argmax<G, MODE>(gate, (size_t)b * ncell + cell, thr);
        grain[cell] = (unsigned char)g;
        indices[(size_t)b * ncell + cell] = g;
        if (MODE == 2 && gate_out != nullptr) {               // the router's int64 gate, a by-product
            const float e = ((const float *)gate)[(size_t)b * ncell + cell];
            longlong2 gg; gg.x = (e <= thr) ? 1 : 0; gg.y = (e > thr) ? 1 : 0;
            *(longlong2 *)(gate_out + 2 * ((size_t)b * ncell + cell)) = gg;
        }
    }
    __syncthreads();
    for (int i = tid; i < H * W4; i += 256) {                  // codebook_mask [B, 1, H, W]
        const int y = i / W4, x = (i - y * W4) * 4;
        f32x4 m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = grain[(y / SC) * wc + (x + j) / SC];
            m[j] = (G == 2) ? (g == 0 ? 0.25f : 1.0f) : (g == 0 ? 0.0625f : (g == 1 ? 0.25f : 1.0f));
        }
        *(f32x4 *)(cmask + ((size_t)b * H + y) * W + x) = m;
    }
    if (dense == 2) {
        // row-complete de-duplicated form: whole rows of cells are packed greedily into pass-1 workgroups of
        // <= DVQ_RD_MAX_TOKENS unique tokens and <= DVQ_RD_MAX_POS output positions.  A row of cells holds at most
        // SC * W <= 128 tokens (the launcher checks W <= 128 / SC), so no workgroup is empty-handed and an image never
        // needs more than HW / 128 of them (the dense grid); unused slots get rows = 0.
        int *wgd = (int *)tok + (size_t)b * (HW / 128) * 4;
        for (int cy = tid; cy < hc; cy += 256) {
            int cnt = 0;
            for (int cx = 0; cx < wc; ++cx) { const int sub = sub_of(grain[cy * wc + cx]); cnt += sub * sub; }
            rowstart[cy] = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            int j = 0, g0 = 0;
            const int maxwg = HW / 128;
            while (g0 < hc && j < maxwg) {
                int U = 0, ng = 0;
                while (g0 + ng < hc && U + rowstart[g0 + ng] <= DVQ_RD_MAX_TOKENS && (ng + 1) * SC * W <= DVQ_RD_MAX_POS) {
                    U += rowstart[g0 + ng];
                    ++ng;
                }
                wgd[4 * j] = g0; wgd[4 * j + 1] = ng; wgd[4 * j + 2] = U; wgd[4 * j + 3] = 0;
                ++j;
                g0 += ng;
            }
            for (; j < maxwg; ++j) { wgd[4 * j] = 0; wgd[4 * j + 1] = 0; wgd[4 * j + 2] = 0; wgd[4 * j + 3] = 0; }
        }
        return;
    }
    if (dense) return;                                        // the dense form needs no token table
    // tokens per output row (by top-left position), then an exclusive scan over the H rows
    const int per = (H + 255) / 256;
    int mine = 0;
    for (int y = tid * per; y < (tid + 1) * per && y < H; ++y) {
        const int cy = y / SC, ry = y - cy * SC;
        int cnt = 0;
        for (int cx = 0; cx < wc; ++cx) cnt += contrib(grain[cy * wc + cx], ry);
        rowstart[y + 1] = cnt;                                 // count for now
        mine += cnt;
    }
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = (tid >= off) ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    {
        int run = scan[tid] - mine;
        for (int y = tid * per; y < (tid + 1) * per && y < H; ++y) {
            const int cnt = rowstart[y + 1];
            rowstart[y + 1] = run + cnt;                       // inclusive -> start of row y + 1
            run += cnt;
        }
        if (tid == 0) rowstart[0] = 0;
    }
    __syncthreads();
    // rank -> token table
    unsigned short *tb = tok + (size_t)b * HW;
    for (int cell = tid; cell < ncell; cell += 256) {
        const int g = grain[cell], sub = sub_of(g), rep = SC / sub;
        const int cy = cell / wc, cx = cell - cy * wc;
        for (int sy = 0; sy < sub; ++sy) {
            const int ry = sy * rep;
            int pre = 0;
            for (int c2 = 0; c2 < cx; ++c2) pre += contrib(grain[cy * wc + c2], ry);
            const int r0 = rowstart[cy * SC + ry] + pre;
            for (int sx = 0; sx < sub; ++sx) tb[r0 + sx] = dvq_tok_pack(g, sy, sx, cell);
        }
    }
    if (tid == 0)
        __hip_atomic_store(&imgcount[b], rowstart[H], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the workgroup that draws the last ticket turns the counts into the prefix imgstart[0 .. B]
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        flag = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!flag) return;
    __threadfence();
    const int perb = (B + 255) / 256;
    int sum = 0;
    for (int i = tid * perb; i < (tid + 1) * perb && i < B; ++i)
        sum += __hip_atomic_load(&imgcount[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    scan[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = (tid >= off) ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int run = scan[tid] - sum;
    for (int i = tid * perb; i < (tid + 1) * perb && i < B; ++i) {
        imgstart[i] = run;
        run += __hip_atomic_load(&imgcount[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 255) imgstart[B] = scan[255];
}

// ---------------------------------------------------------------------------------------------
// pass 1, low-register form
// ---------------------------------------------------------------------------------------------
template <int REP>
__device__ __forceinline__ void store_rep(float *p, float v, int W)
{
    if (REP == 1) {
        *p = v;
    } else if (REP == 2) {
        const f32x2 vv = {v, v};
        *(f32x2 *)p = vv;
        *(f32x2 *)(p + W) = vv;
    } else {
        const f32x4 vv = {v, v, v, v};
#pragma unroll
        for (int r = 0; r < 4; ++r) *(f32x4 *)(p + (size_t)r * W) = vv;
    }
}

template <int D, int NW, int NBUF, int WPS, bool ROUTED>
__global__ __launch_bounds__(NW * 64, WPS) void vq_pass1_kernel(const P1Args a)
{
    constexpr int S16 = D / 16;
    constexpr int IMG_BYTES = S16 * 1024;
    constexpr int TILE_STRIDE = IMG_BYTES + 256;
    constexpr int CPW = (S16 + NW - 1) / NW;
    constexpr int PER_TILE = CPW + 1;
    constexpr int AHEAD = NBUF - 1;                          // tiles in flight
    static_assert(NBUF >= 2 && NBUF <= 4, "ring of 2..4 slots");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *enraw = (float *)(lds + NBUF * IMG_BYTES);        // [NBUF][NW][64] accumulator seeds, per-wave copy

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int K = a.K;
    const int T = dvq_num_tiles(K);

    // ---- which tokens
    int nblk = (int)gridDim.x;
    int total = 0;
    if (ROUTED) {
        total = dvq_routed_total(a.rv);
        const int nslots = (total + 31) / 32;
        nblk = (nslots + NW - 1) / NW;
        if ((int)blockIdx.x >= nblk) {                       // the grid is sized for the all-fine worst case
            if (a.partials != nullptr && tid == 0) a.partials[blockIdx.x] = 0.0;
            return;
        }
    }
    const float sB = a.meta->scale_b;
    if (a.stagger_ticks > 0 && (int)blockIdx.x < a.stagger_first && (((int)blockIdx.x / a.stagger_blocks) & 1)) {
        // phase offset for every other layer of the first generation of resident workgroups: without it all of
        // them run prologue (HBM) -> code loop (matrix cores) -> epilogue (HBM) in lockstep and the two
        // resources are used one after the other instead of side by side
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }

    auto issue_piece = [&](int t, int q) {
        const int tt = (t < T) ? t : T - 1;                  // past the end: harmless repeat, counts stay constant
        const char *src = a.img + (size_t)tt * TILE_STRIDE;
        if (q < CPW) {
            int chunk = wave * CPW + q;
            if (chunk >= S16) chunk = S16 - 1;
            glds16(src + chunk * 1024 + lane * 16, lds + (t % NBUF) * IMG_BYTES + chunk * 1024);
        } else {
            glds4(src + IMG_BYTES + lane * 4, enraw + ((t % NBUF) * NW + wave) * 64);
        }
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < PER_TILE; ++q) issue_piece(t, q);
    };
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) issue(t);

    const int tile_id = xcd_swizzle(blockIdx.x, nblk);
    const float *zsrc;                                       // channel 8h of this lane's token
    int stride, rep = 1, Wout = 0, HWout;                    // ROUTED: stride and rep are per lane
    int n;                                                   // output position of the token, -1 = no token
    int tokid;                                               // what the exact list carries for it
    if (ROUTED) {
        const int t0 = (tile_id * NW + wave) * 32;           // wave-uniform: first token of the slot
        const int b0 = (t0 < total && !a.rv.dense) ? dvq_routed_image_search(a.rv, t0) : 0;
        tokid = t0 + c;
        const DvqTok tk = dvq_routed_lookup(a.rv, tokid, b0);
        stride = tk.stride;
        rep = tk.rep;
        Wout = a.rv.Wout;
        HWout = a.rv.HWout;
        zsrc = tk.src + (size_t)8 * h * stride;
        n = tk.valid ? (int)tk.n : -1;
    } else {
        const long n_raw = ((long)tile_id * NW + wave) * 32 + c;
        n = (n_raw < a.N) ? (int)n_raw : -1;
        const long q = (n >= 0) ? n : a.N - 1;
        const long bimg = q / a.HW;
        stride = a.HW;
        HWout = a.HW;
        zsrc = a.z + ((size_t)bimg * D + 8 * h) * a.HW + (size_t)(q - bimg * a.HW);
        tokid = n;
    }

    // ---- prologue: z in batches of four k-steps -> fp16 fragments, exact-order norm, bound
    f16x8 zh[S16];
    float xn, thr2W;
    {
        float pa[2][8];
        float amax = 0.0f, zeta2 = 0.0f;
        const float *zpb = zsrc;
        constexpr int BATCH = (S16 < 4) ? S16 : 4;
#pragma unroll
        for (int sb = 0; sb < S16; sb += BATCH) {
            float zf[BATCH][8];
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int q = 0; q < BATCH; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) zf[q][j] = zpb[(size_t)(16 * q + j) * stride];
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int s = sb + q;
                u32x4 packed;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const float v0 = zf[q][2 * j2], v1 = zf[q][2 * j2 + 1];
                    const float q0 = sq_rn(v0), q1 = sq_rn(v1);
                    pa[s & 1][2 * j2] = (s < 2) ? q0 : __fadd_rn(pa[s & 1][2 * j2], q0);
                    pa[s & 1][2 * j2 + 1] = (s < 2) ? q1 : __fadd_rn(pa[s & 1][2 * j2 + 1], q1);
                    amax = vmax_abs(amax, v0);
                    amax = vmax_abs(amax, v1);
                    f32x2 vv = {v0, v1};
                    f16x2 hh = __builtin_convertvector(vv, f16x2);
                    packed[j2] = __builtin_bit_cast(unsigned, hh);
                    const float r0 = v0 - (float)hh[0], r1 = v1 - (float)hh[1];     // exact
                    zeta2 = __builtin_fmaf(r0, r0, zeta2);
                    zeta2 = __builtin_fmaf(r1, r1, zeta2);
                }
                zh[s] = __builtin_bit_cast(f16x8, packed);
            }
            // one batch of loads at a time (register budget): the next batch's addresses depend,
            // opaquely, on this batch's last converted fragment
            zpb += (size_t)16 * BATCH * stride;
            asm volatile("" : "+v"(zpb) : "v"(zh[sb + BATCH - 1]));
        }
        float t8[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float o0 = __shfl_xor(pa[0][l], 32), o1 = __shfl_xor(pa[1][l], 32);
            float a0_ = h == 0 ? pa[0][l] : o0;
            float a1_ = h == 0 ? o0 : pa[0][l];
            float a2_ = h == 0 ? pa[1][l] : o1;
            float a3_ = h == 0 ? o1 : pa[1][l];
            t8[l] = __fadd_rn(__fadd_rn(__fadd_rn(a0_, a1_), a2_), a3_);
        }
        xn = t8[0];
#pragma unroll
        for (int l = 1; l < 8; ++l) xn = __fadd_rn(xn, t8[l]);
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        zeta2 += __shfl_xor(zeta2, 32);
        thr2W = dvq_filter_threshold(xn, amax, zeta2, sB, a.meta);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the first AHEAD tiles (own DMA) landed during the prologue

    // ---- code loop
    float m1 = -__builtin_inff(), m2 = -__builtin_inff();
    int t1 = 0;
    for (int t = 0; t < T; ++t) {
        f32x16 acc;
        auto read_seeds = [&]() {
            const float *seeds = enraw + ((t % NBUF) * NW + wave) * 64 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 e4 = *(const f32x4 *)(seeds + 8 * g);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * g + q] = e4[q];
            }
        };
        if (NBUF >= 3) {
            // seeds of tile t: this wave's own DMA copy, landed one step ago -> read before the barrier
            read_seeds();
            if (t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 3) * PER_TILE) : "memory");
            __builtin_amdgcn_s_barrier();                    // tile t (everybody's DMA) landed; t-1 consumed
        } else {
            if (t > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_seeds();
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        if (S16 != 16) issue(t + AHEAD);                     // D = 256: pieces ride between the MFMAs below
        const unsigned tile_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(
                                    lds + (t % NBUF) * IMG_BYTES + lane * 16);
        f16x8 a0, a1, a2, a3;
        asm volatile("" : "+v"(acc));
        __builtin_amdgcn_sched_barrier(0);
#define DVQ_RD(dst, S) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(tile_a), "i"((S) * 1024))
#define DVQ_MM(src, S, WAIT, Q)                                                         \
        asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" ::: "memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                              \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(src, zh[S], acc, 0, 0, 0);         \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if ((S) + 4 < S16) { DVQ_RD(src, ((S) + 4 < S16 ? (S) + 4 : 0)); }              \
        if ((Q) >= 0 && (Q) < PER_TILE) issue_piece(t + AHEAD, (Q));
        DVQ_RD(a0, 0); DVQ_RD(a1, 1); DVQ_RD(a2, 2); DVQ_RD(a3, 3);
        __builtin_amdgcn_s_setprio(1);
        if (S16 == 16) {
            DVQ_MM(a0, 0, 3, -1) DVQ_MM(a1, 1, 3, 0) DVQ_MM(a2, 2, 3, -1) DVQ_MM(a3, 3, 3, -1)
            DVQ_MM(a0, 4, 3, 1) DVQ_MM(a1, 5, 3, -1) DVQ_MM(a2, 6, 3, -1) DVQ_MM(a3, 7, 3, 2)
            DVQ_MM(a0, 8, 3, -1) DVQ_MM(a1, 9, 3, -1) DVQ_MM(a2, 10, 3, 3) DVQ_MM(a3, 11, 3, -1)
            DVQ_MM(a0, 12, 3, -1) DVQ_MM(a1, 13, 2, 4) DVQ_MM(a2, 14, 1, -1) DVQ_MM(a3, 15, 0, -1)
        } else if (S16 == 8) {
            DVQ_MM(a0, 0, 3, -1) DVQ_MM(a1, 1, 3, -1) DVQ_MM(a2, 2, 3, -1) DVQ_MM(a3, 3, 3, -1)
            DVQ_MM(a0, 4, 3, -1) DVQ_MM(a1, 5, 2, -1) DVQ_MM(a2, 6, 1, -1) DVQ_MM(a3, 7, 0, -1)
        } else {
            DVQ_MM(a0, 0, 3, -1) DVQ_MM(a1, 1, 2, -1) DVQ_MM(a2, 2, 1, -1) DVQ_MM(a3, 3, 0, -1)
        }
#undef DVQ_MM
#undef DVQ_RD
        __builtin_amdgcn_s_setprio(0);
        const float om = m1;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float g0 = __uint_as_float((__float_as_uint(acc[r]) & 0xFFFFFFF0u) | (unsigned)r);
            float g1 = __uint_as_float((__float_as_uint(acc[r + 1]) & 0xFFFFFFF0u) | (unsigned)(r + 1));
            float md = __builtin_amdgcn_fmed3f(m1, g0, g1);
            m1 = vmax3_raw(m1, g0, g1);
            m2 = vmax_raw(m2, md);
        }
        t1 = (m1 != om) ? t : t1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // surplus ring DMA

    // ---- decision, one queue atomic per wave
    int code;
    float thr;
    bool undecided, hopeless;
    const bool valid = n >= 0;
    {
        const float o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
        const int ot = __shfl_xor(t1, 32);
        const bool other_wins = (o1 > m1) || (o1 == m1 && h == 1);
        const float best = other_wins ? o1 : m1;
        const float second = fmaxf(other_wins ? m1 : o1, fmaxf(m2, o2));
        const int wt = other_wins ? ot : t1;
        const int wh = other_wins ? (h ^ 1) : h;
        const int r = (int)(__float_as_uint(best) & 15u);
        code = wt * 32 + (r & 3) + 8 * (r >> 2) + 4 * wh;
        thr = best - thr2W;
        const bool final_ok = (best - second) > thr2W;
        hopeless = !(code < K) || !(thr == thr);
        undecided = valid && !hopeless && !final_ok;
    }
    const unsigned long long umask = __ballot(undecided && h == 0);
    const int shard = blockIdx.x & (DVQ_QSHARDS - 1);
    int slot = -1;
    if (umask != 0ull) {                                    // wave-uniform
        int slot_raw = 0;
        if (lane == 0) slot_raw = atomicAdd(&a.counters[DVQ_QCOUNT0 + shard], (int)__popcll(umask));
        const int base = __shfl(slot_raw, 0);
        slot = undecided ? base + (int)__popcll(umask & ((1ull << c) - 1ull)) : -1;
        if (slot >= a.rec_cap) { hopeless = true; slot = -1; }          // shard full -> exact list
    }
    if (valid && hopeless && h == 0) {
        int pos = atomicAdd(&a.counters[1], 1);
        a.exact_list[pos] = tokid;
    }

    // ---- epilogue: z again (cache resident), chosen codebook row, z_q, loss term, record of a queued token
    float lsum = 0.0f;
    if (valid && !hopeless) {
        if (h == 0) {
#pragma unroll 1
            for (int ry = 0; ry < rep; ++ry)
#pragma unroll 1
                for (int rx = 0; rx < rep; ++rx) a.codes[(size_t)n + (size_t)ry * Wout + rx] = (long long)code;
        }
        if (a.zq != nullptr || a.partials != nullptr || slot >= 0) {
            const float *ep = a.E + (size_t)code * D + 8 * h;
            const float m = (a.mask != nullptr) ? a.mask[n] : 1.0f;
            char *rec = (slot >= 0) ? a.records + ((size_t)shard * a.rec_cap + slot) * rec_bytes(D) : nullptr;
            const long bimg = n / HWout;
            const size_t zq0 = ((size_t)bimg * D + 8 * h) * HWout + (size_t)(n - bimg * HWout);
            auto finish = [&](auto store_tag) {
                constexpr bool STORE = decltype(store_tag)::value;
                float *zqp = STORE ? a.zq + zq0 : nullptr;
                const float *zpe = zsrc, *epe = ep;          // advance by two k-steps per batch
                constexpr int EB = (S16 < 2) ? S16 : 2;
#pragma unroll
                for (int s0 = 0; s0 < S16; s0 += EB) {
                    float zf[EB][8];
                    f32x4 eg[EB][2];
#pragma unroll
                    for (int q = 0; q < EB; ++q) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) zf[q][j] = zpe[(size_t)(16 * q + j) * stride];
                        eg[q][0] = *(const f32x4 *)(epe + 16 * q);
                        eg[q][1] = *(const f32x4 *)(epe + 16 * q + 4);
                    }
                    float v[EB][8];
#pragma unroll
                    for (int q = 0; q < EB; ++q) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float e = eg[q][j >> 2][j & 3];
                            float diff = __fsub_rn(e, zf[q][j]);
                            v[q][j] = __fadd_rn(zf[q][j], diff);
                            lsum = __fadd_rn(lsum, __fmul_rn(__fmul_rn(diff, diff), m));
                        }
                    }
                    if (STORE) {
                        // one branch per batch on the lane's replication (a wave may mix grain types)
                        auto put = [&](auto rep_tag) {
                            constexpr int REP = decltype(rep_tag)::value;
#pragma unroll
                            for (int q = 0; q < EB; ++q)
#pragma unroll
                                for (int j = 0; j < 8; ++j)
                                    store_rep<REP>(zqp + (size_t)(16 * (s0 + q) + j) * HWout, v[q][j], Wout);
                        };
                        if (!ROUTED || rep == 1) put(std::integral_constant<int, 1>{});
                        else if (rep == 2) put(std::integral_constant<int, 2>{});
                        else put(std::integral_constant<int, 4>{});
                    }
                    if (rec != nullptr) {
#pragma unroll
                        for (int q = 0; q < EB; ++q) {
                            const int s = s0 + q;
                            f32x4 lo = {zf[q][0], zf[q][1], zf[q][2], zf[q][3]};
                            f32x4 hi = {zf[q][4], zf[q][5], zf[q][6], zf[q][7]};
                            *(f32x4 *)(rec + (16 * s + 8 * h) * 4) = lo;
                            *(f32x4 *)(rec + (16 * s + 8 * h + 4) * 4) = hi;
                        }
                    }
                    zpe += (size_t)16 * EB * stride;
                    epe += 16 * EB;
                    asm volatile("" : "+v"(zpe), "+v"(epe) : "v"(lsum));     // next batch's loads wait for this one
                }
            };
            if (a.zq != nullptr) finish(std::true_type{});
            else finish(std::false_type{});
            lsum *= (float)(rep * rep);                      // every covered position carries the same term
            if (rec != nullptr && h == 0) {
                RecMeta rm;
                rm.n = n; rm.xn = xn; rm.thr = thr; rm.tokid = tokid; rm.prov = code;
                rm.best = ~0ull; rm.rep = rep;
                *(RecMeta *)(rec + (size_t)D * 4) = rm;
            }
        }
    }
    if (a.partials != nullptr) {
        double dsum = (double)lsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        __syncthreads();
        double *red = (double *)lds;
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w];
            a.partials[blockIdx.x] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int D, int NW, int NBUF, int WPS, bool ROUTED>
static int launch_p1(const P1Args &a, int nblocks, hipStream_t st)
{
    static unsigned long long done = 0;
    const size_t shmem = (size_t)NBUF * (D / 16) * 1024 + (size_t)NBUF * NW * 64 * sizeof(float);
    int rc = dvq_allow_dynamic_lds((const void *)vq_pass1_kernel<D, NW, NBUF, WPS, ROUTED>, (int)shmem, &done);
    if (rc) return rc;
    hipLaunchKernelGGL((vq_pass1_kernel<D, NW, NBUF, WPS, ROUTED>), dim3(nblocks), dim3(NW * 64), shmem, st, a);
    return (int)hipGetLastError();
}

// variant: 0 = 4 waves, 2-slot ring, 4 workgroups / CU (<= 128 VGPRs)
//          1 = 4 waves, 3-slot ring, 3 workgroups / CU (<= 168 VGPRs)
//          2 = 8 waves, 4-slot ring, 2 workgroups / CU (<= 128 VGPRs, one codebook stream per 256 tokens)
//          3 = 8 waves, 3-slot ring, 2 workgroups / CU
// -> tokens per workgroup
int dvq_pass1_tokens_per_block(int variant) { return (variant >= 2) ? 256 : 128; }

template <int D, bool ROUTED>
static int launch_p1_variant(int variant, const P1Args &a, int nblocks, hipStream_t st)
{
    switch (variant) {
    case 0: return launch_p1<D, 4, 2, 4, ROUTED>(a, nblocks, st);
    case 1: return launch_p1<D, 4, 3, 3, ROUTED>(a, nblocks, st);
    case 2: return launch_p1<D, 8, 4, 4, ROUTED>(a, nblocks, st);
    default: return launch_p1<D, 8, 3, 4, ROUTED>(a, nblocks, st);
    }
}

int dvq_launch_pass1_lowreg(int D, bool routed, int variant, const P1Args &a, int nblocks, hipStream_t st)
{
    switch (D) {
    case 64:  return routed ? launch_p1_variant<64, true>(variant, a, nblocks, st) : launch_p1_variant<64, false>(variant, a, nblocks, st);
    case 128: return routed ? launch_p1_variant<128, true>(variant, a, nblocks, st) : launch_p1_variant<128, false>(variant, a, nblocks, st);
    case 256: return routed ? launch_p1_variant<256, true>(variant, a, nblocks, st) : launch_p1_variant<256, false>(variant, a, nblocks, st);
    default:  return -1000;
    }
}

int dvq_launch_routed_prepass(int G, int gate_mode, const void *gate, float thr, int B, int hc, int wc,
                              long long *indices, float *cmask, long long *gate_out, int *imgcount,
                              unsigned short *tok, int *imgstart, int *ticket, int dense, hipStream_t st)
{
#define DVQ_PRE(GG, MM) hipLaunchKernelGGL((routed_prepass_kernel<GG, MM>), dim3(B), dim3(256), 0, st, gate, thr, B, hc, wc, indices, cmask, gate_out, imgcount, tok, imgstart, ticket, dense)
    if (G == 2 && gate_mode == 2) DVQ_PRE(2, 2);
    else if (G == 2 && gate_mode == 1) DVQ_PRE(2, 1);
    else if (G == 2) DVQ_PRE(2, 0);
    else if (gate_mode == 1) DVQ_PRE(3, 1);
    else DVQ_PRE(3, 0);
#undef DVQ_PRE
    return (int)hipGetLastError();
}
