// vq_assign_filter.hip -- nearest-codebook assignment at matrix-core speed with a proof obligation.
//
// Same contract as vq_assign_exact.hip (output identical bit for bit), reached in three steps:
//
//  pass 1   vq_assign_filter_kernel: approximate scores on the fp16 matrix cores
//             G_j = (zh . eh_j) - 2^(a+b-1) en_j   ~   -2^(a+b-1) (d_j - xn)
//           zh = fp16(2^a z), eh = fp16(2^b e), fp32 accumulate, en_j the exact reference norm
//           pre-loaded as the MFMA accumulator.  Per token a running top-2 and a RIGOROUS bound W
//           on |G_j - truth| that also covers the reference's own fp32 rounding.  The best code is
//           written for every token (codes, z_q, loss term -- z is read from HBM exactly once and
//           kept in registers in fp32 for z_q).  If best - second > 2W no other code can win in
//           the reference arithmetic either and the token is final.  Otherwise its operands are
//           dumped to a compact record and it is queued for the resolver.
//  resolve  vq_resolve_kernel (queued tokens only, a few %): re-runs the fp16 scores from the dumped
//           fragments, collects every code within 2W of the best, evaluates those few with the
//           bit-exact sequential fp32 FMA chain and the reference's d = fl(fl(xn+en) - 2 dot),
//           takes the first-index minimum, and rewrites codes / z_q / loss term if the winner
//           differs from pass 1's provisional choice.
//  exact    vq_assign_exact_kernel over a second list: NaN/Inf tokens, tokens fp16 cannot scale,
//           record or candidate overflow (normally empty; the kernel exits at once).
//
// Error budget (real-number analysis; zeta = 2^a z - zh and eta_j = 2^b e_j - eh_j are the ACTUAL
// rounding residuals, their 2-norms are computed, so fp16 subnormals need no special case):
//   |2^(a+b) z.e_j - zh.eh_j| <= ||zeta|| ||eh_j|| + ||zh|| ||eta_j|| + ||zeta|| ||eta_j||     (Cauchy-Schwarz)
//   MFMA fp32 accumulation        <= gamma' (||zh|| ||eh_j|| + |seed|),  gamma' = 2^-13  (>= 4x the
//                                    worst case of 272 roundings of 2^-23)
//   4 mantissa bits replaced by the accumulator-register index           <= 2^-19 |G|
//   reference side, d = fl(fl(xn+en) - 2 dotc), dotc the D-term fp32 chain:
//                                 <= 2^(a+b) [u(1+u)(xn+en) + (u + gamma_D)(1+gamma_D) ||z|| ||e_j||]
// W is the sum with ||e_j||, en_j, ||eta_j|| replaced by their maxima over the codebook.
#include "dvq_filter.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// z is read once and z_q written once per launch: stream them past L2 (nt) so that the codebook
// image and the fp32 codebook rows keep their lines
#define DVQ_LOAD_Z(p) __builtin_nontemporal_load(p)
#define DVQ_STORE_ZQ(p, v) __builtin_nontemporal_store((v), (p))
// the same through buffer instructions (resource = wave-uniform base, vector byte offset, scalar byte offset; aux 2 = nt)
#define DVQ_BUF_LOAD(rsrc, voff, soff, AUX) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rsrc), (voff), (soff), (AUX)))
#define DVQ_BUF_STORE(v, rsrc, voff, soff) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(v)), (rsrc), (voff), (soff), 2)
// the per-lane select prologue (SEL = 1) reads lines of the coarser branches that neighbouring waves read again:
// plain loads keep them in L2
#define DVQ_LOAD_SEL(p) (*(p))
#ifndef DVQ_WIDE_MIN_K
#define DVQ_WIDE_MIN_K 2048      // codebook size from which pass 1 takes the two-blocks-per-wave form (whole op at B = 256: -2 % at 1024, +8 % at 2048, +10 % at 4096 and 16384)
#endif
// (Round 3's timing-only ablation switches, the per-CU anti-phase lock, the early-DMA variant and the per-workgroup clock stamps
// left this file in round 4: their results are in profiles/archive/r03_pass1_*.json and DESIGN.md section 5.1, the code in git history
// up to commit "Feature-router gate as a tiled GEMM".)

// ---------------------------------------------------------------------------------------------
// prep: meta (scale, norm maxima, finiteness), fp16 tile images, rounding-residual norm
//   image of tile t: [s < D/16][lane < 64][j < 8] halves = fp16(2^b E[32t + (lane&31)][16s + 8(lane>>5) + j])
//   -> the A fragment of k-step s is ONE ds_read_b128 at s*1024 + lane*16 (lane-linear, conflict-free)
// ---------------------------------------------------------------------------------------------
// Round 6: ONE kernel behind the f32 prep (vq_assign_exact.hip: codebook_prep_f32_kernel), which leaves per-workgroup partial
// maxima in the padding of its tiles -- until now six launches (partial scan, scan, two image kernels, residual norms: 42 us of
// every training step for 1 MiB of codebook).  A workgroup owns CPW codes of a tile, as the f32 prep does: every workgroup
// reduces the partials to the meta values itself (a few KiB from L2; maxima and an OR: any order gives the same bits) and workgroup 0
// writes them; a thread converts octets of channels and stores them into BOTH images (the 32x32x16 order and the 16x16x32 order
// hold the same fp16 values); the rounding residuals go through LDS so that a code's squared residual norm is summed in the order
// it always was (lane-strided, xor tree); etamax was zeroed by the f32 prep.
int dvq_prep_codes_per_workgroup(int K);                 // vq_assign_exact.hip
template <int CPW>
__global__ __launch_bounds__(256) void codebook_prep_f16_kernel(const float *__restrict__ E, int K, int D,
                                                                const float *__restrict__ tiles32,
                                                                const float *__restrict__ en_all,
                                                                DvqF16Meta *__restrict__ meta, char *__restrict__ img,
                                                                char *__restrict__ img16)
{
    extern __shared__ float r2[];                            // [CPW][D] squared rounding residuals
    __shared__ float s_a[4], s_e[4];
    __shared__ int s_b[4];
    constexpr int SUBS = 32 / CPW;
    const int t = blockIdx.x / SUBS, sub = blockIdx.x % SUBS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = dvq_num_tiles(K);
    float amax = 0.0f, enmax = 0.0f;
    int bad = 0;
    {
        const size_t tf = dvq_tile_floats(D);
        for (int i = tid; i < T * SUBS; i += 256) {
            const f32x4 p = *(const f32x4 *)(tiles32 + (size_t)(i / SUBS) * tf + 32 * D + 32 + 4 * (i % SUBS));
            amax = fmaxf(amax, p[0]);
            enmax = fmaxf(enmax, p[1]);
            bad |= p[2] != 0.0f;
        }
        for (int off = 32; off > 0; off >>= 1) {
            amax = fmaxf(amax, __shfl_xor(amax, off));
            enmax = fmaxf(enmax, __shfl_xor(enmax, off));
            bad |= __shfl_xor(bad, off);
        }
        if (lane == 0) { s_a[wave] = amax; s_e[wave] = enmax; s_b[wave] = bad; }
        __syncthreads();
        amax = fmaxf(fmaxf(s_a[0], s_a[1]), fmaxf(s_a[2], s_a[3]));
        enmax = fmaxf(fmaxf(s_e[0], s_e[1]), fmaxf(s_e[2], s_e[3]));
        bad = s_b[0] | s_b[1] | s_b[2] | s_b[3];
    }
    int bexp = 0;
    if (amax > 0.0f) {
        int e;
        (void)frexpf(amax, &e);         // amax = m 2^e, m in [0.5, 1)
        bexp = 15 - e;                  // 2^b amax in [2^14, 2^15)
    }
    if (bexp > 100 || bexp < -100) bad = 1;
    const float sb = ldexpf(1.0f, bad ? 0 : bexp);
    if (blockIdx.x == 0 && tid == 0) {
        meta->ok = bad ? 0 : 1;
        meta->b_exp = bexp;
        meta->scale_b = sb;
        meta->emax = sqrtf(enmax) * 1.00001f;
        meta->enmax = enmax;
    }
    // image of tile t = [fp16 image: D/16 x 1 KiB][tail 256 B: seed[32] = -2^(b-1) en_j, the MFMA accumulator
    // start value that turns the dot product into the score; codes >= K get a huge negative FINITE
    // seed (they never win, and packing the register index into the low mantissa bits cannot turn
    // them into NaNs as it would for -inf); pad[32]]
    //   32x32x16 order:  [s < D/16][lane < 64][j < 8] = fp16(2^b E[32t + (lane & 31)][16 s + 8 (lane >> 5) + j])
    //   16x16x32 order (image "16"): fragment F = c2 * (D/32) + s' (c2 < 2 code halves, s' < D/32 k-steps of 32), lane l, j < 8:
    //                    fp16(2^b E[32t + 16 c2 + (l & 15)][32 s' + 8 (l >> 4) + j]); same seeds tail.
    const int KG = D / 8, S32 = D / 32;
    const size_t tile_bytes = (size_t)D * 64 + 256;
    char *t8 = img + (size_t)t * tile_bytes, *t16 = img16 + (size_t)t * tile_bytes;
    for (int u = tid; u < CPW * KG; u += 256) {
        const int cl = u / KG, o = u - cl * KG, c = sub * CPW + cl;
        const int code = t * 32 + c;
        f16x8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (code < K) ? E[(size_t)code * D + 8 * o + j] * sb : 0.0f;
            const _Float16 hh = (_Float16)v;                 // round to nearest even
            hv[j] = hh;
            const float r = v - (float)hh;
            r2[cl * D + 8 * o + j] = r * r;
        }
        *(f16x8 *)(t8 + (((o >> 1) * 64 + 32 * (o & 1) + c) * 16)) = hv;
        *(f16x8 *)(t16 + ((((c >> 4) * S32 + (o >> 2)) * 64 + 16 * (o & 3) + (c & 15)) * 16)) = hv;
    }
    if (sub == 0 && tid < 64) {
        const int code = t * 32 + tid;
        float v = 0.0f;
        if (tid < 32) v = (code < K) ? fmaxf(-0.5f * sb * en_all[code], DVQ_SEED_PAD) : DVQ_SEED_PAD;
        ((float *)(t8 + (size_t)D * 64))[tid] = v;
        ((float *)(t16 + (size_t)D * 64))[tid] = v;
    }
    __syncthreads();
    // etamax = max_j || 2^b e_j - fp16(2^b e_j) ||_2 (each residual is exact in fp32), rounded up.  A wave per code.
    float best = 0.0f;
    for (int cl = wave; cl < CPW; cl += 4) {
        float sum = 0.0f;
        for (int k = lane; k < D; k += 64) sum += r2[cl * D + k];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        best = fmaxf(best, sum);
    }
    if (lane == 0 && best > 0.0f) {
        float v = sqrtf(best) * 1.001f;     // (the summation order differs from a sequential sum by a few ulp: inside the 0.1 % margin)
        atomicMax((int *)&meta->etamax, __float_as_int(v));       // positive floats order as ints
    }
}

// ---------------------------------------------------------------------------------------------
// pass 1: 4-wave workgroups of 128 consecutive tokens, TWO per CU (<= 256 VGPRs).  A wave keeps its
// 32 tokens twice in registers -- fp32 (D/2 VGPRs, read once, reused for z_q and the resolver
// record: z is never re-read, HBM traffic = the algorithmic bytes) and fp16 MFMA fragments (D/4).
// The code loop runs on v_mfma_f32_16x16x32_f16 (tile image "16" of the prep buffer): the latents are
// converted in the load layout (lane = token, 8 consecutive channels) and permuted into the B-operand
// order through a 2-KiB per-wave LDS scratch; every A fragment (16 codes x 32 k) feeds two MFMAs.
//
// SEL: 0 = dense z.
//      1 = the router select fused in (DvqRouted, dense view): token n is output position n, its source
//          vector sits in the encoder branch that won its cell (per-lane source pointer and channel stride).
//      2 = the same for a 32-wide output grid (every reference config): the workgroup's four output rows
//          need exactly ONE 128-B line per channel of the 2x-coarser branch (dual: coarse; triple: median)
//          and one 32-B piece of the 4x-coarser one (triple: coarse).  Those are DMA'd ONCE per workgroup
//          into ring slots the code loop does not need yet and read back with ds_read_b32; the fine branch
//          is read by every lane with the dense kernel's load pattern.  With the per-lane form (SEL = 1) the
//          two / four waves that share a coarse line each fetched it (PMC: 2.1x the coarse bytes).
//
// CONV: the model's 1x1 quant_conv (qconv.hip) runs as the PROLOGUE: instead of loading its latents a wave computes them,
//       h = W x + bias for its 32 tokens, on v_mfma_f32_32x32x16_f16 at fp32 grade (x = hi + lo per token, W = hi + lo,
//       hi*hi + hi*lo + lo*hi; qconv.hip's arithmetic), streaming x in k-steps of 16 input channels (8 loads per lane, three
//       k-steps in flight) and the weight images through the code ring's four slots (16 KiB per k-step: 8 row tiles x hi / lo).
//       The rows of a weight tile are permuted (qconv_row_channel) so that the 128 accumulator registers of a lane ARE
//       zf[s][j] in the layout the rest of the kernel expects; h never goes to memory (except the rows of tokens handed to
//       the exact-list kernel, which reads them from cv.h_buf).  The per-token power-of-two scale of x follows the running
//       maximum: when a k-step brings a value that would leave the fp16 range the accumulators are rescaled (exact, a
//       workgroup-rare event), so no second pass over x is needed.
// ---------------------------------------------------------------------------------------------
#ifdef DVQ_TUNING
// diagnostic of the tuning build only: per-token (best, second, 2W, code) of the production arithmetic for the bound audit
// (tools/bound_audit.py --production).  Written to a buffer of its own; no output value is computed from it.
__device__ float *g_dvq_tokdbg = nullptr;                  // [N][4]
// ... and stage stamps of the split form's workgroups (100-MHz wall clock): [workgroup][8] (tools/archive/split_timeline.py)
__device__ unsigned long long *g_dvq_stamps = nullptr;
#define DVQ_STAMP(i) do { if (SPLIT && g_dvq_stamps != nullptr && threadIdx.x == 0) g_dvq_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
// ... and of the resolver's workgroups, behind those: [4096 + workgroup][8]
#define DVQ_RSTAMP(i) do { if (g_dvq_stamps != nullptr && threadIdx.x == 0 && blockIdx.y == 0) g_dvq_stamps[(size_t)(4096 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define DVQ_STAMP(i) do { } while (0)
#define DVQ_RSTAMP(i) do { } while (0)
#endif

// (Round 5 built the resolver INTO this launch -- consumer workgroups appended to the grid, records handed over with sc1
// write-through stores / stamps / sc1 loads, decisions through a rewrite list -- bit-exact and slower: the consumers get slots only
// when the last generation of token blocks retires, and what they do beside those blocks costs the blocks as much as it would cost
// afterwards: profiles/r05_fused_consumers_negative.json; the code is in git history, commit "Fused form of the filter path".)
// NT: the latents are read with the non-temporal hint (a launch streams more than the 256-MB memory-side cache holds: keep L2 for
// the code image and the codebook rows) or with plain loads (vq_assign_filter_cached_kernel: a batch whose features FIT that cache
// was just written by the encoder / read by the router gate, and plain loads are served from it: -6 % on the configs[3] per-GPU
// step, profiles/archive/r04_cache_policy.json)
// FLAT: the latents are ROW-MAJOR [N, D] (a token's channels contiguous: quantize2_list.py:153-170, channel_last inputs,
// VQEmbedding.forward) -- the same tensor as [B = N, D, HW = 1], but read and written as what it is: a lane's 8 channels of a
// k-step are 32 contiguous bytes = two 16-byte accesses (32 loads and 32 stores per lane instead of 128 each; with lane = token
// and 4-byte accesses at a stride of D * 4 bytes every wave-instruction touched 64 lines for 256 useful bytes).
// SPLIT (small batches: fewer token blocks than CUs; vq_assign_filter_split_kernel): `ksplit` workgroups share a token block, each
// scores it against its own slice of the code tiles -- a lone workgroup's code loop is an issue-bound ~1330 cycles per tile whoever
// else is on the chip, 20 of the 27 us the kernel takes for BASELINE configs[0] (1024 tokens on 8 of 256 CUs) -- and leaves
// (best, second, code) per token in `split`; the workgroup that takes a block's last ticket merges them (lower slice wins ties, as
// the lower tile does in the loop) and runs the epilogue of the whole block.  Everything downstream sees what one workgroup
// would have produced, up to which of two equal scores is called best (tokens that close are undecided either way).
template <int D, int SEL, bool CONV, bool FOLD, bool NT, bool FLAT = false, bool SPLIT = false>
__device__ __forceinline__ void pass1_body(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted &rv, const DvqConv &cv,
    f32x4 *__restrict__ split = nullptr, int ksplit = 1)
{
    static_assert(!SPLIT || SEL != 2, "the split form: dense or per-lane select; plain, with the conv prologue, or on the folded codebook");
    static_assert(!(SPLIT && FLAT && SEL != 0), "row-major latents are a dense op");
    static_assert(!CONV || (D == 256 && SEL != 2), "the conv prologue exists for D = 256, dense or per-lane select");
    static_assert(!(CONV && FOLD), "the conv is either computed (CONV) or folded into the code image (FOLD)");
    static_assert(!FLAT || (SEL == 0 && !CONV), "the row-major form is a dense op");
    constexpr int NW = 4;
    constexpr int S16 = D / 16;
    constexpr int S32 = S16 / 2;
    constexpr int IMG_BYTES = S16 * 1024;
    constexpr int TILE_STRIDE = IMG_BYTES + 256;
    constexpr int CPW = (S16 + NW - 1) / NW;
    static_assert(CPW * NW == S16 && CPW <= 4, "a wave's chunks of a code tile are contiguous and within the instruction offset");
    constexpr int PER_TILE = CPW + 1;
    constexpr int NBUF = 4;
    // FLAT: the per-wave transposition image of half a row per token (see the prologue)
    constexpr int FLAT_RSH = D * 2 + 16;                     // bytes per token in the image
    constexpr int FLAT_TRW = 32 * FLAT_RSH;                  // bytes per wave
    constexpr int FLAT_LPT = D * 2 / 16;                     // lanes (16-byte pieces) per token-half
    constexpr int FLAT_TPI = 64 / FLAT_LPT;                  // tokens per wave-instruction
    constexpr int FLAT_IPH = 32 / FLAT_TPI;                  // wave-instructions per half
    static_assert(!FLAT || NW * FLAT_TRW <= NBUF * IMG_BYTES + NBUF * NW * 64 * 4 + NW * 2048, "the images fit the kernel's LDS");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *enraw = (float *)(lds + NBUF * IMG_BYTES);        // [NBUF][NW][64] accumulator seeds, per-wave copy
    DVQ_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    // SPLIT: this workgroup's slice of the code tiles [t_lo, t_lo + T) -- the loop below runs on slice-relative tile numbers
    const int ks = SPLIT ? (int)(blockIdx.x % (unsigned)ksplit) : 0;
    const int t_lo = SPLIT ? (int)((long)dvq_num_tiles(K) * ks / ksplit) : 0;
    const int T = SPLIT ? (int)((long)dvq_num_tiles(K) * (ks + 1) / ksplit) - t_lo : dvq_num_tiles(K);
    if constexpr (SPLIT) img += (size_t)t_lo * TILE_STRIDE;
    const float sB = meta->scale_b;
    char *scr = lds + NBUF * IMG_BYTES + NBUF * NW * 64 * 4 + wave * 2048;   // this wave's permutation scratch

    // DMA of code tile t into its ring slot, in PER_TILE pieces (q < CPW: 1 KiB of the image, q == CPW:
    // this wave's copy of the seeds).  Past the end: harmless repeat, so the counts stay constant.
    auto issue_piece = [&](int t, int q) {
        const int tt = (t < T) ? t : T - 1;
        const char *src = img + (size_t)tt * TILE_STRIDE;
        if (q < CPW) {
            // this wave's CPW chunks are contiguous (S16 = NW * CPW for every supported D): one base, the chunk as the
            // instruction offset, which applies to the global and the LDS address alike
            const char *s0 = src + wave * (CPW * 1024) + lane * 16;
            char *d0 = lds + (t & (NBUF - 1)) * IMG_BYTES + wave * (CPW * 1024);
            switch (q) {
            case 0: glds16_off<0>(s0, d0); break;
            case 1: glds16_off<1024>(s0, d0); break;
            case 2: glds16_off<2048>(s0, d0); break;
            default: glds16_off<3072>(s0, d0); break;
            }
        } else {
            glds4(src + IMG_BYTES + lane * 4, enraw + ((t & (NBUF - 1)) * NW + wave) * 64);
        }
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < PER_TILE; ++q) issue_piece(t, q);
    };
    const int tile_id = SPLIT ? (int)(blockIdx.x / (unsigned)ksplit) : xcd_swizzle(blockIdx.x, gridDim.x);
    // SEL == 2 parks the coarser branches in the ring slots from `pre` on: 2 slots = D x 128 B for the 2x-coarser
    // branch (dual: slots 2, 3; triple: slots 1, 2), slot 3 for the triple's 4x-coarser branch (D x 32 B)
    const int pre = (SEL == 2) ? ((rv.G == 2) ? 2 : 1) : 3;  // code tiles in flight before the prologue

    const int n_raw = (tile_id * NW + wave) * 32 + c;
    const int n = (n_raw < N) ? n_raw : -1;
    auto token_base = [&]() -> size_t {
        const long nn = (n >= 0) ? n : N - 1;
        if constexpr (FLAT) return (size_t)nn * D + 8 * h;
        const long bimg = nn / HW;
        const int hw = (int)(nn - bimg * HW);
        return ((size_t)bimg * D + 8 * h) * HW + hw;
    };
    // The 128 loads and 128 stores of a lane go through BUFFER instructions: a wave-uniform base (the resource: lane 0's token, the
    // smallest of the wave, or the image's base) + a 32-bit lane offset in ONE vector register + the channel's stride in a scalar
    // register -- no vector instruction per access (global_load / global_store took one 64-bit vector add each: 270 of a block's
    // ~8000 instructions).  D * HW < 2^29 (checked by the launcher) keeps every byte offset below 2^31.
    auto wave_base = [&](const float *p0) -> __amdgpu_buffer_rsrc_t {     // resource at p0 + (lane 0's token_base())
        const size_t tb = token_base();
        const size_t tb0 = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(tb >> 32)) << 32) |
                           (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(tb & 0xFFFFFFFFu));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p0 + tb0), 0, -1, 0x00020000);
    };
    auto lane_off = [&]() -> unsigned {                      // byte offset of this lane's token_base() from lane 0's
        const size_t tb = token_base();
        const size_t tb0 = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(tb >> 32)) << 32) |
                           (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(tb & 0xFFFFFFFFu));
        return (unsigned)(tb - tb0) * 4u;
    };
    float zf[S16][8];
    float sel_mask = 1.0f;                                   // SEL: the codebook_mask value of this lane's cell (1 / rep^2), NEGATED for
                                                             // the copies of a coarser cell other than its first position (one register
                                                             // through the code loop instead of two)
    int sel_g = 0;                                           // SEL == 2: grain of this lane's cell
    unsigned stg_a = 0, stg_b = 0;                           // SEL == 2: LDS byte address of this lane's value of channel 8h in the
                                                             // image of the 2x-coarser / 4x-coarser branch
    // ---- CONV: h = W x + bias into zf (see the header).  zp = this lane's x at input channel 8h, st = channel stride.
    auto conv_prologue = [&](const float *zp, size_t st) __attribute__((always_inline)) {
        constexpr int QIMG = S16 * 1024;                     // one weight image (hi or lo) of a row tile
        constexpr int QTILE = 2 * QIMG + 256;
        float *bias_l = enraw;                               // [D] bias, channel order (the seeds area is idle until the code loop)
        // group k = the weight images of k-step k (4 pieces of 1 KiB per wave -> ring slot k & 3: [row tile][hi | lo]) and this
        // lane's 8 x values of it.  All of it asm / DMA with counted waits: 12 vector-memory operations per group and wave.
        float xr[3][8];
        const float *xp = zp;
        auto issue_w = [&](int k, int q) __attribute__((always_inline)) {       // weight piece q < 4 of group k
            const int i = 4 * wave + q;                      // piece: row tile i >> 1, hi / lo i & 1
            glds16(cv.wimg + (size_t)(i >> 1) * QTILE + (i & 1) * QIMG + k * 1024 + lane * 16,
                   lds + (k & 3) * IMG_BYTES + i * 1024);
        };
        auto issue_x = [&](int k, int j) __attribute__((always_inline)) {       // x value j < 8 of group k (in order j = 0 .. 7)
            asm volatile("global_load_dword %0, %1, off nt" : "=v"(xr[k % 3][j]) : "v"(xp) : "memory");
            xp += (j == 7) ? 9 * st : st;                    // after the last one: skip the other lane half's 8 channels
        };
        auto issue_group = [&](int k) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_w(k, q);
#pragma unroll
            for (int j = 0; j < 8; ++j) issue_x(k, j);
        };
        glds4(cv.bias + wave * 64 + lane, bias_l + wave * 64);
        issue_group(0);
        issue_group(1);
        issue_group(2);
        f32x16 acc[8];
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t8][r] = 0.0f;
        int ea = 100;                                        // x is scaled by 2^ea (per token; both lane halves agree)
        float sa = ldexpf(1.0f, 100);
        // group g has landed for this wave when at most the (up to two) younger groups are outstanding; its 8 values are scaled
        // and split into the hi / lo B fragments.  The scale follows the running maximum: a value that would reach 2^15 after
        // scaling moves it (exact rescale of the accumulators by a power of two; wave-uniform branch, rare after the first
        // k-steps).
        f16x8 xh, xl;
        auto take_group = [&](int g) __attribute__((always_inline)) {
            if (g <= S16 - 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (g == S16 - 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(xr[g % 3][j])); xv[j] = xr[g % 3][j]; }
            float m = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = vmax_abs(m, xv[j]);
            m = fmaxf(m, __shfl_xor(m, 32));
            const bool grow = (m > 0.0f) && (m < __builtin_inff()) && (m * sa >= 32768.0f);
            if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
                int e;
                (void)frexpf(grow ? m : 1.0f, &e);
                int en = 14 - e;
                en = en > 100 ? 100 : (en < -100 ? -100 : en);
                en = grow ? en : ea;
                if (g > 0) {
                    const float f = ldexpf(1.0f, en - ea);
#pragma unroll
                    for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t8][r] *= f;
                }
                ea = en;
                sa = ldexpf(1.0f, ea);
            }
            u32x4 ph, pl;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const float v0 = xv[2 * j2] * sa, v1 = xv[2 * j2 + 1] * sa;
                const f32x2 vv = {v0, v1};
                const f16x2 hh = __builtin_convertvector(vv, f16x2);
                const f32x2 rr = {v0 - (float)hh[0], v1 - (float)hh[1]};
                const f16x2 ll = __builtin_convertvector(rr, f16x2);
                ph[j2] = __builtin_bit_cast(unsigned, hh);
                pl[j2] = __builtin_bit_cast(unsigned, ll);
            }
            xh = __builtin_bit_cast(f16x8, ph);
            xl = __builtin_bit_cast(f16x8, pl);
        };
        take_group(0);
#pragma unroll
        for (int s = 0; s < S16; ++s) {
            __builtin_amdgcn_s_barrier();                    // k-step s of the weights landed (everybody's pieces); s - 1 consumed
            asm volatile("" ::: "memory");
            // group s + 3 (into the slot of k-step s - 1 and the x registers already converted) is issued piece by piece BETWEEN
            // the row tiles below: each of its 12 vector-memory instructions then issues in the shadow of MFMAs already in the pipe
            const f16x8 bh = xh, bl = xl;
            // the 16 weight fragments of the k-step (per row tile: lo, then hi) through three rotating registers, each read
            // CONV_AHEAD fragments before its MFMAs behind a counted lgkmcnt: left to hipcc every ds_read_b128 was followed by
            // a full LDS round trip (lgkmcnt(0)) in front of its MFMA -- 16 exposed round trips per 24 MFMAs
            const unsigned wa = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(lds + (s & 3) * IMG_BYTES) + lane * 16;
            f16x8 wf[3];
#define CV_OFF(Q) ((((Q) & 1) ? ((Q) - 1) : ((Q) + 1)) * 1024)      /* fragment Q: even = lo of tile Q / 2 (stored second), odd = hi */
#define CV_RD(Q) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(wf[(Q) % 3]) : "v"(wa), "i"(CV_OFF(Q)))
#define CV_WAIT(N, Q) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(wf[(Q) % 3]) :: "memory")
            CV_RD(0); CV_RD(1); CV_RD(2);
            __builtin_amdgcn_sched_barrier(0);
#define CV_TILE(T8, W0, W1)                                                                                              \
            CV_WAIT(W0, 2 * (T8));                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            acc[T8] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(2 * (T8)) % 3], bh, acc[T8], 0, 0, 0);   /* small terms first (qconv.hip) */ \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            if (2 * (T8) + 3 < 16) { CV_RD(2 * (T8) + 3 < 16 ? 2 * (T8) + 3 : 0); }                                       \
            CV_WAIT(W1, 2 * (T8) + 1);                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            acc[T8] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(2 * (T8) + 1) % 3], bl, acc[T8], 0, 0, 0);               \
            acc[T8] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(2 * (T8) + 1) % 3], bh, acc[T8], 0, 0, 0);               \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            if (2 * (T8) + 4 < 16) { CV_RD(2 * (T8) + 4 < 16 ? 2 * (T8) + 4 : 0); }
            // outstanding reads at each wait: fragments Q .. min(Q + 2, 15); "all but the youngest N" = Q has landed
#define CV_W(Q) if (s + 3 < S16) { issue_w(s + 3, Q); __builtin_amdgcn_sched_barrier(0); }
#define CV_X(J) if (s + 3 < S16) { issue_x(s + 3, J); issue_x(s + 3, (J) + 1); __builtin_amdgcn_sched_barrier(0); }
            CV_TILE(0, 2, 2) CV_W(0) CV_TILE(1, 2, 2) CV_W(1) CV_TILE(2, 2, 2) CV_W(2) CV_TILE(3, 2, 2) CV_W(3)
            CV_TILE(4, 2, 2) CV_X(0) CV_TILE(5, 2, 2) CV_X(2) CV_TILE(6, 2, 2) CV_X(4) CV_TILE(7, 1, 0) CV_X(6)
#undef CV_W
#undef CV_X
#undef CV_TILE
#undef CV_WAIT
#undef CV_RD
#undef CV_OFF
            __builtin_amdgcn_sched_barrier(0);               // the MFMAs are issued; the next k-step's conversion runs under them
            if (s + 1 < S16) take_group(s + 1);
        }
        const float unscale = ldexpf(cv.meta->inv_scale_w, -ea);
#pragma unroll
        for (int s = 0; s < S16; ++s) {
            const f32x4 b0 = *(const f32x4 *)(bias_l + 16 * s + 8 * h), b1v = *(const f32x4 *)(bias_l + 16 * s + 8 * h + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                zf[s][j] = __builtin_fmaf(acc[s >> 1][8 * (s & 1) + j], unscale, (j < 4) ? b0[j & 3] : b1v[j & 3]);
            __builtin_amdgcn_sched_barrier(0);               // in place, one k-step's bias at a time (hoisted bias reads spill)
        }
        if (cv.h_all && n >= 0) {                            // tests: the conv's output for every token
            float *hp = cv.h_buf + token_base();
#pragma unroll
            for (int s = 0; s < S16; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) hp[(size_t)(16 * s + j) * HW] = zf[s][j];
        }
        __builtin_amdgcn_s_barrier();                        // every wave is done with the weight slots and the bias:
        asm volatile("" ::: "memory");                       // the ring and the seeds area go to the code tiles
        for (int t = 0; t < 3; ++t) issue(t);
        __builtin_amdgcn_sched_barrier(0);                   // (the conversion below must not be hoisted over this: all of zf is ready)
    };
    if (SEL != 0) {
        // the router select, fused in: grain of this position's cell straight from the gate, source = the branch
        // that won the cell; indices / codebook_mask / the int64 gate are written here as by-products.
        // Ordinary loads whose values are used while an LDS-DMA is in flight make hipcc drain the whole vector-memory
        // queue (s_waitcnt vmcnt(0)), so: the gate is fetched BEFORE the first DMA is issued, and (SEL == 2) reduced only
        // after this wave's loads are on their way.
        const int nn = (n >= 0) ? n : (int)(N - 1);
        const int b = nn / HW, pos = nn - b * HW;
        const int y = pos / rv.Wout, x = pos - y * rv.Wout;
        const int SC = rv.sub[rv.G - 1];
        const size_t cell = (size_t)b * rv.hc * rv.wc + (y / SC) * rv.wc + x / SC;
        const DvqGateRaw graw = dvq_gate_fetch(rv.gate, rv.gate_mode, rv.G, cell);
        auto by_products = [&](int g) {
            const int rep_g = rv.rep[g];
            sel_mask = 1.0f / (float)(rep_g * rep_g);        // 1, 0.25, 0.0625: exact
            if (n >= 0 && h == 0 && rv.cmask_out != nullptr) {
                rv.cmask_out[n] = sel_mask;
                if (y % SC == 0 && x % SC == 0) {
                    rv.indices_out[cell] = g;
                    if (rv.gate_mode == 2 && rv.gate_out != nullptr) {
                        const float e = graw.f[0];
                        longlong2 gg; gg.x = (e <= rv.thr) ? 1 : 0; gg.y = (e > rv.thr) ? 1 : 0;
                        *(longlong2 *)(rv.gate_out + 2 * cell) = gg;
                    }
                }
            }
            if (!(y % rep_g == 0 && x % rep_g == 0)) sel_mask = -sel_mask;
        };
        if (SEL == 1) {
            const int g = dvq_gate_reduce(graw, rv.gate_mode, rv.G, rv.thr);
            by_products(g);
            int stride_l;
            const float *zp = dvq_dense_source(rv, b, y, x, g, stride_l) + (size_t)8 * h * stride_l;
            const size_t st = (size_t)stride_l;
            if constexpr (CONV) {
                conv_prologue(zp, st);
            } else {
            for (int t = 0; t < pre; ++t) issue(t);
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int s = 0; s < S16; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) zf[s][j] = DVQ_LOAD_SEL(zp + (size_t)(16 * s + j) * st);
            __builtin_amdgcn_s_setprio(0);
            }
        } else {
            for (int t = 0; t < pre; ++t) issue(t);
            // workgroup = output rows y0 .. y0 + 3 of image b (wave = row, lane = column); both are wave-uniform
            const int bw = __builtin_amdgcn_readfirstlane(b);
            const int y0 = __builtin_amdgcn_readfirstlane(y) - wave;
            char *img_a = lds + pre * IMG_BYTES;             // 2x-coarser branch [D][32 floats]
            char *img_b = lds + 3 * IMG_BYTES;               // 4x-coarser branch [D][8 floats] (triple only)
            {
                // branch G-2 (rep 2): rows y0/2, y0/2 + 1 of a 16-wide grid = 32 consecutive floats per channel;
                // a wave-instruction moves 8 channels x 8 pieces of 16 B
                const int ga = rv.G - 2;
                const int plane = rv.hc * rv.sub[ga] * 16;
                const float *src = rv.src[ga] + (size_t)bw * D * plane + (size_t)(y0 >> 1) * 16 + (lane & 7) * 4;
                for (int i = wave; i < D / 8; i += NW)
                    glds16(src + (size_t)(i * 8 + (lane >> 3)) * plane, img_a + i * 1024);
            }
            if (rv.G == 3) {
                // branch 0 (rep 4): row y0/4 of an 8-wide grid = 8 floats per channel; 32 channels x 2 pieces per instruction
                const int plane = rv.hc * 8;
                const float *src = rv.src[0] + (size_t)bw * D * plane + (size_t)(y0 >> 2) * 8 + (lane & 1) * 4;
                for (int i = wave; i < D / 32; i += NW)
                    glds16(src + (size_t)(i * 32 + (lane >> 1)) * plane, img_b + i * 1024);
            }
            stg_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)img_a +
                    (unsigned)((8 * h) * 128 + (((wave >> 1) * 16 + (c >> 1)) << 2));
            stg_b = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)img_b +
                    (unsigned)((8 * h) * 32 + ((c >> 2) << 2));
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void *)(rv.src[rv.G - 1] + (size_t)bw * D * HW), 0,
                                                                                -1, 0x00020000);      // the image's plane stack
            const unsigned zo = (unsigned)(8 * h * HW + pos) * 4u;
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int s = 0; s < S16; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) zf[s][j] = DVQ_BUF_LOAD(zr, zo, (16 * s + j) * HW * 4, NT ? 2 : 0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the branch images (and its loads) landed;
            __builtin_amdgcn_s_barrier();                      // the barrier makes that true for the other waves' pieces
            asm volatile("" ::: "memory");
            sel_g = dvq_gate_reduce(graw, rv.gate_mode, rv.G, rv.thr);
            by_products(sel_g);
        }
    } else if constexpr (CONV) {
        conv_prologue(z + token_base(), (size_t)HW);
    } else if constexpr (FLAT) {
        // Row-major latents.  Read as what they are -- every wave-instruction fetches whole 128-byte lines (a token's HALF row,
        // D * 2 bytes, is contiguous: lane = 16-byte piece) -- and turned into the (token, 8 channels of a k-step) register
        // layout through a wave-private LDS image [32 tokens][D / 2 floats + 16 B pad] (the pad makes the b128 reads of lanes
        // c .. c + 7 hit distinct banks), one half of the channels at a time.  The image lives where the code ring will: the
        // first code tiles are DMA'd after a workgroup barrier, and land while the fragments are converted.
        // (The direct form -- lane = token, two 16-byte loads per k-step at a stride of D * 4 bytes -- touched every line from
        // eight instructions and ran at 2x the NCHW kernel's time; profiles/r05_flat.json.)
        char *tr = lds + wave * FLAT_TRW;
        const long n0 = ((long)tile_id * NW + wave) * 32;    // the wave's first token (its 32 tokens are consecutive rows)
        f32x4 tmp[2][FLAT_IPH];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int i = 0; i < FLAT_IPH; ++i) {
                long tk = n0 + i * FLAT_TPI + lane / FLAT_LPT;
                tk = tk < N ? tk : N - 1;
                const f32x4 *src = (const f32x4 *)(z + (size_t)tk * D + h2 * (D / 2)) + (lane % FLAT_LPT);
                tmp[h2][i] = NT ? __builtin_nontemporal_load(src) : *src;
            }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int i = 0; i < FLAT_IPH; ++i)
                *(f32x4 *)(tr + (i * FLAT_TPI + lane / FLAT_LPT) * FLAT_RSH + (lane % FLAT_LPT) * 16) = tmp[h2][i];
#pragma unroll
            for (int sp = 0; sp < S16 / 2; ++sp) {
                const int s = h2 * (S16 / 2) + sp;
                const f32x4 lo = *(const f32x4 *)(tr + c * FLAT_RSH + (16 * sp + 8 * h) * 4);
                const f32x4 hi = *(const f32x4 *)(tr + c * FLAT_RSH + (16 * sp + 8 * h + 4) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { zf[s][j] = lo[j]; zf[s][4 + j] = hi[j]; }
            }
        }
        __syncthreads();                                     // every wave has read its image: the region becomes the code ring
        for (int t = 0; t < pre; ++t) issue(t);
    } else {
        for (int t = 0; t < pre; ++t) issue(t);
        const __amdgpu_buffer_rsrc_t zr = wave_base(z);
        const unsigned zo = lane_off();
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int s = 0; s < S16; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) zf[s][j] = DVQ_BUF_LOAD(zr, zo, (16 * s + j) * HW * 4, NT ? 2 : 0);
        __builtin_amdgcn_s_setprio(0);
    }
    f16x8 zb[2][S32];                                        // B operands of the 16x16x32 loop, [token half][k-step of 32]
    float xn, thr2W;
    {
        float pa[2][8];
        float amax = 0.0f, zeta2 = 0.0f;
        f16x8 zprev = {};
#pragma unroll
        for (int s = 0; s < S16; ++s) {
            if (SEL == 2) {
                // lanes whose cell went to a coarser branch: its value of channel 16 s + 8 h + j replaces the fine one
                // (the LDS reads execute under the lanes' exec mask and land in the same registers: no select needed)
                if (sel_g == rv.G - 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        asm volatile("ds_read_b32 %0, %1 offset:%c2" : "+v"(zf[s][j]) : "v"(stg_a), "i"((16 * s + j) * 128));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (rv.G == 3 && sel_g == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        asm volatile("ds_read_b32 %0, %1 offset:%c2" : "+v"(zf[s][j]) : "v"(stg_b), "i"((16 * s + j) * 32));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(zf[s][j]));
            }
            u32x4 packed;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const float v0 = zf[s][2 * j2], v1 = zf[s][2 * j2 + 1];
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
            const f16x8 zcur = __builtin_bit_cast(f16x8, packed);
            if (s & 1) {
                // tokens 16 t2 + (lane & 15), k = 32 s' + 8 (lane >> 4) + j  <-  lane (c, h) = (16 t2 + (lane & 15), (lane >> 4) & 1),
                // k-step 2 s' + (lane >> 5) of the load layout: through the per-wave LDS scratch (a wave's LDS
                // operations execute in order, so no barrier; 128 ds_bpermutes instead spilled 54 VGPRs)
                const int sp = s >> 1;
                *(f16x8 *)(scr + lane * 16) = zprev;
                *(f16x8 *)(scr + 1024 + lane * 16) = zcur;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int srcl = 16 * t2 + (lane & 15) + 32 * ((lane >> 4) & 1);
                    zb[t2][sp] = *(const f16x8 *)(scr + (lane >> 5) * 1024 + srcl * 16);
                }
            }
            zprev = zcur;
            if constexpr (CONV) __builtin_amdgcn_sched_barrier(0);   // zf is complete before the loop: keep the k-steps in order
        }
        float t8[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float o0 = __shfl_xor(pa[0][l], 32), o1 = __shfl_xor(pa[1][l], 32);
            float a0 = h == 0 ? pa[0][l] : o0;
            float a1 = h == 0 ? o0 : pa[0][l];
            float a2 = h == 0 ? pa[1][l] : o1;
            float a3 = h == 0 ? o1 : pa[1][l];
            t8[l] = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
        }
        xn = t8[0];
#pragma unroll
        for (int l = 1; l < 8; ++l) xn = __fadd_rn(xn, t8[l]);
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        zeta2 += __shfl_xor(zeta2, 32);
        // FOLD: z is the conv's input, `img` / `meta` the folded codebook E W (vq_fold.hip), and the bound also covers the conv
        thr2W = FOLD ? dvq_fold_threshold(xn, amax, zeta2, sB, (const DvqFoldMeta *)meta)
                     : dvq_filter_threshold(xn, amax, zeta2, sB, meta);
    }
    if (SEL == 2) {
        __builtin_amdgcn_s_barrier();                        // every wave has read the branch images: the slots join the ring
        asm volatile("" ::: "memory");
    }
    for (int t = pre; t < 3; ++t) issue(t);                  // (SEL == 2) the code tiles that waited for those slots
    DVQ_STAMP(1);

    // ---- 16x16x32 code loop: fragment F = c2 * S32 + s' of the tile feeds two MFMAs (token halves t2 = 0, 1);
    // accumulator acc16[c2][t2][i] = code 16 c2 + 4 (lane >> 4) + i against token 16 t2 + (lane & 15)
    float best, second;
    int code;
    {
        const int q16 = lane >> 4;
        float b1[2] = {-__builtin_inff(), -__builtin_inff()}, b2[2] = {-__builtin_inff(), -__builtin_inff()};
        int bt[2] = {0, 0};
        // Per tile: barrier -> the first four A-fragment reads are issued -> the running top-2 is updated with the PREVIOUS
        // tile's scores (plain VALU work that hides the LDS latency of those reads) -> the accumulators are re-seeded ->
        // MFMA chain.  A wave's instruction ISSUE, not the matrix pipe, bounds this loop: about 1330 cycles per tile, of which
        // the pipe is busy 512; a workgroup alone on a CU takes as long as two sharing it (profiles/archive/r03_pass1_antiphase_ab.json,
        // r03_pass1_loop_ablation.json).  Moving the top-2 update into the shadow of the MFMAs (one code half behind them, no
        // second accumulator set) changed nothing, as that model predicts: 42.7k vs 42.6k cycles per loop
        // (profiles/archive/r03_pass1_half_tile_pipelining.json; git history has the code).
        f32x4 acc16[2][2];
        auto top2 = [&](int tt) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float om = b1[t2];
#pragma unroll
                for (int r = 0; r < 8; r += 2) {             // r = 4 c2 + i
                    // running top-2 over the pair (g0, g1): with b2 <= b1 the new second-best is max(b2, med3(b1, g0, g1))
                    // and the new best max3(b1, g0, g1): 2.5 VALU ops per score; the register index rides in 4 mantissa bits
                    const float v0 = acc16[r >> 2][t2][r & 3], v1 = acc16[(r + 1) >> 2][t2][(r + 1) & 3];
                    float g0 = __uint_as_float((__float_as_uint(v0) & 0xFFFFFFF0u) | (unsigned)r);
                    float g1 = __uint_as_float((__float_as_uint(v1) & 0xFFFFFFF0u) | (unsigned)(r + 1));
                    float md = __builtin_amdgcn_fmed3f(b1[t2], g0, g1);
                    b1[t2] = vmax3_raw(b1[t2], g0, g1);
                    b2[t2] = vmax_raw(b2[t2], md);
                }
                bt[t2] = (b1[t2] != om) ? tt : bt[t2];
            }
        };
    for (int t = 0; t < T; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");   // all but the youngest tile's DMA: tiles <= t + 1 landed
            __builtin_amdgcn_s_barrier();                    // tile t (everybody's DMA) landed; t-1 consumed
            asm volatile("" ::: "memory");
            if (S16 != 16) issue(t + 3);                     // D = 256: pieces ride between the MFMAs below
            // A fragments: hand-placed LDS reads, four k-steps ahead of the MFMA that consumes them
            // (ds_read returns in order: lgkmcnt(3) = "the oldest of my four reads has landed")
            const unsigned tile_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(
                                        lds + (t & (NBUF - 1)) * IMG_BYTES + lane * 16);
            f16x8 a0, a1, a2, a3;
#define DVQ_RD(dst, S) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(tile_a), "i"((S) * 1024))
            DVQ_RD(a0, 0); DVQ_RD(a1, 1); DVQ_RD(a2, 2); DVQ_RD(a3, 3);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) top2(t - 1);
            __builtin_amdgcn_sched_barrier(0);
            {
                // accumulator seeds of tile t: this wave's own DMA copy (landed by the wait above), read behind the fragments
                const unsigned seed_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(
                                            enraw + ((t & (NBUF - 1)) * NW + wave) * 64 + 4 * q16);
                f32x4 e0, e1;
                asm volatile("ds_read_b128 %0, %1" : "=v"(e0) : "v"(seed_a));
                asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(e1) : "v"(seed_a));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
                acc16[0][0] = e0; acc16[0][1] = e0; acc16[1][0] = e1; acc16[1][1] = e1;
            }
            __builtin_amdgcn_sched_barrier(0);
#define DVQ_PIECE(Q) issue_piece(t + 3, Q);
#define DVQ_MM(src, F, WAIT, NEXT)                                                                             \
            asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" ::: "memory");                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            acc16[(F) / S32][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[0][(F) % S32], acc16[(F) / S32][0], 0, 0, 0); \
            acc16[(F) / S32][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[1][(F) % S32], acc16[(F) / S32][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if ((F) + 4 < S16) { DVQ_RD(src, ((F) + 4 < S16 ? (F) + 4 : 0)); }                                  \
            NEXT
            __builtin_amdgcn_s_setprio(1);
            if (S16 == 16) {
                // the next ring tile's DMA pieces are issued between MFMAs: each ~100-cycle issue stall
                // then overlaps the MFMA already in the pipe instead of preceding the whole chain
                DVQ_MM(a0, 0, 0, ) DVQ_MM(a1, 1, 1, DVQ_PIECE(0)) DVQ_MM(a2, 2, 2, ) DVQ_MM(a3, 3, 3, )
                DVQ_MM(a0, 4, 3, DVQ_PIECE(1)) DVQ_MM(a1, 5, 3, ) DVQ_MM(a2, 6, 3, ) DVQ_MM(a3, 7, 3, DVQ_PIECE(2))
                DVQ_MM(a0, 8, 3, ) DVQ_MM(a1, 9, 3, ) DVQ_MM(a2, 10, 3, DVQ_PIECE(3)) DVQ_MM(a3, 11, 3, )
                DVQ_MM(a0, 12, 3, ) DVQ_MM(a1, 13, 2, DVQ_PIECE(4)) DVQ_MM(a2, 14, 1, ) DVQ_MM(a3, 15, 0, )
            } else if (S16 == 8) {
                DVQ_MM(a0, 0, 0, ) DVQ_MM(a1, 1, 1, ) DVQ_MM(a2, 2, 2, ) DVQ_MM(a3, 3, 3, )
                DVQ_MM(a0, 4, 3, ) DVQ_MM(a1, 5, 2, ) DVQ_MM(a2, 6, 1, ) DVQ_MM(a3, 7, 0, )
            } else {
                DVQ_MM(a0, 0, 0, ) DVQ_MM(a1, 1, 0, ) DVQ_MM(a2, 2, 0, ) DVQ_MM(a3, 3, 0, )
            }
#undef DVQ_MM
#undef DVQ_RD
#undef DVQ_PIECE
            __builtin_amdgcn_s_setprio(0);
        }
        top2(T - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // surplus ring DMA
        // merge the four lane groups of a token column (lower lane wins ties), then hand the results to the lanes
        // that own the token in the (c, h) layout of the prologue / epilogue
        float rb[2], rs[2];
        int rc[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            float mb = b1[t2], ms = b2[t2];
            int mt = bt[t2], mq = q16;
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float o1 = __shfl_xor(mb, off), o2 = __shfl_xor(ms, off);
                const int ot = __shfl_xor(mt, off), oq = __shfl_xor(mq, off);
                const bool other_wins = (o1 > mb) || (o1 == mb && ((lane ^ off) < lane));
                ms = fmaxf(other_wins ? mb : o1, fmaxf(ms, o2));
                mb = other_wins ? o1 : mb;
                mt = other_wins ? ot : mt;
                mq = other_wins ? oq : mq;
            }
            const int r = (int)(__float_as_uint(mb) & 15u);
            rb[t2] = mb;
            rs[t2] = ms;
            rc[t2] = (mt + t_lo) * 32 + 16 * (r >> 2) + 4 * mq + (r & 3);
        }
        const int srcl = c & 15;
        const float x0 = __shfl(rb[0], srcl), x1 = __shfl(rb[1], srcl);
        const float y0 = __shfl(rs[0], srcl), y1 = __shfl(rs[1], srcl);
        const int c0 = __shfl(rc[0], srcl), c1 = __shfl(rc[1], srcl);
        best = (c >> 4) ? x1 : x0;
        second = (c >> 4) ? y1 : y0;
        code = (c >> 4) ? c1 : c0;
    }
    DVQ_STAMP(2);
    if constexpr (SPLIT) {
        // hand-off without fences (MI355X guide, inter-workgroup visibility: every payload store write-through (sc1) and drained by
        // its wave, the workgroup's barrier, ONE agent-scope add per workgroup; the workgroup whose add came last reads with sc1
        // loads after a barrier its adding wave joins).  A __threadfence() pair instead cost 3-70 us with the grid size.
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        __shared__ int s_last;
        gu64 *mine = (gu64 *)(split + ((size_t)tile_id * ksplit) * 128 + wave * 32 + c);      // [block][slice][128 tokens] x 16 B
        if (h == 0) {
            gu64 *e = mine + (size_t)ks * 128 * 2;
            __hip_atomic_store(e, ((unsigned long long)__float_as_uint(second) << 32) | __float_as_uint(best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(e + 1, (unsigned long long)(unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        DVQ_STAMP(3);
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&counters[DVQ_SPLIT_TICKET0 + tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == ksplit - 1);
            if (old == ksplit - 1)                           // self-cleaning (DVQ_MODE_WS_CLEAN)
                __hip_atomic_store(&counters[DVQ_SPLIT_TICKET0 + tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        DVQ_STAMP(4);
        if (!s_last) return;
        float mb = -__builtin_inff(), ms = -__builtin_inff();
        int mc = 0;
        unsigned long long e0[DVQ_SPLIT_MAX_SLICES], e1[DVQ_SPLIT_MAX_SLICES];   // all slices' entries in flight at once (past the end: repeats)
#pragma unroll
        for (int k = 0; k < DVQ_SPLIT_MAX_SLICES; ++k) {
            const int kk = k < ksplit ? k : ksplit - 1;
            e0[k] = __hip_atomic_load(mine + (size_t)kk * 128 * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            e1[k] = __hip_atomic_load(mine + (size_t)kk * 128 * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < DVQ_SPLIT_MAX_SLICES; ++k) {
            const float eb = __uint_as_float((unsigned)e0[k]), es = __uint_as_float((unsigned)(e0[k] >> 32));
            const bool other_wins = k < ksplit && eb > mb;
            ms = (k < ksplit) ? fmaxf(other_wins ? mb : eb, fmaxf(ms, es)) : ms;
            mc = other_wins ? (int)(unsigned)e1[k] : mc;
            mb = other_wins ? eb : mb;
        }
        best = mb; second = ms; code = mc;
        DVQ_STAMP(5);
    }
    const float thr = best - thr2W;
    const bool final_ok = (best - second) > thr2W;
    const bool valid = n >= 0;
    bool hopeless = !(code < K) || !(thr == thr);
#ifdef DVQ_TUNING
    if (g_dvq_tokdbg != nullptr && valid && h == 0) {
        f32x4 dbg = {best, second, thr2W, (float)code};
        *(f32x4 *)(g_dvq_tokdbg + 4 * (size_t)n) = dbg;
    }
#endif
    // undecided tokens are queued for the resolver.  The slot comes from an atomic whose result is not
    // needed until the record is written, so: bump the shard counter now (one atomic per wave, lane 0,
    // by the number of undecided tokens), run the z_q / loss phase while it is in flight, and only then
    // read it back and dump the records.
    const bool undecided = valid && !hopeless && !final_ok;
    // routed op: the rep x rep output positions of a coarser cell are copies of ONE vector -- same scores, same bound, undecided
    // together -- so only the cell's first position queues a record (RecMeta.rep) and the resolver corrects all of them: 37 %
    // fewer records at a fine ratio of 0.5 (dual), and the resolver's chunks then fit one per CU
    const int sel_rep = (SEL == 0 || sel_mask < 0.0f) ? (SEL == 0 ? 1 : 0) : (sel_mask == 1.0f ? 1 : (sel_mask == 0.25f ? 2 : 4));
    const bool queued = undecided && sel_rep > 0;           // sel_rep: 0 for a copy, else positions per edge of the lane's cell
    const unsigned long long umask = __ballot(queued && h == 0);
    // (SPLIT: which workgroup merges a block varies from run to run -- the shard is a function of the tokens, per wave, so that the
    // queue's layout, the fallback counts and the resolver's chunks do not)
    const int shard = (SPLIT ? tile_id * NW + wave : (int)blockIdx.x) & (DVQ_QSHARDS - 1);
    int slot_raw = 0;
    if (umask != 0ull && lane == 0) slot_raw = atomicAdd(&counters[DVQ_QCOUNT0 + shard], (int)__popcll(umask));
    if (valid && hopeless && h == 0) {
        int pos = atomicAdd(&counters[DVQ_C_EXACT], 1);
        exact_list[pos] = n;
    }
    // (CONV: the exact-list kernel computes the conv output of its tokens itself, from the conv's input -- round 6; through round 5
    // pass 1 spilled their rows to a full-size [B, D, HW] scratch tensor, 256 MiB per stream at B = 256)
    float lsum = 0.0f;
    float m_tok = 1.0f;
    if (valid && !hopeless) {
        if (h == 0) codes[n] = (long long)code;
        m_tok = (SEL != 0) ? __builtin_fabsf(sel_mask) : ((mask != nullptr) ? mask[n] : 1.0f);
        if (zq != nullptr || partials != nullptr) {
            const float *ep = E + (size_t)code * D + 8 * h;
            // gathers per batch: 2 k-steps (A/B on MI355X: 2 beats 1, 4, 8 and a 3-deep pipeline) where other workgroups hide the
            // latency; the split form's merging workgroup is alone on its CU and takes 8 (two round trips instead of eight)
            constexpr int SB = SPLIT ? ((S16 < 8) ? S16 : 8) : ((S16 < 2) ? S16 : 2);
            // `zq != nullptr` is tested ONCE (a scalar branch on the kernel argument): with the test
            // inside the loop on the per-lane pointer every one of the 128 stores became its own
            // exec-masked branch to an out-of-line block.
            auto finish = [&](auto store_tag) {
                constexpr bool STORE = decltype(store_tag)::value;
                const __amdgpu_buffer_rsrc_t qr = wave_base(STORE ? zq : (float *)E);
                const unsigned qo = lane_off();
                int hw4 = HW * 4;                            // opaque: the 128 scalar offsets are recomputed here (two scalar
                asm volatile("" : "+s"(hw4));                // instructions each) instead of living in spilled SGPRs since the prologue
#pragma unroll
                for (int s0 = 0; s0 < S16; s0 += SB) {
                    f32x4 eg[SB][2];
#pragma unroll
                    for (int q = 0; q < SB; ++q) {
                        eg[q][0] = *(const f32x4 *)(ep + 16 * (s0 + q));
                        eg[q][1] = *(const f32x4 *)(ep + 16 * (s0 + q) + 4);
                    }
#pragma unroll
                    for (int q = 0; q < SB; ++q) {
                        const int s = s0 + q;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float e = eg[q][j >> 2][j & 3];
                            if constexpr (FOLD) {           // the registers hold the conv's INPUT: z_q := e[code] (within 1e-6 of
                                if (STORE) DVQ_BUF_STORE(e, qr, qo, (16 * s + j) * hw4);   // fl(h + fl(e - h))), no loss term
                            } else {
                            float diff = __fsub_rn(e, zf[s][j]);
                            if (STORE) DVQ_BUF_STORE(__fadd_rn(zf[s][j], diff), qr, qo, (16 * s + j) * hw4);
                            lsum = __builtin_fmaf(diff, diff, lsum);   // the token's loss weight is applied once, below
                            }
                        }
                    }
                }
            };
            if constexpr (!FLAT) {
                if (zq != nullptr) finish(std::true_type{});
                else finish(std::false_type{});
                lsum *= m_tok;
            }
        }
    }
    if constexpr (FLAT) {
        // Row-major z_q: the lanes' values go through the wave's LDS image (one half of the channels at a time) and leave as
        // whole 128-byte lines, 16 bytes per lane -- the prologue's path backwards.  The image overlays the code ring: every wave
        // is past its last tile (barrier) and its surplus DMA has landed (the wait after the loop).
        char *flat_tr = lds + wave * FLAT_TRW;
        const bool store = zq != nullptr;                    // kernel argument: uniform
        if (store) __syncthreads();
        const bool mine = valid && !hopeless && (store || partials != nullptr);
        const unsigned long long okmask = __ballot(valid && !hopeless && h == 0);
        const long n0 = ((long)tile_id * NW + wave) * 32;
        const float *ep = E + (size_t)(mine ? code : 0) * D + 8 * h;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            if (mine) {
                constexpr int SB = (S16 / 2 < 2) ? S16 / 2 : 2;
#pragma unroll
                for (int sp0 = 0; sp0 < S16 / 2; sp0 += SB) {
                    f32x4 eg[SB][2];
#pragma unroll
                    for (int q = 0; q < SB; ++q) {
                        eg[q][0] = *(const f32x4 *)(ep + 16 * (h2 * (S16 / 2) + sp0 + q));
                        eg[q][1] = *(const f32x4 *)(ep + 16 * (h2 * (S16 / 2) + sp0 + q) + 4);
                    }
#pragma unroll
                    for (int q = 0; q < SB; ++q) {
                        const int sp = sp0 + q, s = h2 * (S16 / 2) + sp;
                        f32x4 o[2];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float e = eg[q][j >> 2][j & 3];
                            if constexpr (FOLD) {
                                o[j >> 2][j & 3] = e;           // z_q := e[code] (see the NCHW form)
                            } else {
                                const float diff = __fsub_rn(e, zf[s][j]);
                                o[j >> 2][j & 3] = __fadd_rn(zf[s][j], diff);
                                lsum = __builtin_fmaf(diff, diff, lsum);
                            }
                        }
                        if (store) {
                            *(f32x4 *)(flat_tr + c * FLAT_RSH + (16 * sp + 8 * h) * 4) = o[0];
                            *(f32x4 *)(flat_tr + c * FLAT_RSH + (16 * sp + 8 * h + 4) * 4) = o[1];
                        }
                    }
                }
            }
            if (store) {
#pragma unroll
                for (int i = 0; i < FLAT_IPH; ++i) {
                    const int tk = i * FLAT_TPI + lane / FLAT_LPT;
                    const f32x4 v = *(const f32x4 *)(flat_tr + tk * FLAT_RSH + (lane % FLAT_LPT) * 16);
                    if ((okmask >> tk) & 1ull)              // (hopeless tokens: the exact-list kernel writes their rows)
                        __builtin_nontemporal_store(v, (f32x4 *)(zq + (size_t)(n0 + tk) * D + h2 * (D / 2)) + (lane % FLAT_LPT));
                }
            }
        }
        lsum *= m_tok;
    }
    DVQ_STAMP(6);
    if (umask != 0ull) {                                    // wave-uniform
        const int base = __shfl(slot_raw, 0);
        int slot = base + (int)__popcll(umask & ((1ull << c) - 1ull));   // rank among the wave's queued tokens
        slot = queued ? slot : -1;
        if (queued && slot >= rec_cap) {                    // shard full: full exact evaluation instead; the
            if (h == 0) {                                   // provisional code / z_q written above are overwritten
                const int rr = sel_rep > 0 ? sel_rep : 1;   // by the exact-list kernel, the loss term is dropped here
                for (int ry = 0; ry < rr; ++ry)
                    for (int rx = 0; rx < rr; ++rx) {
                        int pos = atomicAdd(&counters[DVQ_C_EXACT], 1);
                        exact_list[pos] = n + ry * rv.Wout + rx;
                    }
            }
            lsum = -(float)(sel_rep * sel_rep - 1) * lsum;  // ... for every copy of the cell (their terms equal this lane's bit for bit)
            slot = -1;
        }
        if (slot >= 0) {
            char *rec = records + ((size_t)shard * rec_cap + slot) * rec_bytes(D);
#pragma unroll
            for (int s = 0; s < S16; ++s) {
                f32x4 lo = {zf[s][0], zf[s][1], zf[s][2], zf[s][3]};
                f32x4 hi = {zf[s][4], zf[s][5], zf[s][6], zf[s][7]};
                *(f32x4 *)(rec + (16 * s + 8 * h) * 4) = lo;
                *(f32x4 *)(rec + (16 * s + 8 * h + 4) * 4) = hi;
            }
            if (h == 0) {
                RecMeta rm;
                rm.n = n; rm.xn = xn; rm.thr = thr; rm.m = m_tok; rm.prov = code;
                rm.best = ~0ull; rm.rep = sel_rep > 0 ? sel_rep : 1;
                *(RecMeta *)(rec + (size_t)D * 4) = rm;
            }
        }
    }
    if (partials != nullptr) {
        double dsum = (double)lsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        __syncthreads();
        double *red = (double *)lds;
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) partials[SPLIT ? tile_id : (int)blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    DVQ_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// pass 1, large codebooks ("wide" form, D = 256): a wave scores TWO blocks of 32 tokens against every
// code tile, so each A fragment read from LDS feeds two MFMAs and the ring DMA / barrier per tile are
// amortised over 32 MFMAs instead of 16.  There is no room left for the fp32 copy of z (the two blocks'
// fp16 fragments take 128 VGPRs): z is read again in the epilogue -- 2 KiB per token next to the
// >= 2 MiB of codebook every token is scored against.  Same top-2 tracking, same bound, same queue,
// records and outputs as vq_assign_filter_kernel.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_wide_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, int nparts_pass1)
{
    constexpr int NW = 4;
    constexpr int S16 = D / 16;
    static_assert(S16 == 16, "the wide form is written for D = 256");
    constexpr int IMG_BYTES = S16 * 1024;
    constexpr int TILE_STRIDE = IMG_BYTES + 256;
    constexpr int CPW = (S16 + NW - 1) / NW;
    constexpr int PER_TILE = CPW + 1;
    constexpr int NBUF = 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *enraw = (float *)(lds + NBUF * IMG_BYTES);        // [NBUF][NW][64] accumulator seeds, per-wave copy

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int T = dvq_num_tiles(K);
    const float sB = meta->scale_b;

    auto issue_piece = [&](int t, int q) {
        const int tt = (t < T) ? t : T - 1;
        const char *src = img + (size_t)tt * TILE_STRIDE;
        if (q < CPW) {
            const char *s0 = src + wave * (CPW * 1024) + lane * 16;   // (one base + instruction offsets: vq_assign_filter_kernel)
            char *d0 = lds + (t & (NBUF - 1)) * IMG_BYTES + wave * (CPW * 1024);
            switch (q) {
            case 0: glds16_off<0>(s0, d0); break;
            case 1: glds16_off<1024>(s0, d0); break;
            case 2: glds16_off<2048>(s0, d0); break;
            default: glds16_off<3072>(s0, d0); break;
            }
        } else {
            glds4(src + IMG_BYTES + lane * 4, enraw + ((t & (NBUF - 1)) * NW + wave) * 64);
        }
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < PER_TILE; ++q) issue_piece(t, q);
    };
    issue(0);
    issue(1);
    issue(2);

    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    int nn[2];                                               // token of this lane in block u, -1 = past the end
    size_t zbase[2];                                         // element offset of its channel 8h
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long n_raw = ((long)tile_id * NW + wave) * 64 + 32 * u + c;
        nn[u] = (n_raw < N) ? (int)n_raw : -1;
        const long q = (nn[u] >= 0) ? nn[u] : N - 1;
        const long bimg = q / HW;
        zbase[u] = ((size_t)bimg * D + 8 * h) * HW + (size_t)(q - bimg * HW);
    }

    // ---- prologue: per block, z in batches of four k-steps -> fp16 fragments, exact-order norm, bound
    f16x8 zh[2][2];                                          // only the current pair of k-steps lives in the load layout
    f16x8 zb[2][2][S16 / 2];                                 // [block][token half][k-step of 32] in 16x16x32 operand order
    float xn[2], thr2W[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float *zp = z + zbase[u];
        float pa[2][8];
        float amax = 0.0f, zeta2 = 0.0f;
        const float *zpb = zp;                              // advances by four k-steps per batch
#pragma unroll
        for (int sb = 0; sb < S16; sb += 4) {
            float zf[4][8];
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) zf[q][j] = DVQ_LOAD_Z(zpb + (size_t)(16 * q + j) * HW);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
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
                zh[u][s & 1] = __builtin_bit_cast(f16x8, packed);
                if (s & 1) {                                 // same permutation as vq_assign_filter_kernel, per-wave LDS scratch
                    char *scr = lds + NBUF * IMG_BYTES + NBUF * NW * 64 * 4 + wave * 2048;
                    *(f16x8 *)(scr + lane * 16) = zh[u][0];
                    *(f16x8 *)(scr + 1024 + lane * 16) = zh[u][1];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const int srcl = 16 * t2 + (lane & 15) + 32 * ((lane >> 4) & 1);
                        zb[u][t2][s >> 1] = *(const f16x8 *)(scr + (lane >> 5) * 1024 + srcl * 16);
                    }
                }
            }
            // one batch of 32 loads at a time (register budget): the next batch's addresses depend,
            // opaquely, on this batch's last converted fragment
            zpb += (size_t)64 * HW;
            asm volatile("" : "+v"(zpb) : "v"(zb[u][1][(sb + 3) >> 1]));
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
        float x = t8[0];
#pragma unroll
        for (int l = 1; l < 8; ++l) x = __fadd_rn(x, t8[l]);
        xn[u] = x;
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        zeta2 += __shfl_xor(zeta2, 32);
        thr2W[u] = dvq_filter_threshold(x, amax, zeta2, sB, meta);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tiles 0..2 (own DMA) landed during the prologue

    // ---- code loop
    int code[2];
    float thr[2];
    bool undecided[2], hopeless[2], valid[2];
    float bestv[2], secondv[2];
    {
        // 16x16x32 form: every A fragment (16 codes x 32 k) feeds four MFMAs (two blocks x two token halves)
        constexpr int S32 = S16 / 2;
        const int q16 = lane >> 4;
        float b1[2][2], b2[2][2];
        int bt[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) { b1[u][t2] = -__builtin_inff(); b2[u][t2] = -__builtin_inff(); bt[u][t2] = 0; }
        for (int t = 0; t < T; ++t) {
            const float *seeds = enraw + ((t & (NBUF - 1)) * NW + wave) * 64 + 4 * q16;
            f32x4 acc16[2][2][2];                            // [block][code half][token half]
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const f32x4 e4 = *(const f32x4 *)(seeds + 16 * c2);
                acc16[0][c2][0] = e4; acc16[0][c2][1] = e4; acc16[1][c2][0] = e4; acc16[1][c2][1] = e4;
            }
            if (t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned tile_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(
                                        lds + (t & (NBUF - 1)) * IMG_BYTES + lane * 16);
            f16x8 a0, a1, a2, a3;
            asm volatile("" : "+v"(acc16[0][0][0]), "+v"(acc16[0][0][1]), "+v"(acc16[0][1][0]), "+v"(acc16[0][1][1]),
                              "+v"(acc16[1][0][0]), "+v"(acc16[1][0][1]), "+v"(acc16[1][1][0]), "+v"(acc16[1][1][1]));
            __builtin_amdgcn_sched_barrier(0);
#define DVQ_RD(dst, S) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(tile_a), "i"((S) * 1024))
#define DVQ_MM4(src, F, WAIT, NEXT)                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" ::: "memory");                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            acc16[0][(F) / S32][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[0][0][(F) % S32], acc16[0][(F) / S32][0], 0, 0, 0); \
            acc16[0][(F) / S32][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[0][1][(F) % S32], acc16[0][(F) / S32][1], 0, 0, 0); \
            acc16[1][(F) / S32][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[1][0][(F) % S32], acc16[1][(F) / S32][0], 0, 0, 0); \
            acc16[1][(F) / S32][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[1][1][(F) % S32], acc16[1][(F) / S32][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            if ((F) + 4 < S16) { DVQ_RD(src, ((F) + 4 < S16 ? (F) + 4 : 0)); }                                             \
            NEXT
            DVQ_RD(a0, 0); DVQ_RD(a1, 1); DVQ_RD(a2, 2); DVQ_RD(a3, 3);
            __builtin_amdgcn_s_setprio(1);
            DVQ_MM4(a0, 0, 3, ) DVQ_MM4(a1, 1, 3, issue_piece(t + 3, 0);) DVQ_MM4(a2, 2, 3, ) DVQ_MM4(a3, 3, 3, )
            DVQ_MM4(a0, 4, 3, issue_piece(t + 3, 1);) DVQ_MM4(a1, 5, 3, ) DVQ_MM4(a2, 6, 3, ) DVQ_MM4(a3, 7, 3, issue_piece(t + 3, 2);)
            DVQ_MM4(a0, 8, 3, ) DVQ_MM4(a1, 9, 3, ) DVQ_MM4(a2, 10, 3, issue_piece(t + 3, 3);) DVQ_MM4(a3, 11, 3, )
            DVQ_MM4(a0, 12, 3, ) DVQ_MM4(a1, 13, 2, issue_piece(t + 3, 4);) DVQ_MM4(a2, 14, 1, ) DVQ_MM4(a3, 15, 0, )
#undef DVQ_MM4
#undef DVQ_RD
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const float om = b1[u][t2];
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const float v0 = acc16[u][r >> 2][t2][r & 3], v1 = acc16[u][(r + 1) >> 2][t2][(r + 1) & 3];
                        float g0 = __uint_as_float((__float_as_uint(v0) & 0xFFFFFFF0u) | (unsigned)r);
                        float g1 = __uint_as_float((__float_as_uint(v1) & 0xFFFFFFF0u) | (unsigned)(r + 1));
                        float md = __builtin_amdgcn_fmed3f(b1[u][t2], g0, g1);
                        b1[u][t2] = vmax3_raw(b1[u][t2], g0, g1);
                        b2[u][t2] = vmax_raw(b2[u][t2], md);
                    }
                    bt[u][t2] = (b1[u][t2] != om) ? t : bt[u][t2];
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // surplus ring DMA
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float rb[2], rs[2];
            int rc[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                float mb = b1[u][t2], ms = b2[u][t2];
                int mt = bt[u][t2], mq = q16;
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float o1 = __shfl_xor(mb, off), o2 = __shfl_xor(ms, off);
                    const int ot = __shfl_xor(mt, off), oq = __shfl_xor(mq, off);
                    const bool other_wins = (o1 > mb) || (o1 == mb && ((lane ^ off) < lane));
                    ms = fmaxf(other_wins ? mb : o1, fmaxf(ms, o2));
                    mb = other_wins ? o1 : mb;
                    mt = other_wins ? ot : mt;
                    mq = other_wins ? oq : mq;
                }
                const int r = (int)(__float_as_uint(mb) & 15u);
                rb[t2] = mb; rs[t2] = ms;
                rc[t2] = mt * 32 + 16 * (r >> 2) + 4 * mq + (r & 3);
            }
            const int srcl = c & 15;
            const float x0 = __shfl(rb[0], srcl), x1 = __shfl(rb[1], srcl);
            const float y0 = __shfl(rs[0], srcl), y1 = __shfl(rs[1], srcl);
            const int c0 = __shfl(rc[0], srcl), c1 = __shfl(rc[1], srcl);
            bestv[u] = (c >> 4) ? x1 : x0;
            secondv[u] = (c >> 4) ? y1 : y0;
            code[u] = (c >> 4) ? c1 : c0;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        thr[u] = bestv[u] - thr2W[u];
        const bool final_ok = (bestv[u] - secondv[u]) > thr2W[u];
        valid[u] = nn[u] >= 0;
        hopeless[u] = !(code[u] < K) || !(thr[u] == thr[u]);
        undecided[u] = valid[u] && !hopeless[u] && !final_ok;
    }
    const unsigned long long um0 = __ballot(undecided[0] && h == 0), um1 = __ballot(undecided[1] && h == 0);
    const int shard = blockIdx.x & (DVQ_QSHARDS - 1);
    int slot_raw = 0;
    const int nund = (int)__popcll(um0) + (int)__popcll(um1);
    if (nund != 0 && lane == 0) slot_raw = atomicAdd(&counters[DVQ_QCOUNT0 + shard], nund);
    int slot[2] = {-1, -1};
    if (nund != 0) {                                        // wave-uniform
        const int base = __shfl(slot_raw, 0);
        const unsigned long long lt = (1ull << c) - 1ull;
        slot[0] = undecided[0] ? base + (int)__popcll(um0 & lt) : -1;
        slot[1] = undecided[1] ? base + (int)__popcll(um0) + (int)__popcll(um1 & lt) : -1;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (slot[u] >= rec_cap) { hopeless[u] = true; slot[u] = -1; }     // shard full -> exact list
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (valid[u] && hopeless[u] && h == 0) {
            int pos = atomicAdd(&counters[DVQ_C_EXACT], 1);
            exact_list[pos] = nn[u];
        }

    // ---- epilogue per block: z again, chosen codebook row, z_q, loss term, record of a queued token
    float lsum = 0.0f;
    auto epilogue = [&](const int n, const bool active, const int cd, const size_t zb, const int sl,
                        const float xnu, const float thru) {
        if (!active) return;
        if (h == 0) codes[n] = (long long)cd;
        const float *zp = z + zb;
        const float *ep = E + (size_t)cd * D + 8 * h;
        const float m = (mask != nullptr) ? mask[n] : 1.0f;
        char *rec = (sl >= 0) ? records + ((size_t)shard * rec_cap + sl) * rec_bytes(D) : nullptr;
        auto finish = [&](auto store_tag) {
            constexpr bool STORE = decltype(store_tag)::value;
            float *zqp = STORE ? zq + zb : nullptr;
            const float *zpe = zp, *epe = ep;                // advance by two k-steps per batch
#pragma unroll
            for (int s0 = 0; s0 < S16; s0 += 2) {
                float zf[2][8];
                f32x4 eg[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) zf[q][j] = DVQ_LOAD_Z(zpe + (size_t)(16 * q + j) * HW);
                    eg[q][0] = *(const f32x4 *)(epe + 16 * q);
                    eg[q][1] = *(const f32x4 *)(epe + 16 * q + 4);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int s = s0 + q;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float e = eg[q][j >> 2][j & 3];
                        float diff = __fsub_rn(e, zf[q][j]);
                        if (STORE) DVQ_STORE_ZQ(zqp + (size_t)(16 * s + j) * HW, __fadd_rn(zf[q][j], diff));
                        lsum = __fadd_rn(lsum, __fmul_rn(__fmul_rn(diff, diff), m));
                    }
                    if (rec != nullptr) {
                        f32x4 lo = {zf[q][0], zf[q][1], zf[q][2], zf[q][3]};
                        f32x4 hi = {zf[q][4], zf[q][5], zf[q][6], zf[q][7]};
                        *(f32x4 *)(rec + (16 * s + 8 * h) * 4) = lo;
                        *(f32x4 *)(rec + (16 * s + 8 * h + 4) * 4) = hi;
                    }
                }
                zpe += (size_t)32 * HW;
                epe += 32;
                asm volatile("" : "+v"(zpe), "+v"(epe) : "v"(lsum));     // next batch's loads wait for this one
            }
        };
        if (zq != nullptr) finish(std::true_type{});
        else finish(std::false_type{});
        if (rec != nullptr && h == 0) {
            RecMeta rm;
            rm.n = n; rm.xn = xnu; rm.thr = thru; rm.m = m; rm.prov = cd;
            rm.best = ~0ull; rm.rep = 1;
            *(RecMeta *)(rec + (size_t)D * 4) = rm;
        }
    };
    epilogue(nn[0], valid[0] && !hopeless[0], code[0], zbase[0], slot[0], xn[0], thr[0]);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    epilogue(nn[1], valid[1] && !hopeless[1], code[1], zbase[1], slot[1], xn[1], thr[1]);
    if (partials != nullptr) {
        double dsum = (double)lsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        __syncthreads();
        double *red = (double *)lds;
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {                                     // this grid is half the standard one: fill both slots
            partials[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
            if (2 * (int)blockIdx.x + 1 < nparts_pass1) partials[2 * blockIdx.x + 1] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// resolver: queued tokens, RES_SLOTS (= 32, one MFMA column set) per workgroup.  The queue is short
// (a few % of the tokens), so the work is spread for LATENCY: the four waves of a workgroup share
// the same 32 tokens and each takes every fourth code tile, reading its A fragments straight from
// the L2-resident prep image (no LDS ring, no barrier in the loop).
// One caller, vq_resolve_kernel (a launch of its own behind pass 1; large codebooks: sliced code
// tiles, the wide / pipe forms of pass 1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long order_key(float d, int code)
{
    d = d + 0.0f;                                 // -0 -> +0: equal distances tie on the index
    unsigned u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone map, finite d
    return ((unsigned long long)u << 32) | (unsigned)code;
}

#ifndef DVQ_RES_WAVES
#define DVQ_RES_WAVES 4          // waves per resolver workgroup (they split the code tiles)
#endif
// ATen-order sum of squares of v[0 .. D) (oracle/dvq_oracle.c: dvq_oracle_sumsq; D a multiple of 32)
__device__ __forceinline__ float aten_sumsq(const float *v, int D)
{
    float a[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) a[m] = 0.0f;
    for (int k0 = 0; k0 < D; k0 += 32)
#pragma unroll
        for (int m = 0; m < 32; ++m) a[m] = __fadd_rn(a[m], sq_rn(v[k0 + m]));
    float s = 0.0f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float tl = __fadd_rn(__fadd_rn(__fadd_rn(a[l], a[l + 8]), a[l + 16]), a[l + 24]);
        s = (l == 0) ? tl : __fadd_rn(s, tl);
    }
    return s;
}

// LDS of one resolver workgroup (statics of vq_resolve_kernel, a carve of pass 1's dynamic region for the consumers)
template <int D>
struct ResLds {
    static constexpr int RB = D * 4 + 32;                    // bytes per record (rec_bytes(D))
    static constexpr int SREC = 0;                           // [RES_SLOTS][RB] this workgroup's records
    static constexpr int CAND = SREC + RES_SLOTS * RB;       // [RES_CAND] unsigned
    static constexpr int BEST = CAND + RES_CAND * 4;         // [RES_SLOTS] u64
    static constexpr int REWR = BEST + RES_SLOTS * 8;        // [RES_SLOTS] int
    static constexpr int MISC = REWR + RES_SLOTS * 4;        // [8] int: 0 candidate count, 1 rewrite count, 2 last slice, 3 overflow flag,
                                                             //          4 live slots of the chunk, 5 chunk is the shard's last
    static constexpr int RED = MISC + 32;                    // [DVQ_RES_WAVES] double
    static constexpr int BYTES = RED + 8 * DVQ_RES_WAVES;
    static_assert(CAND % 16 == 0 && BEST % 8 == 0 && RED % 8 == 0, "carve alignment");
};

#ifndef DVQ_FOLD_ABL
#define DVQ_FOLD_ABL 0           // timing experiments of the tuning build only (results WRONG): 1 no conv loop, 2 no xn, 4 no h write-back
#endif

// One chunk: the records [base, base + nlive) (record indices; nlive <= RES_SLOTS), code tiles [t_begin, t_end).
// FOLD (vq_fold.hip): the records hold the conv's INPUT x and `img` / `meta` are the folded codebook: the enumeration below
// runs on x exactly as pass 1 scored it (its candidate set contains the reference's winner for every h inside the conv's
// tolerance); the workgroup then computes h = W x + bias for its 32 tokens -- qconv.hip's split-fp16 arithmetic, bit-identical
// to dvq_qconv_f32 -- in place over x, and the exact chains / the rewrite run on that h against the codebook itself.
// nslice > 1 (large codebooks): each slice resolves its candidates locally, merges its per-token best into the record with a
// 64-bit atomicMin, and the slice that arrives last at the chunk's ticket carries on (and puts the ticket back to zero).
// Returns (thread 0) the chunk's loss correction; *not_last is set for a slice that is not the chunk's last.
// HW = positions per image of the OUTPUT grid; a routed token (RecMeta.rep > 1) covers rep x rep positions, rows Wout apart,
// all rewritten with the same values.
template <int D, bool FOLD>
__device__ __forceinline__ double resolve_chunk(
    char *__restrict__ L, const int base, const int nlive, const int t_begin, const int t_end,
    const char *__restrict__ img, const float *__restrict__ en_all, const float *__restrict__ E, int HW, int Wout,
    float *__restrict__ zq, long long *__restrict__ codes, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int nslice, int *__restrict__ ticket, float *__restrict__ h_spill, const DvqConv &cv,
    const bool want_loss, bool *not_last)
{
    // h_spill (conv fused into pass 1; null otherwise): [B, D, HW] buffer the exact-list kernel reads its tokens' latents from.
    // Pass 1 spills the rows of ITS hand-offs; the tokens the resolver itself sends to that list (candidate overflow, no
    // candidate) get their row written here, from the record (which holds the conv's output).
    using LL = ResLds<D>;
    constexpr int S16 = D / 16;
    constexpr int IMG_BYTES = S16 * 1024;
    constexpr int TILE_STRIDE = IMG_BYTES + 256;
    constexpr int RW = DVQ_RES_WAVES;
    constexpr int RB = LL::RB;
    char *srec = L + LL::SREC;
    unsigned *cand = (unsigned *)(L + LL::CAND);
    unsigned long long *best = (unsigned long long *)(L + LL::BEST);
    int *rewrite = (int *)(L + LL::REWR);
    int *misc = (int *)(L + LL::MISC);
    double *red = (double *)(L + LL::RED);

    int tid = threadIdx.x, lane = tid & 63;                  // (re-derived after the enumeration loop, see there)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c = lane & 31, h = lane >> 5;
    *not_last = false;
    if (tid < RES_SLOTS) best[tid] = ~0ull;
    if (tid < 4) misc[tid] = 0;
    {
        // all of a thread's pieces are in flight before the first is stored (the rolled loop paid one memory latency per piece:
        // 4.3 of a chunk's 13 us)
        constexpr int NP = (RES_SLOTS * (RB / 16) + RW * 64 - 1) / (RW * 64);
        const int npieces = nlive * (RB / 16);
        f32x4 tmp[NP];
        const f32x4 *src = (const f32x4 *)(records + (size_t)base * RB);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * RW * 64;
            tmp[k] = src[i < npieces ? i : 0];
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * RW * 64;
            if (i < npieces) ((f32x4 *)srec)[i] = tmp[k];
        }
    }
    __syncthreads();
    DVQ_RSTAMP(2);
    const bool live = c < nlive;
    const char *rec = srec + (live ? c : 0) * RB;            // (FOLD: taken again after the enumeration loop)
    f16x8 zh[S16];
#pragma unroll
    for (int s = 0; s < S16; ++s) {                    // same RNE f32 -> f16 conversion as pass 1
        const f32x4 lo = *(const f32x4 *)(rec + (16 * s + 8 * h) * 4);
        const f32x4 hi = *(const f32x4 *)(rec + (16 * s + 8 * h + 4) * 4);
        u32x4 packed;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            f32x2 vv = {j2 < 2 ? lo[2 * j2] : hi[2 * j2 - 4], j2 < 2 ? lo[2 * j2 + 1] : hi[2 * j2 - 3]};
            f16x2 hh = __builtin_convertvector(vv, f16x2);
            packed[j2] = __builtin_bit_cast(unsigned, hh);
        }
        zh[s] = __builtin_bit_cast(f16x8, packed);
    }
    const RecMeta rm = *(const RecMeta *)(rec + (size_t)D * 4);
    const float thr = live ? rm.thr : __builtin_inff();
    __syncthreads();
    // ---- enumerate: every code whose approximate score reaches best - 2W.  The A fragments of a tile come
    // straight from L2 (16 KiB per tile) and a workgroup is one latency chain (about one workgroup per CU is
    // active), so the next tile's sixteen loads are in flight while this tile's MFMAs run (two fragment sets;
    // same-box A/B: 31.9 -> 28.3 us at configs[2]; 64 tokens per workgroup instead: 35.5 us).
    {
        auto fetch = [&](int t, f16x8 (&a)[S16], f32x4 (&en4)[4]) {
            const char *tile = img + (size_t)t * TILE_STRIDE;
            const float *enr = (const float *)(tile + IMG_BYTES) + 4 * h;      // accumulator seeds of the tile
#pragma unroll
            for (int s = 0; s < S16; ++s) a[s] = *(const f16x8 *)(tile + s * 1024 + lane * 16);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) en4[g4] = *(const f32x4 *)(enr + 8 * g4);   // rows 8g + 4h + q
        };
        auto score = [&](int t, const f16x8 (&a)[S16], const f32x4 (&en4)[4]) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < S16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], zh[s], acc, 0, 0, 0);
            unsigned hits = 0;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float g = acc[4 * g4 + q] + en4[g4][q];                            // padding: -3e38
                    hits |= (g >= thr) ? (1u << (4 * g4 + q)) : 0u;
                }
            while (hits) {
                int r = __builtin_ctz(hits);
                hits &= hits - 1;
                int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                int pos = atomicAdd(&misc[0], 1);
                if (pos < RES_CAND) cand[pos] = ((unsigned)c << 20) | (unsigned)code;
            }
        };
        // Two tiles in flight per wave.  Every fetch is UNCONDITIONAL (past the wave's last tile it re-reads that tile and the
        // result is not scored): with `if (t + RW < t_end) fetch(..)` the wait in front of a tile's first MFMA was computed over
        // both paths -- vmcnt(5): the just-issued prefetch had to land too, so no tile was ever fetched under another's MFMAs
        // (0.85 us per tile on an idle chip instead of the MFMA chain's 0.25).
        f16x8 a0[S16], a1[S16];
        f32x4 e0[4], e1[4];
        const int tw = t_begin + wave;
        if (tw < t_end) {
            const int tl = tw + (t_end - 1 - tw) / RW * RW;      // this wave's last tile
            auto clampt = [&](int t) { return t < tl ? t : tl; };
            // (sched_barrier: left alone the scheduler interleaves the two fetches, and the register the first MFMA needs is
            // then among the last loads issued)
            fetch(tw, a0, e0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(clampt(tw + RW), a1, e1);
            __builtin_amdgcn_sched_barrier(0);
            for (int t = tw; t < t_end; t += 2 * RW) {
                score(t, a0, e0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(clampt(t + 2 * RW), a0, e0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + RW < t_end) score(t + RW, a1, e1);
                __builtin_amdgcn_sched_barrier(0);
                fetch(clampt(t + 3 * RW), a1, e1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the thread's ids from mbcnt instead of the registers that held them: kept live across the enumeration loop (256 registers,
    // every one in use) they were this kernel's spills (2 - 7 dwords of scratch per lane)
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    tid = wave * 64 + lane;
    c = lane & 31;
    h = lane >> 5;
    if constexpr (FOLD) {
        const char *rec = srec + (c < nlive ? c : 0) * RB;       // (shadows the first: the same address from the fresh ids)
        constexpr int T8 = D / 32, NT = (T8 + RW - 1) / RW;      // row tiles of the weight; this wave takes wave, wave + RW, ..
        constexpr int QIMG = S16 * 1024, QTILE = 2 * QIMG + 256;
        float amax = 0.0f;
#pragma unroll
        for (int s = 0; s < S16; ++s) {
            const f32x4 lo = *(const f32x4 *)(rec + (16 * s + 8 * h) * 4);
            const f32x4 hi = *(const f32x4 *)(rec + (16 * s + 8 * h + 4) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { amax = vmax_abs(amax, lo[j]); amax = vmax_abs(amax, hi[j]); }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        int ea = 0;                                              // per-token power-of-two scale, as qconv_kernel
        if (amax > 0.0f && amax < __builtin_inff()) { int e; (void)frexpf(amax, &e); ea = 14 - e; }
        ea = ea > 100 ? 100 : (ea < -100 ? -100 : ea);
        const float sa = ldexpf(1.0f, ea);
        const float unscale = ldexpf(cv.meta->inv_scale_w, -ea);
        // branch-free: a wave whose tile index runs past the last tile (D = 64: waves 2, 3) recomputes the last one and
        // writes the same values again.  The weight fragments come straight from L2, SB k-steps (SB * NT * 2 loads of 16 B per
        // lane) in flight at a time -- with a conditional per tile hipcc waited for every single load (12 us of the kernel).
        constexpr int SB = S16 < 8 ? S16 : 8;
        int tile_of[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) tile_of[i] = (wave + i * RW < T8) ? wave + i * RW : T8 - 1;
        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
        for (int s0 = 0; s0 < ((DVQ_FOLD_ABL & 1) ? 0 : S16); s0 += SB) {
            f16x8 wh[SB][NT], wl[SB][NT];
#pragma unroll
            for (int q = 0; q < SB; ++q)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const char *wt = cv.wimg + (size_t)tile_of[i] * QTILE + (s0 + q) * 1024 + lane * 16;
                    wh[q][i] = *(const f16x8 *)wt;
                    wl[q][i] = *(const f16x8 *)(wt + QIMG);
                }
            __builtin_amdgcn_sched_barrier(0);                   // all of the batch's loads are issued before its first MFMA
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const int s = s0 + q;
                const f32x4 lo = *(const f32x4 *)(rec + (16 * s + 8 * h) * 4);
                const f32x4 hi = *(const f32x4 *)(rec + (16 * s + 8 * h + 4) * 4);
                u32x4 ph, pl;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const float v0 = (j2 < 2 ? lo[2 * j2] : hi[2 * j2 - 4]) * sa, v1 = (j2 < 2 ? lo[2 * j2 + 1] : hi[2 * j2 - 3]) * sa;
                    const f32x2 vv = {v0, v1};
                    const f16x2 hh = __builtin_convertvector(vv, f16x2);
                    const f32x2 rr = {v0 - (float)hh[0], v1 - (float)hh[1]};
                    const f16x2 ll = __builtin_convertvector(rr, f16x2);
                    ph[j2] = __builtin_bit_cast(unsigned, hh);
                    pl[j2] = __builtin_bit_cast(unsigned, ll);
                }
                const f16x8 xh = __builtin_bit_cast(f16x8, ph), xl = __builtin_bit_cast(f16x8, pl);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[q][i], xh, acc[i], 0, 0, 0);     // small terms first (qconv.hip)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[q][i], xl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[q][i], xh, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                         // every wave has read x: h takes its place
        if (live && !(DVQ_FOLD_ABL & 4)) {
            float *hrow = (float *)(srec + c * RB);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t8 = tile_of[i];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * t8 + 16 * (r >> 3) + 8 * h + (r & 7);     // qconv_row_channel's inverse
                    hrow[ch] = __builtin_fmaf(acc[i][r], unscale, cv.bias[ch]);
                }
            }
        }
        __syncthreads();
        if (!(DVQ_FOLD_ABL & 2)) {
            // the reference's norm of h (ATen order: 32 partial sums a[i % 32], ((a[l] + a[l+8]) + a[l+16]) + a[l+24], then l = 0..7
            // left to right) for the exact chains: 8 lanes per token, lane l owns a[l], a[l+8], a[l+16], a[l+24]
            static_assert(RW * 64 >= RES_SLOTS * 8, "8 lanes per queued token");
            const int tk = (tid >> 3) < RES_SLOTS ? (tid >> 3) : RES_SLOTS - 1, l8 = tid & 7;   // (threads past the last token: idle repeats)
            const float *hv = (const float *)(srec + tk * RB);
            float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int k0 = 0; k0 < D; k0 += 32)
#pragma unroll
                for (int g = 0; g < 4; ++g) a4[g] = __fadd_rn(a4[g], sq_rn(hv[k0 + l8 + 8 * g]));
            const float tl = __fadd_rn(__fadd_rn(__fadd_rn(a4[0], a4[1]), a4[2]), a4[3]);
            float sn = __shfl(tl, lane & ~7);
#pragma unroll
            for (int i = 1; i < 8; ++i) sn = __fadd_rn(sn, __shfl(tl, (lane & ~7) + i));
            if (l8 == 0 && (tid >> 3) < nlive) ((RecMeta *)(srec + tk * RB + (size_t)D * 4))->xn = sn;
        }
        __syncthreads();
    }
    __syncthreads();
    const int ncand_raw = misc[0];
    bool overflow = ncand_raw > RES_CAND;             // hand the whole group to the exact list
    const int ncand = overflow ? 0 : ncand_raw;

    DVQ_RSTAMP(3);
    // ---- exact chains: one thread per (token, candidate)
    for (int i = tid; i < ncand; i += RW * 64) {
        const unsigned pc = cand[i];
        const int sl = (int)(pc >> 20), code = (int)(pc & 0xFFFFFu);
        const char *r2 = srec + sl * RB;
        const f32x4 *zv = (const f32x4 *)r2;
        const f32x4 *ev = (const f32x4 *)(E + (size_t)code * D);
        const float xn = ((const RecMeta *)(r2 + (size_t)D * 4))->xn;
        float acc = 0.0f;
        // the codebook row in batches of 16 x 16 B, the next batch in flight while this one feeds the (sequential) chain
        constexpr int CB = 16, NB = D / 4 / CB;
        f32x4 eb[2][CB];
#pragma unroll
        for (int q = 0; q < CB; ++q) eb[0][q] = ev[q];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) {
#pragma unroll
                for (int q = 0; q < CB; ++q) eb[(b + 1) & 1][q] = ev[(b + 1) * CB + q];
            }
#pragma unroll
            for (int q = 0; q < CB; ++q) {
                const f32x4 a = zv[b * CB + q], bb = eb[b & 1][q];
                acc = __builtin_fmaf(a[0], bb[0], acc);
                acc = __builtin_fmaf(a[1], bb[1], acc);
                acc = __builtin_fmaf(a[2], bb[2], acc);
                acc = __builtin_fmaf(a[3], bb[3], acc);
            }
        }
        float bias = __fadd_rn(xn, en_all[code]);
        float d = __builtin_fmaf(-2.0f, acc, bias);
        atomicMin(&best[sl], order_key(d, code));
    }
    __syncthreads();

    {
        if (nslice > 1) {
            // merge across slices through the records; the last slice of this chunk carries on
            int *oflag = ticket + 1;
            if (tid < nlive && best[tid] != ~0ull) {
                RecMeta *gm = (RecMeta *)(records + (size_t)(base + tid) * rec_bytes(D) + (size_t)D * 4);
                atomicMin(&gm->best, best[tid]);
            }
            if (overflow && tid == 0) atomicOr(oflag, 1);
            // everything handed over is an agent-scope atomic (and read back with agent-scope loads): what the ticket needs is
            // that every wave's atomics have COMPLETED before it is taken -- a counter wait per wave and the barrier, no cache
            // write-back / invalidate (a __threadfence() pair here cost 1.5 - 3 us per slice on a small batch)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                misc[2] = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nslice - 1);
            __syncthreads();
            if (!misc[2]) { *not_last = true; return 0.0; }   // not the last slice (its partial is written by the last)
            if (tid < nlive) {
                const RecMeta *gm = (const RecMeta *)(records + (size_t)(base + tid) * rec_bytes(D) + (size_t)D * 4);
                best[tid] = __hip_atomic_load(&gm->best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) {
                misc[3] = __hip_atomic_load(oflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // every slice has been here: the
                __hip_atomic_store(ticket + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // pair is clean for the next op
            }
            __syncthreads();
            overflow = misc[3] != 0;
        }
    }

    DVQ_RSTAMP(4);
    // ---- winners; slots whose winner differs from pass 1 are rewritten
    if (tid < nlive) {
        const char *r2 = srec + tid * RB;
        const RecMeta m2 = *(const RecMeta *)(r2 + (size_t)D * 4);
        if (overflow || best[tid] == ~0ull) {
            for (int ry = 0; ry < m2.rep; ++ry)               // cannot resolve here: full exact evaluation of every position
                for (int rx = 0; rx < m2.rep; ++rx) {         // the token stands for; pass 1's loss terms are taken back below
                    int pos = atomicAdd(&counters[DVQ_C_EXACT], 1);
                    exact_list[pos] = m2.n + ry * Wout + rx;
                }
            int pos = atomicAdd(&misc[1], 1);
            rewrite[pos] = (tid << 20) | 0xFFFFF;
        } else {
            int win = (int)(best[tid] & 0xFFFFFFFFu);
            if (win != m2.prov) {
                int pos = atomicAdd(&misc[1], 1);
                rewrite[pos] = (tid << 20) | win;
            }
        }
    }
    __syncthreads();
    const int nrew = misc[1];
    double dsum = 0.0;
    for (int i = wave; i < nrew; i += RW) {           // one wave per rewritten token, 4 channels per lane
        const int sl = rewrite[i] >> 20, win = rewrite[i] & 0xFFFFF;
        const bool take_back_only = win == 0xFFFFF;   // token went to the exact list
        const char *r2 = srec + sl * RB;
        const RecMeta m2 = *(const RecMeta *)(r2 + (size_t)D * 4);
        const long n = m2.n;
        const int rep = m2.rep;
        const long bimg = n / HW;
        const int hw = (int)(n - bimg * HW);
        const float m = m2.m;
        float delta = 0.0f;
        for (int k0 = lane * 4; k0 < D; k0 += 256) {
            f32x4 zv = *(const f32x4 *)(r2 + k0 * 4);
            if (take_back_only && h_spill != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float *hp = h_spill + ((size_t)bimg * D + k0 + j) * HW + hw;
                    for (int ry = 0; ry < rep; ++ry)
                        for (int rx = 0; rx < rep; ++rx) hp[(size_t)ry * Wout + rx] = zv[j];
                }
            }
            f32x4 eo = *(const f32x4 *)(E + (size_t)m2.prov * D + k0);
            f32x4 en_ = take_back_only ? eo : *(const f32x4 *)(E + (size_t)win * D + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dn = __fsub_rn(en_[j], zv[j]), dold = __fsub_rn(eo[j], zv[j]);
                if (zq != nullptr && !take_back_only) {
                    float *zp = zq + ((size_t)bimg * D + k0 + j) * HW + hw;
                    // FOLD: z_q := e[code] as in pass 1 (the record holds h by now; fl(h + fl(e - h)) is within 1e-6 of it)
                    const float v = FOLD ? en_[j] : __fadd_rn(zv[j], dn);
                    for (int ry = 0; ry < rep; ++ry)
                        for (int rx = 0; rx < rep; ++rx) zp[(size_t)ry * Wout + rx] = v;
                }
                float tn = take_back_only ? 0.0f : __fmul_rn(__fmul_rn(dn, dn), m);
                delta += tn - __fmul_rn(__fmul_rn(dold, dold), m);
            }
        }
        if (lane == 0 && !take_back_only)
            for (int ry = 0; ry < rep; ++ry)
                for (int rx = 0; rx < rep; ++rx) codes[n + (long)ry * Wout + rx] = (long long)win;
        delta *= (float)(rep * rep);
        dsum += (double)delta;
    }
    DVQ_RSTAMP(5);
    double tot = 0.0;
    if (want_loss) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 0; w < RW; ++w) tot += red[w];
        }
    }
    return tot;
}

template <int D, bool FOLD>
__global__ __launch_bounds__(DVQ_RES_WAVES * 64, DVQ_RES_WAVES > 4 ? 1 : 2) void vq_resolve_kernel(
    const char *__restrict__ img, const DvqF16Meta *__restrict__ meta, const float *__restrict__ en_all,
    const float *__restrict__ E, int HW, int K,
    float *__restrict__ zq, long long *__restrict__ codes, double *__restrict__ partials,
    int *__restrict__ counters, int *__restrict__ exact_list, char *__restrict__ records, int rec_cap,
    int nslice, int *__restrict__ chunk_sync, int Wout, float *__restrict__ h_spill, const DvqConv cv,
    const double *__restrict__ p1_partials, int np1)
{
    __shared__ __attribute__((aligned(16))) char L[ResLds<D>::BYTES];
    (void)meta;
    DVQ_RSTAMP(0);
    // The workgroup that writes chunk blockIdx.x's partial (also for an empty chunk) folds its share of pass 1's per-block loss
    // sums into it, in a fixed order: the list kernel's finishing workgroup then adds gridDim.x numbers instead of np1 more.
    auto p1_share = [&]() -> double {
        const int per = (np1 + (int)gridDim.x - 1) / (int)gridDim.x;
        const int i0 = (int)blockIdx.x * per, i1 = (i0 + per < np1) ? i0 + per : np1;
        double a = 0.0;
        for (int i = i0; i < i1; ++i) a += p1_partials[i];
        return a;
    };
    // block -> (shard, chunk): the first DVQ_QSHARDS blocks take chunk 0 of every shard, and so on
    const int shard = blockIdx.x & (DVQ_QSHARDS - 1), chunk = blockIdx.x / DVQ_QSHARDS;
    int total = counters[DVQ_QCOUNT0 + shard];
    total = total < rec_cap ? total : rec_cap;
    total += shard * rec_cap;                                   // end of this shard's filled run
    const int base = shard * rec_cap + chunk * RES_SLOTS;
    const int slice = blockIdx.y;
    if (base >= total) {
        if (partials != nullptr && threadIdx.x == 0 && slice == 0) partials[blockIdx.x] = p1_share();
        return;
    }
    const int T = dvq_num_tiles(K);
    int tps = (T + nslice - 1) / nslice;                        // tiles per slice, a multiple of 4
    tps = (tps + DVQ_RES_WAVES - 1) / DVQ_RES_WAVES * DVQ_RES_WAVES;
    const int t_begin = slice * tps, t_end = (t_begin + tps < T) ? t_begin + tps : T;
    const int nlive = (total - base < RES_SLOTS) ? total - base : RES_SLOTS;
    DVQ_RSTAMP(1);
    bool not_last;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // (a scalar: thread 0 is found again below without
    const double tot = resolve_chunk<D, FOLD>(L, base, nlive, t_begin, t_end, img, en_all, E, HW, Wout, zq, codes, counters,   // keeping threadIdx.x in a register)
                                              exact_list, records, nslice, chunk_sync + 2 * blockIdx.x, h_spill, cv,
                                              partials != nullptr, &not_last);
    const bool thread0 = wave0 == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0;
    if (!not_last && partials != nullptr && thread0) partials[blockIdx.x] = tot + p1_share();
    DVQ_RSTAMP(6);
}

template <int D, int SEL, bool CONV, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted rv, const DvqConv cv)
{
    pass1_body<D, SEL, CONV, FOLD, true>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, rec_cap, rv, cv);
}

// row-major latents [N, D] (FLAT, see pass1_body)
template <int D, bool FOLD>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_flat_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted rv, const DvqConv cv)
{
    pass1_body<D, 0, false, FOLD, true, true>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, rec_cap, rv, cv);
}

// small batches: `ksplit` workgroups per token block, each on its slice of the code tiles (SPLIT, see pass1_body)
template <int D, bool FLAT>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_split_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, f32x4 *__restrict__ split, int ksplit)
{
    const DvqRouted rv = {};
    const DvqConv cv = {};
    pass1_body<D, 0, false, false, false, FLAT, true>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records,
                                                      rec_cap, rv, cv, split, ksplit);
}

// ... with the router select fused in (per-lane form; every slice's workgroup writes the same indices / codebook_mask / gate)
template <int D>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_split_sel_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted rv, f32x4 *__restrict__ split, int ksplit)
{
    const DvqConv cv = {};
    pass1_body<D, 1, false, false, false, false, true>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records,
                                                       rec_cap, rv, cv, split, ksplit);
}

// ... and with the 1x1 conv as the prologue (every slice's workgroup computes the block's h itself: 3 x 8.4 MFLOP, nothing to share)
template <int D, int SEL>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_split_conv_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted rv, const DvqConv cv, f32x4 *__restrict__ split, int ksplit)
{
    pass1_body<D, SEL, true, false, true, false, true>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records,
                                                       rec_cap, rv, cv, split, ksplit);
}

// ... and on the conv-folded codebook (loss-free inference / stage-2 tokenisation of single images)
template <int D, int SEL>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_split_fold_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted rv, const DvqConv cv, f32x4 *__restrict__ split, int ksplit)
{
    pass1_body<D, SEL, false, true, false, false, true>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records,
                                                        rec_cap, rv, cv, split, ksplit);
}

// the same kernel with plain loads of the latents, for batches that fit the memory-side cache (dense or staged select, no conv)
template <int D, int SEL, bool FOLD>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_cached_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap, const DvqRouted rv, const DvqConv cv)
{
    pass1_body<D, SEL, false, FOLD, false>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, rec_cap, rv, cv);
}

// ---------------------------------------------------------------------------------------------
// audit aid (dvq_debug_filter_scores_f32, tools/bound_audit.py): pass 1's score arithmetic on a few tokens
// given as rows [n, D] -- same fp16 conversion, same seeded accumulator, same MFMA chain in the same order,
// same index packing, same threshold -- with every score written out instead of reduced to a top-2.
// ---------------------------------------------------------------------------------------------
// audit aid (dvq_debug_filter_scores_f32, tools/bound_audit.py): pass 1's score arithmetic on a few tokens
// given as rows [n, D] -- same fp16 conversion, same seeded accumulator, same MFMA chain in the same order
// (v_mfma_f32_16x16x32_f16 over tile image "16"), same index packing, same threshold -- with every score
// written out instead of reduced to a top-2.  One wave per 32 tokens; A fragments straight from the prep image.
// (The tuning build can also dump best / second / 2W of the PRODUCTION kernel per token: g_dvq_tokdbg.)
// ---------------------------------------------------------------------------------------------
template <int D, bool FOLD>
__global__ __launch_bounds__(64) void filter_scores_debug_kernel(
    const float *__restrict__ tokens, int n, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    int K, float *__restrict__ G, float *__restrict__ thr2W_out, float *__restrict__ xn_out)
{
    constexpr int S16 = D / 16;
    constexpr int S32 = D / 32;
    constexpr int IMG_BYTES = S16 * 1024;
    constexpr int TILE_STRIDE = IMG_BYTES + 256;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int tok = blockIdx.x * 32 + c;
    const bool valid = tok < n;
    const float *zp = tokens + (size_t)(valid ? tok : n - 1) * D + 8 * h;
    const int T = dvq_num_tiles(K), Kpad = 32 * T;
    const float sB = meta->scale_b;
    float pa[2][8];
    float amax = 0.0f, zeta2 = 0.0f;
#pragma unroll
    for (int s = 0; s < S16; ++s) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            const float v0 = zp[16 * s + 2 * j2], v1 = zp[16 * s + 2 * j2 + 1];
            const float q0 = sq_rn(v0), q1 = sq_rn(v1);
            pa[s & 1][2 * j2] = (s < 2) ? q0 : __fadd_rn(pa[s & 1][2 * j2], q0);
            pa[s & 1][2 * j2 + 1] = (s < 2) ? q1 : __fadd_rn(pa[s & 1][2 * j2 + 1], q1);
            amax = vmax_abs(amax, v0);
            amax = vmax_abs(amax, v1);
            f32x2 vv = {v0, v1};
            f16x2 hh = __builtin_convertvector(vv, f16x2);
            const float r0 = v0 - (float)hh[0], r1 = v1 - (float)hh[1];
            zeta2 = __builtin_fmaf(r0, r0, zeta2);
            zeta2 = __builtin_fmaf(r1, r1, zeta2);
        }
    }
    float t8[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float o0 = __shfl_xor(pa[0][l], 32), o1 = __shfl_xor(pa[1][l], 32);
        float a0_ = h == 0 ? pa[0][l] : o0, a1_ = h == 0 ? o0 : pa[0][l];
        float a2_ = h == 0 ? pa[1][l] : o1, a3_ = h == 0 ? o1 : pa[1][l];
        t8[l] = __fadd_rn(__fadd_rn(__fadd_rn(a0_, a1_), a2_), a3_);
    }
    float xn = t8[0];
#pragma unroll
    for (int l = 1; l < 8; ++l) xn = __fadd_rn(xn, t8[l]);
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    zeta2 += __shfl_xor(zeta2, 32);
    const float thr2W = FOLD ? dvq_fold_threshold(xn, amax, zeta2, sB, (const DvqFoldMeta *)meta)
                             : dvq_filter_threshold(xn, amax, zeta2, sB, meta);
    if (valid && h == 0) { thr2W_out[tok] = thr2W; xn_out[tok] = xn; }
    // the 16x16x32 code loop of pass 1: lane (c16, q) holds tokens c16 / 16 + c16 of the block, k = 32 s' + 8 q + j --
    // the same fp16 values pass 1 permutes into this order
    const int c16 = lane & 15, q16 = lane >> 4;
    f16x8 zb[2][S32];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        const int tk = blockIdx.x * 32 + 16 * t2 + c16;
        const float *zq_ = tokens + (size_t)(tk < n ? tk : n - 1) * D + 8 * q16;
#pragma unroll
        for (int sp = 0; sp < S32; ++sp) {
            u32x4 pk;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                f32x2 vv = {zq_[32 * sp + 2 * j2], zq_[32 * sp + 2 * j2 + 1]};
                f16x2 hh = __builtin_convertvector(vv, f16x2);
                pk[j2] = __builtin_bit_cast(unsigned, hh);
            }
            zb[t2][sp] = __builtin_bit_cast(f16x8, pk);
        }
    }
    for (int t = 0; t < T; ++t) {
        const char *tile = img + (size_t)t * TILE_STRIDE;
        const float *seeds = (const float *)(tile + IMG_BYTES) + 4 * q16;
        f32x4 acc16[2][2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const f32x4 e4 = *(const f32x4 *)(seeds + 16 * c2);
            acc16[c2][0] = e4;
            acc16[c2][1] = e4;
        }
#pragma unroll
        for (int F = 0; F < 2 * S32; ++F) {
            const f16x8 a = *(const f16x8 *)(tile + F * 1024 + lane * 16);
            acc16[F / S32][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, zb[0][F % S32], acc16[F / S32][0], 0, 0, 0);
            acc16[F / S32][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, zb[1][F % S32], acc16[F / S32][1], 0, 0, 0);
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int tk = blockIdx.x * 32 + 16 * t2 + c16;
            if (tk < n) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int code = t * 32 + 16 * (r >> 2) + 4 * q16 + (r & 3);
                    G[(size_t)tk * Kpad + code] = __uint_as_float((__float_as_uint(acc16[r >> 2][t2][r & 3]) & 0xFFFFFFF0u) | (unsigned)r);
                }
            }
        }
    }
}

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// the 16x16x32-order image follows the 32x32x16-order one (which the resolver reads)
static size_t dvq_img16_offset(int K, int D)
{
    const size_t tile = (size_t)(D / 16) * 1024 + 256;
    return ((size_t)dvq_num_tiles(K) * tile + 255) / 256 * 256;
}

// fold != 0: `prep` is the buffer of dvq_fold_prepare_f32, `tokens` the conv's inputs
int dvq_launch_filter_scores_debug(const float *tokens, int n, const void *prep, int D, int K, float *G,
                                   float *thr2W, float *xn, float *scale_b_out, hipStream_t st, int fold)
{
    char *base = (char *)prep + dvq_prep_f16_offset(K, D);
    base = (char *)(((uintptr_t)base + 255) / 256 * 256);
    if (fold) base = (char *)prep;
    const DvqF16Meta *meta = (const DvqF16Meta *)base;
    const char *im = base + 256 + dvq_img16_offset(K, D);
    const int blocks = (n + 31) / 32;
    if (fold) {
        switch (D) {
        case 64:  hipLaunchKernelGGL((filter_scores_debug_kernel<64, true>), dim3(blocks), dim3(64), 0, st, tokens, n, im, meta, K, G, thr2W, xn); break;
        case 128: hipLaunchKernelGGL((filter_scores_debug_kernel<128, true>), dim3(blocks), dim3(64), 0, st, tokens, n, im, meta, K, G, thr2W, xn); break;
        case 256: hipLaunchKernelGGL((filter_scores_debug_kernel<256, true>), dim3(blocks), dim3(64), 0, st, tokens, n, im, meta, K, G, thr2W, xn); break;
        default:  return -1000;
        }
    } else
    switch (D) {
    case 64:  hipLaunchKernelGGL((filter_scores_debug_kernel<64, false>), dim3(blocks), dim3(64), 0, st, tokens, n, im, meta, K, G, thr2W, xn); break;
    case 128: hipLaunchKernelGGL((filter_scores_debug_kernel<128, false>), dim3(blocks), dim3(64), 0, st, tokens, n, im, meta, K, G, thr2W, xn); break;
    case 256: hipLaunchKernelGGL((filter_scores_debug_kernel<256, false>), dim3(blocks), dim3(64), 0, st, tokens, n, im, meta, K, G, thr2W, xn); break;
    default:  return -1000;
    }
    if (scale_b_out != nullptr)
        (void)hipMemcpyAsync(scale_b_out, &meta->scale_b, sizeof(float), hipMemcpyDeviceToDevice, st);
    return (int)hipGetLastError();
}

// The op's counter block (dvq_common.h: DVQ_C_*) and the sliced resolver's chunk tickets.  In the steady state nobody launches
// this: every filter-path op puts its live words back to zero itself -- the list kernel's finishing workgroup the counter block,
// each chunk's last resolver slice its ticket pair -- and a caller that keeps track says so with DVQ_MODE_WS_CLEAN (dvq.h).
__global__ void zero_counters_kernel(int *__restrict__ counters, int nwords)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) counters[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int dvq_launch_exact_list(const float *z, const float *prep, const float *E, const float *mask,
                          int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                          const int *list, const int *list_count, DvqLossTail tail, const DvqRouted *rv,
                          hipStream_t st, const DvqConv *fold_conv = nullptr);

// Launch-time choices of the filter path.  They are compile-time constants of the production library; only the
// tuning build (-DDVQ_TUNING: libdvq_tuning.so, tools/) can change them, through dvq_tuning_set().
struct DvqTune {
    int sel_staged;      // routed op on a 32-wide output grid: coarser branches through LDS (SEL = 2) instead of per-lane loads
    int res_slices;      // resolver slices over the code tiles, 0 = by codebook size
    int flat;            // HW == 1 (row-major [N, D]) through the row-major form of pass 1 (0: through the NCHW kernel, for the A/B)
    int split;           // small batches: several workgroups per token block (SPLIT form of pass 1); 0: never (for the A/B)
};
#ifdef DVQ_TUNING
static DvqTune g_tune = {1, 0, 1, 1};
extern "C" __attribute__((visibility("default"))) int dvq_tuning_set(const char *key, int value)
{
    if (!strcmp(key, "sel_staged")) g_tune.sel_staged = value;
    else if (!strcmp(key, "res_slices")) g_tune.res_slices = value;
    else if (!strcmp(key, "flat")) g_tune.flat = value;
    else if (!strcmp(key, "split")) g_tune.split = value;
    else return -1;
    return 0;
}
// device buffer the tuning build's pass 1 writes its per-token diagnostics to (null = off): tokdbg [N][4] f32 = best, second,
// 2W, code; stamps [workgroup][8] u64 = stage times of the split form of pass 1 (null = off)
extern "C" __attribute__((visibility("default"))) int dvq_tuning_buffers(void *stamps, void *tokdbg)
{
    int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dvq_stamps), &stamps, sizeof(void *));
    if (rc) return rc;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dvq_tokdbg), &tokdbg, sizeof(void *));
}
#else
static constexpr DvqTune g_tune = {1, 0, 1, 1};
#endif

static bool dvq_flat_form_enabled() { return g_tune.flat != 0; }

// slots per shard (a multiple of RES_SLOTS); the whole record area holds DVQ_QSHARDS times that
static int shard_capacity(long N)
{
    long cap = N / 8;
    if (cap < 4096) cap = 4096;
    long per = (cap + DVQ_QSHARDS - 1) / DVQ_QSHARDS;
    per = (per + RES_SLOTS - 1) / RES_SLOTS * RES_SLOTS;
    return (int)per;
}

static int rec_capacity(long N) { return DVQ_QSHARDS * shard_capacity(N); }

bool dvq_filter_supported(int D, int HW, int K, long N)
{
    // D * HW < 2^29: pass 1 addresses a wave's loads / stores as a 32-bit byte offset from the wave's first token (buffer instructions)
    return (D == 64 || D == 128 || D == 256) && N < (1L << 31) && K < (1 << 20) && (long)D * HW < (1L << 29);
}

// resolver slices over the code tiles: 1 up to 64 tiles (K <= 2048), then one per 64 tiles, at most 8
static int resolver_slices(int K)
{
    if (g_tune.res_slices >= 1 && g_tune.res_slices <= 8) return g_tune.res_slices;
    int T = dvq_num_tiles(K);
    int ns = (T + 63) / 64;
    return ns < 1 ? 1 : (ns > 8 ? 8 : ns);
}

// ws_extra: [counters DVQ_COUNTER_BYTES][chunk ticket + overflow flag: 2 ints per resolver chunk]
//           [exact list N ints][records cap * rec_bytes]
// the split form of pass 1 (small batches): slices per token block, 1 = not taken.  As many as keep the grid within one workgroup
// per CU (256) and leave a slice two code tiles, at most 8.
static int split_slices(int K, long N)
{
    const long nb = (N + 127) / 128;
    if (!g_tune.split || nb > DVQ_SPLIT_MAX_BLOCKS) return 1;
    int ks = DVQ_SPLIT_MAX_SLICES;
    while (ks > 1 && (nb * ks > 256 || dvq_num_tiles(K) / ks < 2)) ks >>= 1;
    return ks;
}
static size_t split_bytes(long N)
{
    const long nb = (N + 127) / 128;
    return nb <= DVQ_SPLIT_MAX_BLOCKS ? align256((size_t)nb * DVQ_SPLIT_MAX_SLICES * 128 * sizeof(f32x4)) : 0;
}

size_t dvq_filter_ws_extra_bytes(int D, int HW, int K, long N)
{
    (void)HW; (void)K;
    return DVQ_COUNTER_BYTES + align256((size_t)rec_capacity(N) / RES_SLOTS * 2 * sizeof(int)) +
           align256((size_t)N * sizeof(int)) + align256((size_t)rec_capacity(N) * rec_bytes(D)) + split_bytes(N);
}

int dvq_launch_prep_f16(const float *E, int K, int D, void *prep, hipStream_t st)
{
    char *base = (char *)prep + dvq_prep_f16_offset(K, D);
    base = (char *)(((uintptr_t)base + 255) / 256 * 256);
    DvqF16Meta *meta = (DvqF16Meta *)base;
    char *img = base + 256;
    const float *en_all = (const float *)((char *)prep + dvq_prep_en_offset(K, D));
    const int T = dvq_num_tiles(K);
    char *img16 = img + dvq_img16_offset(K, D);
    if (dvq_prep_codes_per_workgroup(K) == 8)
        hipLaunchKernelGGL(codebook_prep_f16_kernel<8>, dim3(T * 4), dim3(256), 8 * D * sizeof(float), st, E, K, D, (const float *)prep, en_all, meta, img, img16);
    else
        hipLaunchKernelGGL(codebook_prep_f16_kernel<32>, dim3(T), dim3(256), 32 * D * sizeof(float), st, E, K, D, (const float *)prep, en_all, meta, img, img16);
    return (int)hipGetLastError();
}

// partials layout: [pass 1: np1 = its grid][resolver: cap/RES_SLOTS][exact list: min(ceil(N/128), DVQ_EXACT_LIST_BLOCKS)]
static int list_blocks(long N)
{
    long nb = (N + 127) / 128;
    return (int)(nb < DVQ_EXACT_LIST_BLOCKS ? nb : DVQ_EXACT_LIST_BLOCKS);
}

int dvq_launch_routed_prepass(int G, int gate_mode, const void *gate, float thr, int B, int hc, int wc,
                              long long *indices, float *cmask, long long *gate_out, hipStream_t st);

struct FilterWs {
    int *counters, *chunk_sync, *exact_list;
    char *records;
    int cap;
    f32x4 *split;                                            // [token block][slice < 8][128] of the split form (small batches), else null
};

static FilterWs carve_ws(void *ws_extra, long N, int D)
{
    FilterWs w;
    w.counters = (int *)ws_extra;
    w.cap = rec_capacity(N);
    w.chunk_sync = (int *)((char *)ws_extra + DVQ_COUNTER_BYTES);
    const size_t sync_bytes = align256((size_t)w.cap / RES_SLOTS * 2 * sizeof(int));
    w.exact_list = (int *)((char *)ws_extra + DVQ_COUNTER_BYTES + sync_bytes);
    w.records = (char *)ws_extra + DVQ_COUNTER_BYTES + sync_bytes + align256((size_t)N * sizeof(int));
    w.split = split_bytes(N) ? (f32x4 *)(w.records + align256((size_t)w.cap * rec_bytes(D))) : nullptr;
    return w;
}

// SEL = 2 applies to the reference's grids: output rows of 32 positions, whole workgroups of 4 rows per image, and
// branch tensors the 16-byte DMA pieces can address
static bool staged_select_ok(const DvqRouted &rv)
{
    if (!g_tune.sel_staged || rv.Wout != 32 || rv.HWout % 128 != 0) return false;
    for (int g = 0; g < rv.G; ++g)
        if (((uintptr_t)rv.src[g] & 15) != 0) return false;
    return true;
}

template <int D, int SEL, bool CONV = false, bool FOLD = false>
static int launch_pass1_form(const float *z, const char *img16, const DvqF16Meta *meta, const float *E,
                             const float *mask, int HW, int K, long N, float *zq, long long *codes,
                             double *partials, const FilterWs &w, const DvqRouted &rv, hipStream_t st,
                             const DvqConv &cv = DvqConv{})
{
    static unsigned long long done = 0;
    const size_t shmem1 = 4 * (size_t)(D / 16) * 1024 + 4 * 4 * 64 * sizeof(float) + 4 * 2048;
    const unsigned grid = (unsigned)((N + 127) / 128);
    if constexpr (SEL == 0 && !CONV && !FOLD) {
        // fewer token blocks than CUs: several workgroups per block, each on a slice of the code tiles
        const int ks = split_slices(K, N);
        if (ks > 1 && w.split != nullptr) {
            const bool flat = HW == 1 && dvq_flat_form_enabled() && (((uintptr_t)z | (uintptr_t)zq) & 15) == 0;
            static unsigned long long done_s = 0, done_sf = 0;
            int rcs = flat ? dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_kernel<D, true>, (int)shmem1, &done_sf)
                           : dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_kernel<D, false>, (int)shmem1, &done_s);
            if (rcs) return rcs;
            if (flat)
                hipLaunchKernelGGL((vq_assign_filter_split_kernel<D, true>), dim3(grid * ks), dim3(256), shmem1, st,
                                   z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                                   w.cap / DVQ_QSHARDS, w.split, ks);
            else
                hipLaunchKernelGGL((vq_assign_filter_split_kernel<D, false>), dim3(grid * ks), dim3(256), shmem1, st,
                                   z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                                   w.cap / DVQ_QSHARDS, w.split, ks);
            return (int)hipGetLastError();
        }
    }
    if constexpr (SEL == 0 && !CONV) {
        // HW == 1 is a row-major [N, D] tensor: 16-byte accesses along a token's row (rows are 16-byte aligned: D % 16 == 0)
        if (HW == 1 && dvq_flat_form_enabled() && (((uintptr_t)z | (uintptr_t)zq) & 15) == 0) {
            static unsigned long long done_f = 0;
            int rcf = dvq_allow_dynamic_lds((const void *)vq_assign_filter_flat_kernel<D, FOLD>, (int)shmem1, &done_f);
            if (rcf) return rcf;
            hipLaunchKernelGGL((vq_assign_filter_flat_kernel<D, FOLD>), dim3(grid), dim3(256), shmem1, st,
                               z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                               w.cap / DVQ_QSHARDS, rv, cv);
            return (int)hipGetLastError();
        }
    }
    if constexpr (D == 256 && !CONV && SEL != 1) {
        // a batch whose features fit the memory-side cache (with room for what else is live): plain loads instead of non-temporal ones
        if ((size_t)N * D * sizeof(float) <= DVQ_CACHED_MAX_BYTES) {
            static unsigned long long done_c = 0;
            int rcc = dvq_allow_dynamic_lds((const void *)vq_assign_filter_cached_kernel<D, SEL, FOLD>, (int)shmem1, &done_c);
            if (rcc) return rcc;
            hipLaunchKernelGGL((vq_assign_filter_cached_kernel<D, SEL, FOLD>), dim3(grid), dim3(256), shmem1, st,
                               z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                               w.cap / DVQ_QSHARDS, rv, cv);
            return (int)hipGetLastError();
        }
    }
    int rc = dvq_allow_dynamic_lds((const void *)vq_assign_filter_kernel<D, SEL, CONV, FOLD>, (int)shmem1, &done);
    if (rc) return rc;
    hipLaunchKernelGGL((vq_assign_filter_kernel<D, SEL, CONV, FOLD>), dim3(grid), dim3(256), shmem1, st,
                       z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                       w.cap / DVQ_QSHARDS, rv, cv);
    return (int)hipGetLastError();
}

template <int D>
static int launch_pass1(const float *z, const char *img, const DvqF16Meta *meta, const float *E,
                        const float *mask, int HW, int K, long N, float *zq, long long *codes,
                        double *partials, const FilterWs &w, bool force_wide, const DvqRouted *rv,
                        hipStream_t st, const DvqConv *cv, const DvqConv *fold_cv)
{
    const int nb1 = (int)((N + 127) / 128);
    const char *img16 = img + dvq_img16_offset(K, D);       // the code loop runs on v_mfma_f32_16x16x32_f16
    const DvqRouted none = {};
    const DvqConv nocv = {};
    if (fold_cv != nullptr) {                                // img / meta: the folded codebook; z (or the branches): the conv's input
        if (const int ks = split_slices(K, N); ks > 1 && w.split != nullptr) {   // small batch: several workgroups per token block
            static unsigned long long done_f0 = 0, done_f1 = 0;
            const size_t shm = 4 * (size_t)(D / 16) * 1024 + 4 * 4 * 64 * sizeof(float) + 4 * 2048;
            int rcs = rv != nullptr ? dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_fold_kernel<D, 1>, (int)shm, &done_f1)
                                    : dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_fold_kernel<D, 0>, (int)shm, &done_f0);
            if (rcs) return rcs;
            if (rv != nullptr)
                hipLaunchKernelGGL((vq_assign_filter_split_fold_kernel<D, 1>), dim3((unsigned)nb1 * ks), dim3(256), shm, st,
                                   z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                                   w.cap / DVQ_QSHARDS, *rv, *fold_cv, w.split, ks);
            else
                hipLaunchKernelGGL((vq_assign_filter_split_fold_kernel<D, 0>), dim3((unsigned)nb1 * ks), dim3(256), shm, st,
                                   z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                                   w.cap / DVQ_QSHARDS, none, *fold_cv, w.split, ks);
            return (int)hipGetLastError();
        }
        if (rv == nullptr)
            return launch_pass1_form<D, 0, false, true>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, none, st, *fold_cv);
        if (staged_select_ok(*rv))
            return launch_pass1_form<D, 2, false, true>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, *rv, st, *fold_cv);
        return launch_pass1_form<D, 1, false, true>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, *rv, st, *fold_cv);
    }
    if (cv != nullptr) {                                     // the 1x1 conv as the prologue (D = 256; the ABI layer checked)
        if constexpr (D == 256) {
            if (const int ks = split_slices(K, N); ks > 1 && w.split != nullptr) {   // small batch: several workgroups per token block
                static unsigned long long done_c0 = 0, done_c1 = 0;
                const size_t shm = 4 * (size_t)(D / 16) * 1024 + 4 * 4 * 64 * sizeof(float) + 4 * 2048;
                int rcs = rv != nullptr ? dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_conv_kernel<D, 1>, (int)shm, &done_c1)
                                        : dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_conv_kernel<D, 0>, (int)shm, &done_c0);
                if (rcs) return rcs;
                if (rv != nullptr)
                    hipLaunchKernelGGL((vq_assign_filter_split_conv_kernel<D, 1>), dim3((unsigned)nb1 * ks), dim3(256), shm, st,
                                       z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                                       w.cap / DVQ_QSHARDS, *rv, *cv, w.split, ks);
                else
                    hipLaunchKernelGGL((vq_assign_filter_split_conv_kernel<D, 0>), dim3((unsigned)nb1 * ks), dim3(256), shm, st,
                                       z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                                       w.cap / DVQ_QSHARDS, none, *cv, w.split, ks);
                return (int)hipGetLastError();
            }
            if (rv != nullptr)
                return launch_pass1_form<D, 1, true>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, *rv, st, *cv);
            return launch_pass1_form<D, 0, true>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, none, st, *cv);
        } else {
            return -1000;
        }
    }
    if (rv != nullptr) {                                     // select fused in
        if (const int ks = split_slices(K, N); ks > 1 && w.split != nullptr) {   // small batch: several workgroups per token block
            static unsigned long long done_ss = 0;
            const size_t shm = 4 * (size_t)(D / 16) * 1024 + 4 * 4 * 64 * sizeof(float) + 4 * 2048;
            int rcs = dvq_allow_dynamic_lds((const void *)vq_assign_filter_split_sel_kernel<D>, (int)shm, &done_ss);
            if (rcs) return rcs;
            hipLaunchKernelGGL((vq_assign_filter_split_sel_kernel<D>), dim3((unsigned)nb1 * ks), dim3(256), shm, st,
                               z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list, w.records,
                               w.cap / DVQ_QSHARDS, *rv, w.split, ks);
            return (int)hipGetLastError();
        }
        if (staged_select_ok(*rv))
            return launch_pass1_form<D, 2>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, *rv, st, nocv);
        return launch_pass1_form<D, 1>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, *rv, st, nocv);
    }
    if constexpr (D == 256) {
        if (force_wide || (K >= DVQ_WIDE_MIN_K && N >= 256L * 512)) {   // large codebook and enough tokens to fill every CU
                                                             // with two 256-token workgroups: two blocks per wave
            static unsigned long long done_w = 0;
            const unsigned gridw = (unsigned)((N + 255) / 256);
            const size_t shm = 4 * (size_t)(D / 16) * 1024 + 4 * 4 * 64 * sizeof(float) + 4 * 2048;
            int rc = dvq_allow_dynamic_lds((const void *)vq_assign_filter_wide_kernel<D>, (int)shm, &done_w);
            if (rc) return rc;
            hipLaunchKernelGGL((vq_assign_filter_wide_kernel<D>), dim3(gridw), dim3(256), shm, st,
                               z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w.counters, w.exact_list,
                               w.records, w.cap / DVQ_QSHARDS, nb1);
            return (int)hipGetLastError();
        }
    }
    return launch_pass1_form<D, 0>(z, img16, meta, E, mask, HW, K, N, zq, codes, partials, w, none, st, nocv);
}

template <int D>
static int launch_resolver(const char *img, const DvqF16Meta *meta, const float *en_all, const float *E,
                           int HWout, int K, float *zq, long long *codes, double *partials,
                           const FilterWs &w, int Wout, float *h_spill, const DvqFold *fd, hipStream_t st,
                           const double *p1_partials, int np1)
{
    const int nslice = resolver_slices(K);
    if (fd != nullptr)
        hipLaunchKernelGGL((vq_resolve_kernel<D, true>), dim3(w.cap / RES_SLOTS, nslice), dim3(DVQ_RES_WAVES * 64), 0, st, img,
                           meta, en_all, E, HWout, K, zq, codes, partials, w.counters, w.exact_list, w.records,
                           w.cap / DVQ_QSHARDS, nslice, w.chunk_sync, Wout, h_spill, fd->cv, p1_partials, np1);
    else
        hipLaunchKernelGGL((vq_resolve_kernel<D, false>), dim3(w.cap / RES_SLOTS, nslice), dim3(DVQ_RES_WAVES * 64), 0, st, img,
                           meta, en_all, E, HWout, K, zq, codes, partials, w.counters, w.exact_list, w.records,
                           w.cap / DVQ_QSHARDS, nslice, w.chunk_sync, Wout, h_spill, DvqConv{}, p1_partials, np1);
    return (int)hipGetLastError();
}

static int launch_resolver_d(int D, const char *img, const DvqF16Meta *meta, const float *en_all, const float *E,
                             int HWout, int K, float *zq, long long *codes, double *partials,
                             const FilterWs &w, int Wout, float *h_spill, const DvqFold *fd, hipStream_t st,
                             const double *p1_partials, int np1)
{
    switch (D) {
    case 64:  return launch_resolver<64>(img, meta, en_all, E, HWout, K, zq, codes, partials, w, Wout, h_spill, fd, st, p1_partials, np1);
    case 128: return launch_resolver<128>(img, meta, en_all, E, HWout, K, zq, codes, partials, w, Wout, h_spill, fd, st, p1_partials, np1);
    case 256: return launch_resolver<256>(img, meta, en_all, E, HWout, K, zq, codes, partials, w, Wout, h_spill, fd, st, p1_partials, np1);
    default:  return -1000;
    }
}

// Dense op: z [B, D, HW].  Routed op (rv != nullptr): one token per output position of rv (the select fused into
// pass 1); N = B * HWout, mask = the codebook_mask pass 1 writes.
// Kernels of one op: [zero kernel unless ws_clean] -> pass 1 -> resolver -> list kernel (exact list, loss finalize, and its
// finishing workgroup puts the counter block back to zero: the op leaves its workspace clean).
int dvq_launch_filter(const float *z, const void *prep, const float *E, const float *mask,
                      int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                      void *ws_extra, bool pass1_only, bool force_wide, float *loss, float beta,
                      const DvqRouted *rv, hipStream_t st, const DvqConv *cv, const DvqFold *fd, bool ws_clean)
{
    char *base = (char *)prep + dvq_prep_f16_offset(K, D);
    base = (char *)(((uintptr_t)base + 255) / 256 * 256);
    // fd: pass 1 and the resolver's enumeration run on the folded codebook (same section layout); the exact chains, the
    // gathers and the exact-list kernel on the codebook itself
    const DvqF16Meta *meta = (fd != nullptr) ? (const DvqF16Meta *)fd->fprep : (const DvqF16Meta *)base;
    const char *img = (fd != nullptr) ? fd->fprep + 256 : base + 256;
    const float *en_all = (const float *)((const char *)prep + dvq_prep_en_offset(K, D));
    const FilterWs w = carve_ws(ws_extra, N, D);
    const bool routed = rv != nullptr;
    int rc;
    if (!ws_clean) {
        // A kernel rather than hipMemsetAsync: cheaper than the runtime's fill kernel, and the op stays a pure chain of
        // kernel nodes under hipGraph capture.
        const int nwords = DVQ_COUNTER_BYTES / 4 + (resolver_slices(K) > 1 ? w.cap / RES_SLOTS * 2 : 0);
        hipLaunchKernelGGL(zero_counters_kernel, dim3(nwords > 4096 ? 8 : 1), dim3(256), 0, st, w.counters, nwords);
        rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    const int np1 = (int)((N + 127) / 128);
    const DvqConv *fold_cv = fd ? &fd->cv : nullptr;
    switch (D) {
    case 64:  rc = launch_pass1<64>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, w, force_wide, rv, st, cv, fold_cv); break;
    case 128: rc = launch_pass1<128>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, w, force_wide, rv, st, cv, fold_cv); break;
    case 256: rc = launch_pass1<256>(z, img, meta, E, mask, HW, K, N, zq, codes, partials, w, force_wide, rv, st, cv, fold_cv); break;
    default:  return -1000;
    }
    if (rc || pass1_only) return rc;
    const int HWout = routed ? rv->HWout : HW, Wout = routed ? rv->Wout : 0;
    rc = launch_resolver_d(D, img, meta, en_all, E, HWout, K, zq, codes, partials ? partials + np1 : nullptr,
                           w, Wout, nullptr, fd, st, partials, np1);
    if (rc) return rc;
    double *partials3 = partials ? partials + np1 + w.cap / RES_SLOTS : nullptr;
    // the list kernel is the last of the op: it also sums the partials into loss[0..1] and cleans the counter block; the
    // resolver's partials already contain pass 1's (vq_resolve_kernel: p1_share)
    const DvqLossTail tail = {partials ? loss : nullptr, partials ? partials + np1 : nullptr, w.counters + DVQ_C_TICKET,
                              w.cap / RES_SLOTS + list_blocks(N),
                              1.0 / ((double)N * D), beta, w.counters, w.cap / DVQ_QSHARDS};
    const int *list_count = w.counters + DVQ_C_EXACT;
    // conv folded in: the list kernel computes its tokens' h itself, from the conv's input (dense z or the branches)
    if (fd != nullptr)
        return dvq_launch_exact_list(z, (const float *)prep, E, mask, D, HWout, K, N, zq, codes, partials3,
                                     w.exact_list, list_count, tail, rv, st, &fd->cv);
    // conv fused into pass 1: the same -- the list kernel's conv is qconv.hip's arithmetic, what pass 1's prologue computes too (and
    // with h_all it writes the h it scored over pass 1's row of the token)
    if (cv != nullptr)
        return dvq_launch_exact_list(z, (const float *)prep, E, mask, D, HWout, K, N, zq, codes, partials3,
                                     w.exact_list, list_count, tail, rv, st, cv);
    return dvq_launch_exact_list(z, (const float *)prep, E, mask, D, HWout, K, N, zq, codes, partials3,
                                 w.exact_list, list_count, tail, rv, st);
}

// ---- routed op ---------------------------------------------------------------------------------
// filter mode: zero counters -> pass 1 with the select fused in (it derives the grain of every position's cell from the
// gate and writes indices / codebook_mask / gate_out itself: no prepass, no tables) -> resolver -> list + loss finalize.
// exact mode: a prepass writes indices / codebook_mask / gate_out, then every position by the exact chain.
int dvq_launch_exact(const float *z, const float *prep, const float *E, const float *mask,
                     int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                     const DvqRouted *rv, hipStream_t st);
int dvq_launch_loss_finalize(const double *partials, int nparts, double inv_numel, float beta,
                             float *loss, hipStream_t st);

int dvq_launch_routed(int G, int gate_mode, const void *gate, float thr, const float *h_coarse,
                      const float *h_median, const float *h_fine, const void *prep, const float *E,
                      int B, int D, int hc, int wc, int K, float beta, float *zq, long long *codes,
                      float *loss, long long *indices, float *cmask, long long *gate_out,
                      double *partials, void *ws_extra, bool exact, bool pass1_only, hipStream_t st, const DvqConv *cv,
                      const DvqFold *fd, bool ws_clean)
{
    const int SC = (G == 2) ? 2 : 4;
    const int Wout = SC * wc, HWout = SC * hc * Wout;
    const long N = (long)B * HWout;
    int rc = 0;
    if (exact) {
        rc = dvq_launch_routed_prepass(G, gate_mode, gate, thr, B, hc, wc, indices, cmask, gate_out, st);
        if (rc) return rc;
    }
    DvqRouted rv{};
    rv.G = G; rv.B = B; rv.D = D; rv.hc = hc; rv.wc = wc; rv.Wout = Wout; rv.HWout = HWout;
    rv.indices = indices; rv.gate = gate; rv.gate_mode = gate_mode; rv.thr = thr;
    if (!exact) { rv.indices_out = indices; rv.cmask_out = cmask; rv.gate_out = gate_out; }
    if (G == 2) {
        rv.src[0] = h_coarse; rv.src[1] = h_fine; rv.src[2] = nullptr;
        rv.sub[0] = 1; rv.sub[1] = 2; rv.sub[2] = 1;
        rv.rep[0] = 2; rv.rep[1] = 1; rv.rep[2] = 1;
    } else {
        rv.src[0] = h_coarse; rv.src[1] = h_median; rv.src[2] = h_fine;
        rv.sub[0] = 1; rv.sub[1] = 2; rv.sub[2] = 4;
        rv.rep[0] = 4; rv.rep[1] = 2; rv.rep[2] = 1;
    }
    if (exact) {
        rc = dvq_launch_exact(nullptr, (const float *)prep, E, cmask, D, HWout, K, N, zq, codes, partials, &rv, st);
        if (rc || loss == nullptr) return rc;
        return dvq_launch_loss_finalize(partials, (int)((N + 127) / 128), 1.0 / ((double)N * D), beta, loss, st);
    }
    return dvq_launch_filter(nullptr, prep, E, cmask, D, HWout, K, N, zq, codes, partials, ws_extra, pass1_only, false,
                             loss, beta, &rv, st, cv, fd, ws_clean);
}
