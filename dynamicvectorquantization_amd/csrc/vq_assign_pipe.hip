// vq_assign_pipe.hip -- pass 1 of the filter path as a persistent, role-alternating kernel ("pipe" form).
//
// Why: in vq_assign_filter_kernel the three phases of a workgroup (prologue: HBM reads, code loop: matrix cores, epilogue:
// HBM writes) run one after the other, and every resident workgroup is in the same phase at the same time.  The in-kernel
// stamps of round 3 (profiles/r03_pass1_loop_ablation.json) showed that the code loop is bound by each wave's own
// instruction stream, about 1330 cycles per code tile of which the matrix pipe is busy 512 -- and that the five LDS-DMA
// issues a wave makes per tile for the codebook ring cost about 310 of them.  So here
//   * ONE workgroup of 8 waves per CU, persistent over a contiguous range of 128-token blocks, split in two groups of four
//     waves that take turns: while group A runs the code loop of its block, group B writes the z_q of its previous block
//     and loads / converts its next one -- the HBM phases of one block run beside the matrix phase of another in EVERY
//     phase, by construction;
//   * the computing group feeds the codebook ring itself (five LDS-DMA pieces per wave and tile, between its MFMAs; the
//     last three tiles it issues in a phase are the first three of the OTHER group's next phase), so the memory role has no
//     ring duty.  (First form of round 3: the memory role issued all ring DMA -- same time, see STATUS.)
//   * the two groups meet at one s_barrier per code tile (32 per phase), which is also the ring's hand-shake.
// Same per-token arithmetic as vq_assign_filter_kernel (same fp16 conversion, same seeds, same 16x16x32 MFMA chain, same
// top-2 and bound, same queue / records / exact list), hence the same bits out; resolver and list kernel are unchanged.
//
// Scope: D = 256, 32 code tiles (992 < K <= 1024: every reference config), HW % 128 == 0, dense z (SEL 0) or the router
// select fused in on a 32-wide output grid (SEL 2 dual, SEL 3 triple; coarser branches staged through LDS as in
// vq_assign_filter_kernel<D, 2>).  Everything else takes vq_assign_filter_kernel.
//
// Memory-role bookkeeping: the z loads, codebook-row gathers and z_q stores of the memory role are inline asm, issued
// unconditionally for the wave (never inside a lane-dependent branch) in a fixed per-step pattern per (EPI, PRO, STORE, SEL)
// variant, and waited for with counts computed at compile time from that pattern (struct Pat): vector-memory operations
// retire in issue order (one counter, vmcnt), so "wait until at most n younger operations are outstanding" is exact.  hipcc
// would otherwise wait vmcnt(0) at the first use of any ordinary load while an LDS-DMA is in flight; the few ordinary loads
// (gate, queue-slot atomic) sit at the start of a phase.
// STATUS (round 3): bit-exact (tests/test_pipe_form.py), NOT faster than vq_assign_filter_kernel -- 220 vs 203 us (dense)
// and 248 vs 220 us (select fused) at B = 256, in both forms (ring DMA by the memory role: 217-237 / 272).  The ablations
// and cycle accumulators below (DVQ_ABLATE, tools/pipe_probe.py, profiles/r03_pipe_form_ablation.json) show why: the two
// waves of a SIMD share ONE issue stream (one instruction per four cycles), so a phase costs the SUM of the computing wave's
// instructions (12.4 us), the ring DMA issues (6.4 us wherever they sit) and the memory role's (6.6 us) = 25 us -- the time
// the product kernel needs for the same work; counted memory waits are < 2 % of a phase.  Kept in the TUNING build only
// (dvq_tuning_set("pipe", 1)) as the measured negative result DESIGN.md section 5.1 cites.
#include "dvq_filter.h"
#ifdef DVQ_TUNING
#include <type_traits>
#ifndef DVQ_PIPE_PRIO
#define DVQ_PIPE_PRIO 1   // 1: the computing waves raise their priority over the MFMA section; 0: no priorities; 2: the memory role is the high-priority one
#endif
#ifndef DVQ_ABLATE
#define DVQ_ABLATE 0   // timing experiments (WRONG results except 1024): 128 z loads hit one cached line; 256 z_q stores hit one line; 512 no
#endif                 // conversion arithmetic; 1024 cycle accumulators -> stamp slot 4p + 3; 2048 no gathers; 4096 no stores; 8192 no loss math

namespace {

constexpr int PD = 256;
constexpr int PS16 = PD / 16, PS32 = PD / 32, PT = 32;
constexpr int PIMG = PS16 * 1024, PTILE = PIMG + 256, PNBUF = 4;
// LDS map (bytes)
constexpr int L_RING = 0;                       // 4 x 16 KiB code tiles
constexpr int L_SEEDS = PNBUF * PIMG;           // [4 slots][4 issuing waves][64 floats]
constexpr int L_SCR = L_SEEDS + PNBUF * 4 * 256;   // per-wave fragment permutation scratch, 8 x 2 KiB
constexpr int L_IMGA = L_SCR + 8 * 2048;        // 2x-coarser branch of the block being loaded [256][32 floats] (SEL 2 / 3)
constexpr int L_IMGB = L_IMGA + PD * 128;       // 4x-coarser branch [256][8 floats] (SEL 3)
constexpr int L_RED = L_IMGB + PD * 32;         // 8 doubles
constexpr int L_TOTAL = L_RED + 64;

// per-step vector-memory operation pattern of a memory-role wave (see header)
template <bool EPI, bool PRO, bool STORE, int SEL>
struct Pat {
    static constexpr int NSTAGE = (SEL == 2) ? 8 : ((SEL == 3) ? 10 : 0);   // staging DMA instructions per wave (dual: 8, triple: 8 + 2)
    static constexpr int pre() { return EPI ? 4 : 0; }                      // gathers of k-steps 0, 1
    static constexpr int gathers(int t) { return (EPI && t <= 13) ? 2 : 0; }
    static constexpr int stores(int t) { return (EPI && STORE && t < 16) ? 8 : 0; }
    static constexpr int loads(int t) { return (PRO && t >= 16 && t < 24) ? 16 : 0; }
    static constexpr int stage(int t) { return (PRO && t == 16) ? NSTAGE : 0; }
    // order inside a step: gathers, stores | staging DMA, loads   (the ring DMA is the COMPUTING group's: see compute_phase)
    static constexpr int ops(int t) { return gathers(t) + stores(t) + stage(t) + loads(t); }
    static constexpr int base(int t) { int n = pre(); for (int i = 0; i < t; ++i) n += ops(i); return n; }   // ops issued before step t
    static constexpr int cap(int n) { return n > 63 ? 63 : (n < 0 ? 0 : n); }
    // steps 0 and 1 of a memory phase that follows this wave's compute phase: the ring pieces it issued in the last two steps
    // of that phase (tiles 1 and 2 of THIS phase; tile 2's five pieces are younger than tile 1's) must have landed before it
    // arrives at barriers 1 and 2
    static constexpr int lead_wait(int t) { return cap((t == 0 ? 5 : 0) + base(t + 1)); }
    // step u, before using the gathers of k-step u (issued in step u - 2 right after its DMA; k-steps 0, 1 in the pre-step);
    // the wait sits after step u's DMA and gathers
    static constexpr int gather_wait(int u)
    {
        const int issued_before_wait = base(u) + gathers(u);
        const int last = (u < 2) ? (2 * u + 2) : (base(u - 2) + 2);           // index after the last gather of k-step u
        return cap(issued_before_wait - last);
    }
    // step 24 + j, before converting k-steps 2j, 2j + 1 (loaded in step 16 + j)
    static constexpr int load_wait(int j) { return cap(base(24 + j) - base(17 + j)); }
};

struct PipeArgs {
    const float *z;            // dense source [B, D, HW] (SEL 0)
    const char *img;           // tile image "16" of the prep buffer
    const DvqF16Meta *meta;
    const float *E;
    const float *mask;         // [B, HW] or null (SEL 0)
    int HW, K;
    long N;
    float *zq;
    long long *codes;
    double *partials;          // [npart] or null: entry blockIdx.x gets this workgroup's loss sum, the others 0
    int npart;
    int *counters, *exact_list;
    char *records;
    int rec_cap;
    int per_wg;                // 128-token blocks per workgroup
    DvqRouted rv;
};

}  // namespace

#ifdef DVQ_TUNING
// tuning build: s_memrealtime (100 MHz) of lane 0 of each group's first wave at every phase start and at steps 16 / 24 of every
// memory phase: [grid][2 groups][64] u64, slot = 4 * (phase + 1) + {0 start, 1 step 16, 2 step 24}; null = off
__device__ unsigned long long *g_pipe_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) int dvq_tuning_pipe_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_stamps), &p, sizeof(void *)); }
#define PIPE_STAMP(SLOT) do { if (g_pipe_stamps != nullptr && w4 == 0 && lane == 0 && (SLOT) < 64) \
    g_pipe_stamps[((size_t)blockIdx.x * 2 + g) * 64 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PIPE_STAMPV(SLOT, V) do { if (g_pipe_stamps != nullptr && w4 == 0 && lane == 0 && (SLOT) < 64) \
    g_pipe_stamps[((size_t)blockIdx.x * 2 + g) * 64 + (SLOT)] = (long long)(V); } while (0)
#else
#define PIPE_STAMP(SLOT) do { } while (0)
#define PIPE_STAMPV(SLOT, V) do { } while (0)
#endif

template <int SEL, bool STORE>
__global__ __launch_bounds__(512, 2) void vq_assign_pipe_kernel(const PipeArgs a)
{
    [[maybe_unused]] int phase_no = 0;                                        // (stamps of the tuning build only)
    constexpr int D = PD, S16 = PS16, S32 = PS32, T = PT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *enraw = (float *)(lds + L_SEEDS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, w4 = wave & 3;                  // group, wave within the group
    const int c = lane & 31, h = lane >> 5;
    const int q16 = lane >> 4;
    char *scr = lds + L_SCR + wave * 2048;
    const int HW = a.HW;
    const float sB = a.meta->scale_b;
    const int nblk = (int)(a.N >> 7);
    const int b_lo = blockIdx.x * a.per_wg;
    const int b_hi = (b_lo + a.per_wg < nblk) ? b_lo + a.per_wg : nblk;
    const int nmine = (b_hi - b_lo > g) ? (b_hi - b_lo - g + 1) / 2 : 0;    // blocks of this group: b_lo + g, b_lo + g + 2, ...
    const int nother = (b_hi - b_lo > 1 - g) ? (b_hi - b_lo - (1 - g) + 1) / 2 : 0;
    const int R = (b_hi - b_lo + 1) / 2;                     // rounds
    const int bpi = HW >> 7;                                 // blocks per image

    // ---- per-block state of this wave (lane = token (c), channel half (h))
    float zf[S16][8];
    f16x8 zb[2][S32];
    float xn = 0.0f, thr2W = 0.0f, sel_mask = 1.0f;
    float best = 0.0f, second = 0.0f;
    int code = 0;
    int tok_n = 0;                                           // token index of this lane in the block it holds (b*HW + hw)
    double dsum = 0.0;

    // piece q (0..3: 1 KiB of the image, 4: this wave's copy of the seeds) of code tile (t mod 32) into ring slot t mod 4.
    // `vimg` = image base + this lane's offset inside a tile (made opaque once per phase: hipcc otherwise hoists the per-step
    // addresses of a phase out of the phase loop and spills them)
    auto ring_piece = [&](const char *vimg, int t, int q) __attribute__((always_inline)) {
        const char *src = vimg + (size_t)(t & (T - 1)) * PTILE;
        char *dst = lds + L_RING + (t & (PNBUF - 1)) * PIMG + w4 * 4096;
        if (q < 4) glds16(src + q * 1024, dst + q * 1024);
        else glds4(src + (PIMG - w4 * 4096 - lane * 12), enraw + ((t & (PNBUF - 1)) * 4 + w4) * 64);
    };

    // =================================================================================================================
    // compute role: the code loop of the block held in zb; 32 steps, one barrier each; feeds the ring (tile t + 3 in step t)
    // =================================================================================================================
    auto compute_phase = [&]() __attribute__((always_inline)) {
        PIPE_STAMP(4 * phase_no);
        ++phase_no;
        float b1[2] = {-__builtin_inff(), -__builtin_inff()}, b2[2] = {-__builtin_inff(), -__builtin_inff()};
        int bt[2] = {0, 0};
        f32x4 acc16[2][2];
        auto top2 = [&](int tt) __attribute__((always_inline)) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float om = b1[t2];
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
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
        const char *vimg = a.img + w4 * 4096 + lane * 16;
        asm volatile("" : "+v"(vimg));
#if DVQ_ABLATE & 1024
        unsigned ck_vm = 0, ck_bar = 0, ck_lw = 0;
        const int my_phase = phase_no - 1;
#endif
        for (int t = 0; t < T; ++t) {
            // ring: the computing group feeds it (pieces of tile t + 3 between the MFMAs below; past the phase's end they are
            // the first tiles of the next phase, for the other group).  All but the youngest tile's pieces landed = tiles <= t + 1.
#if DVQ_ABLATE & 1024
            const unsigned c0_ = (unsigned)__builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#if DVQ_ABLATE & 1024
            const unsigned c1_ = (unsigned)__builtin_readcyclecounter();
#endif
            __builtin_amdgcn_s_barrier();                    // tile t landed (everybody's pieces); tile t - 1 consumed
            asm volatile("" ::: "memory");
#if DVQ_ABLATE & 1024
            { const unsigned d_ = (unsigned)__builtin_readcyclecounter() - c0_; if (t < 16) ck_bar += d_; else if (t < 24) ck_vm += d_; else ck_lw += d_; }
#endif
            const unsigned tile_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(
                                        lds + L_RING + (t & (PNBUF - 1)) * PIMG + lane * 16);
            f16x8 a0, a1, a2, a3;
#define PIPE_RD(dst, S) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(tile_a), "i"((S) * 1024))
            PIPE_RD(a0, 0); PIPE_RD(a1, 1); PIPE_RD(a2, 2); PIPE_RD(a3, 3);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) top2(t - 1);
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned seed_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(
                                            enraw + ((t & (PNBUF - 1)) * 4 + w4) * 64 + 4 * q16);
                f32x4 e0, e1;
                asm volatile("ds_read_b128 %0, %1" : "=v"(e0) : "v"(seed_a));
                asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(e1) : "v"(seed_a));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
                acc16[0][0] = e0; acc16[0][1] = e0; acc16[1][0] = e1; acc16[1][1] = e1;
            }
            __builtin_amdgcn_sched_barrier(0);
#define PIPE_MM(src, F, WAIT, NEXT)                                                                            \
            asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" ::: "memory");                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            acc16[(F) / S32][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[0][(F) % S32], acc16[(F) / S32][0], 0, 0, 0); \
            acc16[(F) / S32][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[1][(F) % S32], acc16[(F) / S32][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if ((F) + 4 < S16) { PIPE_RD(src, ((F) + 4 < S16 ? (F) + 4 : 0)); }                                  \
            NEXT
#if DVQ_PIPE_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#endif
            PIPE_MM(a0, 0, 0, ) PIPE_MM(a1, 1, 1, ring_piece(vimg, t + 3, 0);) PIPE_MM(a2, 2, 2, ) PIPE_MM(a3, 3, 3, )
            PIPE_MM(a0, 4, 3, ring_piece(vimg, t + 3, 1);) PIPE_MM(a1, 5, 3, ) PIPE_MM(a2, 6, 3, ) PIPE_MM(a3, 7, 3, ring_piece(vimg, t + 3, 2);)
            PIPE_MM(a0, 8, 3, ) PIPE_MM(a1, 9, 3, ) PIPE_MM(a2, 10, 3, ring_piece(vimg, t + 3, 3);) PIPE_MM(a3, 11, 3, )
            PIPE_MM(a0, 12, 3, ) PIPE_MM(a1, 13, 2, ring_piece(vimg, t + 3, 4);) PIPE_MM(a2, 14, 1, ) PIPE_MM(a3, 15, 0, )
#undef PIPE_MM
#undef PIPE_RD
#if DVQ_PIPE_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#endif
        }
#if DVQ_ABLATE & 1024
        PIPE_STAMPV(4 * my_phase + 3, (unsigned long long)ck_bar | ((unsigned long long)ck_vm << 21) | ((unsigned long long)ck_lw << 42));
#endif
        top2(T - 1);
        // merge the four lane groups of a token column (lower lane wins ties), then hand the results to the lanes that own
        // the token in the (c, h) layout
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
            rc[t2] = mt * 32 + 16 * (r >> 2) + 4 * mq + (r & 3);
        }
        const int srcl = c & 15;
        const float x0 = __shfl(rb[0], srcl), x1 = __shfl(rb[1], srcl);
        const float y0 = __shfl(rs[0], srcl), y1 = __shfl(rs[1], srcl);
        const int c0 = __shfl(rc[0], srcl), c1 = __shfl(rc[1], srcl);
        best = (c >> 4) ? x1 : x0;
        second = (c >> 4) ? y1 : y0;
        code = (c >> 4) ? c1 : c0;
    };

    // =================================================================================================================
    // memory role: epilogue of the block this group just scored (EPI), prologue of the block it scores next (PRO), and
    // the ring DMA of all 32 steps.  blk_next: the block to load (PRO).
    // =================================================================================================================
    // EPI / PRO are wave-uniform run-time flags: ONE copy of the code (hipcc allocates registers for a plain
    // "memory phase, compute phase" loop without spilling; four specialised copies joined at the loop head it does not), with
    // every vmcnt count picked from the pattern of the flag combination by a scalar branch.
    auto memory_phase = [&](const bool EPI, const bool PRO, int blk_next, bool prev_mem) __attribute__((always_inline)) {
        using P11 = Pat<true, true, STORE, SEL>;
        using P10 = Pat<true, false, STORE, SEL>;
        using P01 = Pat<false, true, STORE, SEL>;
        using P00 = Pat<false, false, STORE, SEL>;
#define PIPE_WAIT(FN, ARG)                                                                                     \
        do {                                                                                                      \
            if (EPI) { if (PRO) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P11::FN(ARG)) : "memory");              \
                       else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P10::FN(ARG)) : "memory"); }                \
            else { if (PRO) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P01::FN(ARG)) : "memory");                  \
                   else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P00::FN(ARG)) : "memory"); }                    \
        } while (0)
        PIPE_STAMP(4 * phase_no);
        // ---- pre-step: ordinary loads / atomics (hipcc drains the vector-memory queue at their first use while this wave's last
        // ring pieces are in flight: a stall of at most one L2 latency per phase)
        const float *ep = a.E + 8 * h;                       // chosen codebook row of this lane's token (EPI)
        const float *zqb = a.zq;                             // uniform base of this block's image in z_q (EPI)
        unsigned vst = 0;                                    // running byte offset of this lane's next z_q store
        const unsigned chstride = (unsigned)HW * 4u;
        float m_loss = 0.0f;
        f32x4 eg[3][2];
        if (EPI) {
            const float thr = best - thr2W;
            const bool final_ok = (best - second) > thr2W;
            bool hopeless = !(code < a.K) || !(thr == thr);
            const bool undecided = !hopeless && !final_ok;
            const unsigned long long umask = __ballot(undecided && h == 0);
            const int shard = blockIdx.x & (DVQ_QSHARDS - 1);
            int slot = -1;
            if (umask != 0ull) {                             // wave-uniform
                int slot_raw = 0;
                if (lane == 0) slot_raw = atomicAdd(&a.counters[DVQ_QCOUNT0 + shard], (int)__popcll(umask));
                const int base = __shfl(slot_raw, 0);
                slot = undecided ? base + (int)__popcll(umask & ((1ull << c) - 1ull)) : -1;
                if (slot >= a.rec_cap) { hopeless = true; slot = -1; }          // shard full -> exact list
            }
            if (hopeless && h == 0) {
                int pos = atomicAdd(&a.counters[DVQ_C_EXACT], 1);
                a.exact_list[pos] = tok_n;
            }
            if (slot >= 0) {
                char *rec = a.records + ((size_t)shard * a.rec_cap + slot) * rec_bytes(D);
#pragma unroll
                for (int s = 0; s < S16; ++s) {
                    f32x4 lo = {zf[s][0], zf[s][1], zf[s][2], zf[s][3]};
                    f32x4 hi = {zf[s][4], zf[s][5], zf[s][6], zf[s][7]};
                    *(f32x4 *)(rec + (16 * s + 8 * h) * 4) = lo;
                    *(f32x4 *)(rec + (16 * s + 8 * h + 4) * 4) = hi;
                }
                if (h == 0) {
                    RecMeta rm;
                    rm.n = tok_n; rm.xn = xn; rm.thr = thr; rm.prov = code;
                    rm.m = (SEL != 0) ? sel_mask : ((a.mask != nullptr) ? a.mask[tok_n] : 1.0f);
                    rm.best = ~0ull; rm.rep = 1;
                    *(RecMeta *)(rec + (size_t)D * 4) = rm;
                }
            }
            if (!hopeless && h == 0) a.codes[tok_n] = (long long)code;
            // hopeless tokens: every store below still happens (the wave's operation count must not depend on data); the
            // exact-list kernel rewrites their code and z_q, their loss term is dropped here
            ep = a.E + (size_t)(hopeless ? 0 : code) * D + 8 * h;
            m_loss = hopeless ? 0.0f : ((SEL != 0) ? sel_mask : ((a.mask != nullptr) ? a.mask[tok_n] : 1.0f));
            const int bimg = __builtin_amdgcn_readfirstlane(tok_n / HW);     // a block lies inside one image
            zqb = a.zq + (size_t)bimg * D * HW;
            vst = (unsigned)(((8 * h) * HW + (tok_n - bimg * HW)) * 4);
        }
        // gate of the next block's cells, staging addresses (PRO)
        const int nn = PRO ? (blk_next * 128 + w4 * 32 + c) : 0;
        const int nb = PRO ? (blk_next / bpi) : 0;           // image of the next block (wave-uniform)
        const int hw0 = PRO ? ((blk_next - nb * bpi) * 128) : 0;
        const int hwl = hw0 + w4 * 32 + c;
        DvqGateRaw graw;
        size_t cell = 0;
        int cy = 0, cx = 0;
        if (PRO && SEL != 0) {
            const int y = hwl >> 5, x = c;                    // 32-wide output grid
            const int SC = (SEL == 2) ? 2 : 4;
            cy = y; cx = x;
            cell = (size_t)nb * a.rv.hc * a.rv.wc + (y / SC) * a.rv.wc + x / SC;
            graw = dvq_gate_fetch(a.rv.gate, a.rv.gate_mode, a.rv.G, cell);
        }
        if (EPI) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(eg[0][0]) : "v"(ep));
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(eg[0][1]) : "v"(ep));
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(eg[1][0]) : "v"(ep));
            asm volatile("global_load_dwordx4 %0, %1, off offset:80" : "=v"(eg[1][1]) : "v"(ep));
        }
        float lsum = 0.0f;
        // prologue state
        const float *zsrc = a.E;                             // uniform base of the fine / dense source of the next block: channel 0
        unsigned vld = 0;                                    // running byte offset of this lane's next load
        unsigned stg_a = 0, stg_b = 0;
        int sel_g = 0;
        float pa[2][8];
        float amax = 0.0f, zeta2 = 0.0f;
        f16x8 zprev = {};
        if (PRO) {
            const float *src = (SEL == 0) ? a.z : a.rv.src[a.rv.G - 1];
            zsrc = src + (size_t)nb * D * HW;
            vld = (unsigned)(((8 * h) * HW + hwl) * 4);
        }

#if DVQ_ABLATE & 1024
        unsigned tk_bar = 0, tk_gw = 0, tk_lw = 0, tk0 = 0;
#define PIPE_TICK(ACC) do { const unsigned now_ = (unsigned)__builtin_readcyclecounter(); ACC += now_ - tk0; tk0 = now_; } while (0)
#define PIPE_TICK0() do { tk0 = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define PIPE_TICK(ACC) do { } while (0)
#define PIPE_TICK0() do { } while (0)
#endif
#define PIPE_STEP(t)                                                                                            \
        {                                                                                                          \
            PIPE_TICK0();                                                                                          \
            __builtin_amdgcn_s_barrier();                                                                          \
            asm volatile("" ::: "memory");                                                                         \
            if ((t) < 16) PIPE_TICK(tk_bar); else if ((t) < 24) PIPE_TICK(tk_gw); else PIPE_TICK(tk_lw);           \
            if ((t) == 16) PIPE_STAMP(4 * phase_no + 1);                                                           \
            if ((t) == 24) PIPE_STAMP(4 * phase_no + 2);                                                           \
            if (EPI && (t) < 16) {                                                                                 \
                constexpr int u = (t) < 16 ? (t) : 0;                                                              \
                if (u <= 13 && !(DVQ_ABLATE & 2048)) {                                                             \
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(eg[(u + 2) % 3][0]) : "v"(ep), "i"(64 * (u + 2)));          \
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(eg[(u + 2) % 3][1]) : "v"(ep), "i"(64 * (u + 2) + 16));     \
                }                                                                                                  \
                PIPE_WAIT(gather_wait, u);                                                                         \
                asm volatile("" : "+v"(eg[u % 3][0]), "+v"(eg[u % 3][1]) :: "memory");                              \
                _Pragma("unroll")                                                                                  \
                for (int j = 0; j < 8; ++j) {                                                                      \
                    const float e = eg[u % 3][j >> 2][j & 3];                                                      \
                    const float diff = __fsub_rn(e, zf[u][j]);                                                     \
                    if (STORE && !(DVQ_ABLATE & 4096)) {                                                           \
                        asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(vst), "v"(__fadd_rn(zf[u][j], diff)), "s"(zqb) : "memory");   \
                        if (!(DVQ_ABLATE & 256)) vst += chstride;                                                  \
                    }                                                                                              \
                    if (!(DVQ_ABLATE & 8192)) lsum = __fadd_rn(lsum, __fmul_rn(__fmul_rn(diff, diff), m_loss));    \
                }                                                                                                  \
                if (STORE && !(DVQ_ABLATE & 256)) vst += 8 * chstride;                                             \
                asm volatile("" : "+v"(lsum));      /* the loss terms of this k-step are summed HERE, not parked in scratch */ \
            }                                                                                                      \
            if (PRO && (t) >= 16 && (t) < 24) {                                                                    \
                constexpr int j2 = ((t) >= 16 && (t) < 24) ? (t) - 16 : 0;                                         \
                if ((t) == 16 && SEL != 0) stage_issue();                                                          \
                _Pragma("unroll")                                                                                  \
                for (int q = 0; q < 2; ++q) {                                                                      \
                    _Pragma("unroll")                                                                              \
                    for (int j = 0; j < 8; ++j) {                                                                  \
                        asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(zf[2 * j2 + q][j]) : "v"(vld), "s"(zsrc));   \
                        if (!(DVQ_ABLATE & 128)) vld += chstride;                                                  \
                    }                                                                                              \
                    if (!(DVQ_ABLATE & 128)) vld += 8 * chstride;                                                  \
                }                                                                                                  \
            }                                                                                                      \
            if (PRO && (t) >= 24) {                                                                                \
                constexpr int j2 = (t) >= 24 ? (t) - 24 : 0;                                                       \
                if ((t) == 24) select_init();                                                                      \
                PIPE_WAIT(load_wait, j2);                                                                          \
                _Pragma("unroll")                                                                                  \
                for (int q = 0; q < 2; ++q)                                                                        \
                    _Pragma("unroll")                                                                              \
                    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(zf[2 * j2 + q][j]));                        \
                if (!(DVQ_ABLATE & 512)) { convert(2 * j2); convert(2 * j2 + 1); }                                 \
            }                                                                                                      \
            if ((t) < 2 && !prev_mem) PIPE_WAIT(lead_wait, ((t) < 2 ? (t) : 0));                                   \
        }

        // staging DMA of the coarser branches of the next block (as vq_assign_filter_kernel<D, 2>): every wave of the group
        // issues Pat::NSTAGE instructions
        auto stage_issue = [&]() __attribute__((always_inline)) {
            const int y0 = hw0 >> 5;                         // first output row of the block
            char *img_a = lds + L_IMGA, *img_b = lds + L_IMGB;
            {
                const int ga = a.rv.G - 2;
                const int plane = a.rv.hc * a.rv.sub[ga] * 16;
                const float *src = a.rv.src[ga] + (size_t)nb * D * plane + (size_t)(y0 >> 1) * 16 + (lane & 7) * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    glds16(src + (size_t)((w4 + 4 * i) * 8 + (lane >> 3)) * plane, img_a + (w4 + 4 * i) * 1024);
            }
            if (SEL == 3) {
                const int plane = a.rv.hc * 8;
                const float *src = a.rv.src[0] + (size_t)nb * D * plane + (size_t)(y0 >> 2) * 8 + (lane & 1) * 4;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    glds16(src + (size_t)((w4 + 4 * i) * 32 + (lane >> 1)) * plane, img_b + (w4 + 4 * i) * 1024);
            }
        };
        // grain of this lane's cell, by-products, LDS addresses of its coarser-branch values (after the gate has landed: it
        // was fetched in the pre-step)
        auto select_init = [&]() __attribute__((always_inline)) {
            if (SEL == 0) return;
            sel_g = dvq_gate_reduce(graw, a.rv.gate_mode, a.rv.G, a.rv.thr);
            const int rep_g = a.rv.rep[sel_g];
            sel_mask = 1.0f / (float)(rep_g * rep_g);
            const int SC = (SEL == 2) ? 2 : 4;
            if (h == 0 && a.rv.cmask_out != nullptr) {
                a.rv.cmask_out[nn] = sel_mask;
                if (cy % SC == 0 && cx % SC == 0) {
                    a.rv.indices_out[cell] = sel_g;
                    if (a.rv.gate_mode == 2 && a.rv.gate_out != nullptr) {
                        const float e = graw.f[0];
                        longlong2 gg; gg.x = (e <= a.rv.thr) ? 1 : 0; gg.y = (e > a.rv.thr) ? 1 : 0;
                        *(longlong2 *)(a.rv.gate_out + 2 * cell) = gg;
                    }
                }
            }
            stg_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(lds + L_IMGA) +
                    (unsigned)((8 * h) * 128 + (((w4 >> 1) * 16 + (c >> 1)) << 2));
            stg_b = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(lds + L_IMGB) +
                    (unsigned)((8 * h) * 32 + ((c >> 2) << 2));
        };
        // k-step s of the next block: (SEL) coarser-branch values replace the fine ones, fp16 fragments, norm pieces, bound
        auto convert = [&](int s) __attribute__((always_inline)) {
            if (SEL != 0) {
                if (sel_g == a.rv.G - 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        asm volatile("ds_read_b32 %0, %1 offset:%c2" : "+v"(zf[s][j]) : "v"(stg_a), "i"((16 * s + j) * 128));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (SEL == 3 && sel_g == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        asm volatile("ds_read_b32 %0, %1 offset:%c2" : "+v"(zf[s][j]) : "v"(stg_b), "i"((16 * s + j) * 32));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(zf[s][j]));
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
            asm volatile("" : "+v"(amax), "+v"(zeta2));      // (keep this k-step's reductions in this step)
#pragma unroll
            for (int l = 0; l < 8; ++l) asm volatile("" : "+v"(pa[s & 1][l]));
        };

#if DVQ_PIPE_PRIO == 2
        __builtin_amdgcn_s_setprio(2);
#endif
        PIPE_STEP(0) PIPE_STEP(1) PIPE_STEP(2) PIPE_STEP(3) PIPE_STEP(4) PIPE_STEP(5) PIPE_STEP(6) PIPE_STEP(7)
        PIPE_STEP(8) PIPE_STEP(9) PIPE_STEP(10) PIPE_STEP(11) PIPE_STEP(12) PIPE_STEP(13) PIPE_STEP(14) PIPE_STEP(15)
        PIPE_STEP(16) PIPE_STEP(17) PIPE_STEP(18) PIPE_STEP(19) PIPE_STEP(20) PIPE_STEP(21) PIPE_STEP(22) PIPE_STEP(23)
        PIPE_STEP(24) PIPE_STEP(25) PIPE_STEP(26) PIPE_STEP(27) PIPE_STEP(28) PIPE_STEP(29) PIPE_STEP(30) PIPE_STEP(31)
#undef PIPE_STEP
#undef PIPE_WAIT
#if DVQ_PIPE_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
#if DVQ_ABLATE & 1024
        PIPE_STAMPV(4 * phase_no + 3, (unsigned long long)tk_bar | ((unsigned long long)tk_gw << 21) | ((unsigned long long)tk_lw << 42));
#endif

        ++phase_no;
        if (EPI) dsum += (double)lsum;
        if (PRO) {
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
            thr2W = dvq_filter_threshold(xn, amax, zeta2, sB, a.meta);
            tok_n = nn;
        }
    };

    // ---- phases -1 .. 2R (32 barrier steps each).  Group 0 scores its blocks in the even phases, group 1 in the odd ones;
    // the other group is in the memory role: it finishes the block it scored one phase ago and loads the one it scores next.
    //   group 0:  M(load b0)  C  M  C  ...  C  M(finish)  M(idle)
    //   group 1:  M(idle)  M(load b1)  C  M  ...  M  C  M(finish)
    // (the launcher gives every workgroup an even number of blocks: both groups have R rounds)
    // the first three code tiles of the first compute phase (group 0's): issued by group 0 now, landed long before its
    // prologue's loads have (in-order retirement)
    if (g == 0) {
        const char *vimg0 = a.img + w4 * 4096 + lane * 16;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int q = 0; q < 5; ++q) ring_piece(vimg0, t, q);
    }
    if (g == 1) memory_phase(false, false, 0, true);
    for (int r = 0; r < R; ++r) {
        memory_phase(r > 0, true, b_lo + 2 * r + g, r == 0);
        compute_phase();
    }
    memory_phase(true, false, 0, false);
    if (g == 0) memory_phase(false, false, 0, true);
    (void)nmine;
    PIPE_STAMP(4 * phase_no);
    (void)nother;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.partials != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        __syncthreads();
        double *red = (double *)(lds + L_RED);
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < 8; ++w) s += red[w];
            a.partials[blockIdx.x] = s;
            for (int i = (int)gridDim.x + (int)blockIdx.x; i < a.npart; i += (int)gridDim.x) a.partials[i] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool dvq_pipe_supported(int D, int HW, int K, long N, const DvqRouted *rv)
{
    if (D != PD || dvq_num_tiles(K) != PT || (HW & 127) != 0 || (N & 127) != 0) return false;
    if ((N >> 7) < 1024 || ((N >> 7) & 511) != 0) return false;   // >= 4 blocks per CU (else the fill / drain phases dominate),
                                                             // and an even number per workgroup on 256 CUs (both groups run every round)
    if (rv != nullptr) {
        if (rv->Wout != 32 || rv->HWout != HW) return false;
        for (int g = 0; g < rv->G; ++g)
            if (((uintptr_t)rv->src[g] & 15) != 0) return false;
    }
    return true;
}

int dvq_launch_pipe(const float *z, const char *img16, const DvqF16Meta *meta, const float *E, const float *mask,
                    int HW, int K, long N, float *zq, long long *codes, double *partials, int npart, int *counters,
                    int *exact_list, char *records, int rec_cap, const DvqRouted *rv, hipStream_t st)
{
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;
    const int nblk = (int)(N >> 7);
    int per = (nblk + ncu - 1) / ncu;
    per += per & 1;
    while (nblk % per != 0) per += 2;                        // even, and dividing the block count (nblk is a multiple of 512)
    const int grid = nblk / per;
    PipeArgs a;
    a.z = z; a.img = img16; a.meta = meta; a.E = E; a.mask = mask; a.HW = HW; a.K = K; a.N = N; a.zq = zq; a.codes = codes;
    a.partials = partials; a.npart = npart; a.counters = counters; a.exact_list = exact_list; a.records = records;
    a.rec_cap = rec_cap; a.per_wg = per;
    if (rv != nullptr) a.rv = *rv; else a.rv = DvqRouted{};
    const int sel = (rv == nullptr) ? 0 : (rv->G == 2 ? 2 : 3);
    static unsigned long long done[6] = {0, 0, 0, 0, 0, 0};
#define PIPE_LAUNCH(SELV, STOREV, IDX)                                                                                   \
    {                                                                                                                     \
        int rc = dvq_allow_dynamic_lds((const void *)vq_assign_pipe_kernel<SELV, STOREV>, L_TOTAL, &done[IDX]);           \
        if (rc) return rc;                                                                                                \
        hipLaunchKernelGGL((vq_assign_pipe_kernel<SELV, STOREV>), dim3(grid), dim3(512), L_TOTAL, st, a);                 \
    }
    if (zq != nullptr) {
        if (sel == 0) PIPE_LAUNCH(0, true, 0) else if (sel == 2) PIPE_LAUNCH(2, true, 1) else PIPE_LAUNCH(3, true, 2)
    } else {
        if (sel == 0) PIPE_LAUNCH(0, false, 3) else if (sel == 2) PIPE_LAUNCH(2, false, 4) else PIPE_LAUNCH(3, false, 5)
    }
#undef PIPE_LAUNCH
    return (int)hipGetLastError();
}
#endif  // DVQ_TUNING
