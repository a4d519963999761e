// Loop-only experiment for VERDICT r3 item 7: the code loop of pass 1 (K = 1024 as 32 tiles of 32 codes, D = 256, fp16 MFMA scores with
// a running top-2 per token) in three forms, everything but the loop stripped away (no latents loaded, no epilogue: the token operands
// are random fp16 values held in registers, the code tiles stream from a 544-KiB image through the same 4-slot LDS ring by LDS-DMA):
//   A  today's product form: two workgroups of 4 waves per CU (two waves per SIMD), 32 tokens per wave, v_mfma_f32_16x16x32_f16, every
//      code fragment feeds 2 MFMAs, the top-2 update of tile t-1 sits between the first fragment reads of tile t and its MFMA chain
//   B  ONE wave per SIMD (launch bound 1 workgroup / CU: 512 registers), 64 tokens per wave, the same MFMA: every code fragment feeds 4
//      MFMAs (half the LDS reads, ring DMA, barriers and waits per MFMA), two accumulator sets, the top-2 update of tile t-1
//      hand-interleaved into the MFMA gaps of tile t (2 - 3 vector instructions behind each MFMA)
//   C  as B on v_mfma_f32_32x32x16_f16 (half the MFMA instructions for the same flops: 2 per fragment)
// All forms do the same work per CU: 256 tokens x 1024 codes per "generation", GEN generations per launch (the product kernel has 4;
// more here so that the clock governor settles).  Reported: kernel time, shader cycles per 32-code tile and 64 tokens of a SIMD
// (s_memtime around the loop), the in-loop clock (s_memtime / s_memrealtime).  The accumulated top-2 is written out so nothing is dead.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pass1_loop_forms.hip -o /tmp/p1forms && /tmp/p1forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NBUF 4
#define IMG_BYTES 16384                       // one tile: 16 fragments x 1 KiB
#define TILES 32
#define PER_TILE 5                            // DMA pieces per wave and tile: 4 KiB of the image + this wave's 1-KiB seeds copy

__device__ __forceinline__ void glds16(const void *g, void *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
__device__ __forceinline__ float vmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#define SB() __builtin_amdgcn_sched_barrier(0)

struct Stamp { unsigned long long c0, c1, r0, r1; };

// FORM 0 = A, 1 = B, 2 = C
template <int FORM>
__global__ __launch_bounds__(256, FORM == 0 ? 2 : 1) void loop_kernel(const char *__restrict__ img, const f16x8 *__restrict__ ztok, int gens,
                                                                      float *__restrict__ out, Stamp *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];           // [NBUF][IMG_BYTES] | seeds [NBUF][4 waves][256 B]
    char *seeds = lds + NBUF * IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NZ = FORM == 0 ? 16 : 32;                              // token operand fragments per wave (32 / 64 tokens x 256 channels)
    f16x8 zb[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) zb[i] = ztok[((size_t)blockIdx.x * 4 + wave) * 32 * 64 + i * 64 + lane];

    const char *isrc = img;
    char *idst = lds;
    int it = 0;
    auto issue_begin = [&](int t) {                                       // tile t (mod TILES) of the image into slot t % NBUF
        it = t;
        isrc = img + (size_t)(t & (TILES - 1)) * (IMG_BYTES + 1024) + wave * 4096 + lane * 16;
        idst = lds + (t & (NBUF - 1)) * IMG_BYTES + wave * 4096;
    };
    auto issue_piece = [&](int q) {
        if (q < 4) glds16(isrc + q * 1024, idst + q * 1024);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + (size_t)(it & (TILES - 1)) * (IMG_BYTES + 1024) + IMG_BYTES + lane * 4),
                                              (__attribute__((address_space(3))) void *)(seeds + ((it & (NBUF - 1)) * 4 + wave) * 256), 4, 0, 0);
    };
    auto issue = [&](int t) { issue_begin(t);
#pragma unroll
        for (int q = 0; q < PER_TILE; ++q) issue_piece(q); };
    for (int t = 0; t < 3; ++t) issue(t);

    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)lds;
    const unsigned seeds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)seeds;
    const int q16 = lane >> 4;
    const int T = TILES * gens;
    Stamp st;
    st.c0 = __builtin_amdgcn_s_memtime();
    st.r0 = __builtin_amdgcn_s_memrealtime();

#define RD(dst, S) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(tile_a), "i"((S) * 1024))
    if constexpr (FORM == 0) {
        // ---- A: the product loop (vq_assign_filter.hip), 32 tokens per wave
        float b1[2] = {-__builtin_inff(), -__builtin_inff()}, b2[2] = {-__builtin_inff(), -__builtin_inff()};
        int bt[2] = {0, 0};
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        auto top2 = [&](int tt) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float om = b1[t2];
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const float v0 = acc[r >> 2][t2][r & 3], v1 = acc[(r + 1) >> 2][t2][(r + 1) & 3];
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
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_begin(t + 3);
            const unsigned tile_a = lds0 + (t & (NBUF - 1)) * IMG_BYTES + lane * 16;
            f16x8 a0, a1, a2, a3;
            RD(a0, 0); RD(a1, 1); RD(a2, 2); RD(a3, 3);
            SB();
            if (t > 0) top2(t - 1);
            SB();
            {
                const unsigned seed_a = seeds0 + ((t & (NBUF - 1)) * 4 + wave) * 256 + 16 * q16;
                f32x4 e0, e1;
                asm volatile("ds_read_b128 %0, %1" : "=v"(e0) : "v"(seed_a));
                asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(e1) : "v"(seed_a));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
                acc[0][0] = e0; acc[0][1] = e0; acc[1][0] = e1; acc[1][1] = e1;
            }
            SB();
#define PIECE(Q) issue_piece(Q);
#define MM(src, F, WAIT, NEXT)                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" ::: "memory");                                              \
            SB();                                                                                                   \
            acc[(F) / 8][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[(F) % 8], acc[(F) / 8][0], 0, 0, 0);   \
            acc[(F) / 8][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(src, zb[8 + (F) % 8], acc[(F) / 8][1], 0, 0, 0); \
            SB();                                                                                                   \
            if ((F) + 4 < 16) { RD(src, ((F) + 4 < 16 ? (F) + 4 : 0)); }                                            \
            NEXT
            __builtin_amdgcn_s_setprio(1);
            MM(a0, 0, 0, ) MM(a1, 1, 1, PIECE(0)) MM(a2, 2, 2, ) MM(a3, 3, 3, )
            MM(a0, 4, 3, PIECE(1)) MM(a1, 5, 3, ) MM(a2, 6, 3, ) MM(a3, 7, 3, PIECE(2))
            MM(a0, 8, 3, ) MM(a1, 9, 3, ) MM(a2, 10, 3, PIECE(3)) MM(a3, 11, 3, )
            MM(a0, 12, 3, ) MM(a1, 13, 2, PIECE(4)) MM(a2, 14, 1, ) MM(a3, 15, 0, )
#undef MM
#undef PIECE
            __builtin_amdgcn_s_setprio(0);
        }
        top2(T - 1);
        st.c1 = __builtin_amdgcn_s_memtime();
        st.r1 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        out[(size_t)blockIdx.x * 256 + tid] = b1[0] + b2[0] + b1[1] + b2[1] + (float)(bt[0] + bt[1]);
    } else if constexpr (FORM == 1) {
        // ---- B: 64 tokens per wave, four MFMAs per fragment, the previous tile's top-2 update in the gaps
        float b1[4], b2[4], om[4];
        int bt[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { b1[i] = -__builtin_inff(); b2[i] = -__builtin_inff(); bt[i] = 0; om[i] = 0.0f; }
        f32x4 accA[2][4], accB[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { accA[i][j] = f32x4{0, 0, 0, 0}; accB[i][j] = f32x4{0, 0, 0, 0}; }
        auto tile = [&](int t, f32x4 (&cur)[2][4], f32x4 (&prv)[2][4]) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_begin(t + 3);
            const unsigned tile_a = lds0 + (t & (NBUF - 1)) * IMG_BYTES + lane * 16;
            f16x8 a[4];
            RD(a[0], 0); RD(a[1], 1); RD(a[2], 2); RD(a[3], 3);
            const unsigned seed_a = seeds0 + ((t & (NBUF - 1)) * 4 + wave) * 256 + 16 * q16;
            f32x4 e[2];
            asm volatile("ds_read_b128 %0, %1" : "=v"(e[0]) : "v"(seed_a));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(e[1]) : "v"(seed_a));
            // the seeds were issued last: they are needed first (as the C operand of the first MFMAs) -> one full wait per tile
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: "memory");
            SB();
            float g0 = 0.0f, g1 = 0.0f, md = 0.0f;
#pragma unroll
            for (int F = 0; F < 16; ++F) {
                const int c2 = F / 8, s = F % 8;
                const int p2 = F / 4, r = 2 * (F % 4);                   // the previous tile's pair handled behind this fragment's MFMAs
                if (F >= 4) {                                            // fragments 0..3 were waited for with the seeds
                    if (F < 13) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[F & 3]) :: "memory");
                    else if (F == 13) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[F & 3]) :: "memory");
                    else if (F == 14) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[F & 3]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[F & 3]) :: "memory");
                }
                SB();
                cur[c2][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[F & 3], zb[s], s == 0 ? e[c2] : cur[c2][0], 0, 0, 0);
                SB();
                if (r == 0) om[p2] = b1[p2];
                {
                    const float v0 = prv[r >> 2][p2][r & 3], v1 = prv[(r + 1) >> 2][p2][(r + 1) & 3];
                    g0 = __uint_as_float((__float_as_uint(v0) & 0xFFFFFFF0u) | (unsigned)r);
                    g1 = __uint_as_float((__float_as_uint(v1) & 0xFFFFFFF0u) | (unsigned)(r + 1));
                }
                SB();
                cur[c2][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[F & 3], zb[8 + s], s == 0 ? e[c2] : cur[c2][1], 0, 0, 0);
                SB();
                md = __builtin_amdgcn_fmed3f(b1[p2], g0, g1);
                b1[p2] = vmax3_raw(b1[p2], g0, g1);
                SB();
                cur[c2][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[F & 3], zb[16 + s], s == 0 ? e[c2] : cur[c2][2], 0, 0, 0);
                SB();
                b2[p2] = vmax_raw(b2[p2], md);
                if (r == 6) bt[p2] = (b1[p2] != om[p2]) ? t : bt[p2];
                SB();
                cur[c2][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[F & 3], zb[24 + s], s == 0 ? e[c2] : cur[c2][3], 0, 0, 0);
                SB();
                if (F + 4 < 16) { RD(a[F & 3], (F + 4 < 16 ? F + 4 : 0)); }
                if (F == 1) issue_piece(0);
                if (F == 4) issue_piece(1);
                if (F == 7) issue_piece(2);
                if (F == 10) issue_piece(3);
                if (F == 13) issue_piece(4);
                SB();
            }
        };
        for (int t = 0; t < T; t += 2) {
            tile(t, accA, accB);
            tile(t + 1, accB, accA);
        }
        st.c1 = __builtin_amdgcn_s_memtime();
        st.r1 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float o = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o += b1[i] + b2[i] + (float)bt[i];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) o += accB[i][j][0];
        out[(size_t)blockIdx.x * 256 + tid] = o;
    } else {
        // ---- C: 64 tokens per wave on 32x32x16: a fragment = 32 codes x 16 channels feeds two MFMAs (token halves); lane holds 16 codes of one token
        float b1[2], b2[2], om[2];
        int bt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { b1[i] = -__builtin_inff(); b2[i] = -__builtin_inff(); bt[i] = 0; om[i] = 0.0f; }
        f32x16 accA[2], accB[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) { accA[j][i] = 0.0f; accB[j][i] = 0.0f; }
        auto tile = [&](int t, f32x16 (&cur)[2], f32x16 (&prv)[2]) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_begin(t + 3);
            const unsigned tile_a = lds0 + (t & (NBUF - 1)) * IMG_BYTES + lane * 16;
            f16x8 a[4];
            RD(a[0], 0); RD(a[1], 1); RD(a[2], 2); RD(a[3], 3);
            const unsigned seed_a = seeds0 + ((t & (NBUF - 1)) * 4 + wave) * 256 + 16 * (lane >> 5);
            f32x4 e4[4];                                                 // 16 seeds: rows (i & 3) + 8 (i >> 2) + 4 h
            asm volatile("ds_read_b128 %0, %1" : "=v"(e4[0]) : "v"(seed_a));
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(e4[1]) : "v"(seed_a));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(e4[2]) : "v"(seed_a));
            asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(e4[3]) : "v"(seed_a));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e4[0]), "+v"(e4[1]), "+v"(e4[2]), "+v"(e4[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: "memory");
            f32x16 e;
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = e4[i >> 2][i & 3];
            SB();
            float g0 = 0.0f, g1 = 0.0f, md = 0.0f;
#pragma unroll
            for (int F = 0; F < 16; ++F) {
                const int p2 = F / 8, r = 2 * (F % 8);
                if (F >= 4) {
                    if (F < 13) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[F & 3]) :: "memory");
                    else if (F == 13) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[F & 3]) :: "memory");
                    else if (F == 14) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[F & 3]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[F & 3]) :: "memory");
                }
                SB();
                cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[F & 3], zb[F], F == 0 ? e : cur[0], 0, 0, 0);
                SB();
                if (r == 0) om[p2] = b1[p2];
                {
                    const float v0 = prv[p2][r], v1 = prv[p2][r + 1];
                    g0 = __uint_as_float((__float_as_uint(v0) & 0xFFFFFFF0u) | (unsigned)r);
                    g1 = __uint_as_float((__float_as_uint(v1) & 0xFFFFFFF0u) | (unsigned)(r + 1));
                }
                md = __builtin_amdgcn_fmed3f(b1[p2], g0, g1);
                SB();
                cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[F & 3], zb[16 + F], F == 0 ? e : cur[1], 0, 0, 0);
                SB();
                b1[p2] = vmax3_raw(b1[p2], g0, g1);
                b2[p2] = vmax_raw(b2[p2], md);
                if (r == 14) bt[p2] = (b1[p2] != om[p2]) ? t : bt[p2];
                if (F + 4 < 16) { RD(a[F & 3], (F + 4 < 16 ? F + 4 : 0)); }
                if (F == 1) issue_piece(0);
                if (F == 4) issue_piece(1);
                if (F == 7) issue_piece(2);
                if (F == 10) issue_piece(3);
                if (F == 13) issue_piece(4);
                SB();
            }
        };
        for (int t = 0; t < T; t += 2) {
            tile(t, accA, accB);
            tile(t + 1, accB, accA);
        }
        st.c1 = __builtin_amdgcn_s_memtime();
        st.r1 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float o = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) o += b1[i] + b2[i] + (float)bt[i] + accB[i][0];
        out[(size_t)blockIdx.x * 256 + tid] = o;
    }
#undef RD
    if (tid == 0) stamps[blockIdx.x] = st;
}

template <int FORM>
static void run(const char *name, const char *img, const f16x8 *ztok, float *out, Stamp *stamps, int gens, int ncu)
{
    const int grid = FORM == 0 ? 2 * ncu : ncu;
    const size_t shm = FORM == 0 ? NBUF * IMG_BYTES + NBUF * 4 * 256 : 100 * 1024;      // B / C: more than half a CU's LDS -> one workgroup per CU
    (void)hipFuncSetAttribute((const void *)loop_kernel<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(loop_kernel<FORM>, dim3(grid), dim3(256), shm, 0, img, ztok, gens, out, stamps);
    (void)hipDeviceSynchronize();
    const int reps = 50;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(loop_kernel<FORM>, dim3(grid), dim3(256), shm, 0, img, ztok, gens, out, stamps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { printf("\"%s\": \"launch failed\"", name); return; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid);
    (void)hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto &s : h) {
        cyc.push_back((double)(s.c1 - s.c0) / (TILES * gens));
        clk.push_back((double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) * 10.0) );      // s_memrealtime ticks at 100 MHz: GHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double us = ms * 1e3 / reps;
    // per tile and 64 tokens of one SIMD: form A runs two waves (2 x 32 tokens) per SIMD concurrently, B / C one wave of 64
    const double tiles_per_simd = (double)TILES * gens;
    printf("\"%s\": {\"kernel_us\": %.2f, \"us_per_generation\": %.3f, \"counter_ticks_per_tile_median\": %.1f, \"counter_GHz_median\": %.3f, "
           "\"wall_ns_per_tile_of_64_tokens_per_simd\": %.2f}", name, us, us / gens, cyc[cyc.size() / 2], clk[clk.size() / 2], us * 1e3 / tiles_per_simd);
}

int main(int argc, char **argv)
{
    const int gens = argc > 1 ? atoi(argv[1]) : 64;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t img_bytes = (size_t)TILES * (IMG_BYTES + 1024);
    std::vector<_Float16> himg(img_bytes / 2);
    srand(1);
    for (auto &v : himg) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    const size_t nz = (size_t)2 * ncu * 4 * 32 * 64;                      // fragments of 8 halves
    std::vector<_Float16> hz(nz * 8);
    for (auto &v : hz) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    char *img; f16x8 *ztok; float *out; Stamp *stamps;
    (void)hipMalloc(&img, img_bytes); (void)hipMalloc(&ztok, nz * 16); (void)hipMalloc(&out, (size_t)2 * ncu * 256 * 4); (void)hipMalloc(&stamps, 2 * ncu * sizeof(Stamp));
    (void)hipMemcpy(img, himg.data(), img_bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(ztok, hz.data(), nz * 16, hipMemcpyHostToDevice);
    printf("{\"generations_per_launch\": %d, \"cus\": %d, ", gens, ncu);
    run<0>("A_two_waves_per_simd_32_tokens_16x16x32", img, ztok, out, stamps, gens, ncu); printf(", ");
    run<1>("B_one_wave_per_simd_64_tokens_16x16x32", img, ztok, out, stamps, gens, ncu); printf(", ");
    run<2>("C_one_wave_per_simd_64_tokens_32x32x16", img, ztok, out, stamps, gens, ncu); printf(", ");
    run<0>("A_again", img, ztok, out, stamps, gens, ncu);
    printf("}\n");
    return 0;
}
