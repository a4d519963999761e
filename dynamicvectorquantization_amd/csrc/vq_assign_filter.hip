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
static constexpr int RES_SLOTS = 128;             // resolver: queued tokens per workgroup
static constexpr int RES_CAND = 1024;             // resolver: candidate pairs per workgroup

// record of one queued token (written by pass 1, read by the resolver)
//   [zh: D*2 B in fragment order s,h,8][zf: D*4 B in channel order][meta 32 B]
__host__ __device__ inline size_t rec_bytes(int D) { return (size_t)D * 6 + 32; }
struct RecMeta { int n; float xn; float thr; float seed_scale; int prov; int pad[3]; };

// ---------------------------------------------------------------------------------------------
// prep: meta (scale, norm maxima, finiteness), fp16 tile images, rounding-residual norm
//   image of tile t: [s < D/16][lane < 64][j < 8] halves = fp16(2^b E[32t + (lane&31)][16s + 8(lane>>5) + j])
//   -> the A fragment of k-step s is ONE ds_read_b128 at s*1024 + lane*16 (lane-linear, conflict-free)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void codebook_meta_kernel(const float *__restrict__ E, int K, int D,
                                                             const float *__restrict__ en_all,
                                                             DvqF16Meta *__restrict__ meta)
{
    __shared__ float s_max[1024];
    __shared__ float s_en[1024];
    __shared__ int s_bad[1024];
    float amax = 0.0f, enmax = 0.0f;
    int bad = 0;
    const size_t total = (size_t)K * D;
    for (size_t i = threadIdx.x; i < total; i += 1024) {
        float v = fabsf(E[i]);
        bad |= !(v < __builtin_inff());
        amax = fmaxf(amax, v);
    }
    for (int j = threadIdx.x; j < K; j += 1024) {
        float v = en_all[j];
        bad |= !(v < __builtin_inff());
        enmax = fmaxf(enmax, v);
    }
    s_max[threadIdx.x] = amax;
    s_en[threadIdx.x] = enmax;
    s_bad[threadIdx.x] = bad;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + w]);
            s_en[threadIdx.x] = fmaxf(s_en[threadIdx.x], s_en[threadIdx.x + w]);
            s_bad[threadIdx.x] |= s_bad[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        amax = s_max[0];
        enmax = s_en[0];
        bad = s_bad[0];
        int b = 0;
        if (amax > 0.0f) {
            int e;
            (void)frexpf(amax, &e);     // amax = m 2^e, m in [0.5, 1)
            b = 15 - e;                 // 2^b amax in [2^14, 2^15)
        }
        if (b > 100 || b < -100) bad = 1;
        meta->ok = bad ? 0 : 1;
        meta->b_exp = b;
        meta->scale_b = ldexpf(1.0f, bad ? 0 : b);
        meta->emax = sqrtf(enmax) * 1.00001f;
        meta->enmax = enmax;
        meta->etamax = 0.0f;            // filled by codebook_eta_kernel
    }
}

__global__ __launch_bounds__(256) void codebook_prep_f16_kernel(const float *__restrict__ E, int K, int D,
                                                                const DvqF16Meta *__restrict__ meta,
                                                                _Float16 *__restrict__ img)
{
    const float sb = meta->scale_b;
    const int S16 = D / 16;
    const size_t per_tile = (size_t)S16 * 512;
    const size_t total = (size_t)dvq_num_tiles(K) * per_tile;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int t = (int)(i / per_tile);
        int r = (int)(i - (size_t)t * per_tile);
        int s = r >> 9, lane = (r >> 3) & 63, j = r & 7;
        int code = t * 32 + (lane & 31);
        int k = 16 * s + 8 * (lane >> 5) + j;
        float v = (code < K) ? E[(size_t)code * D + k] * sb : 0.0f;
        img[i] = (_Float16)v;       // round to nearest even
    }
}

// etamax = max_j || 2^b e_j - fp16(2^b e_j) ||_2 (each residual is exact in fp32), rounded up
__global__ __launch_bounds__(256) void codebook_eta_kernel(const float *__restrict__ E, int K, int D,
                                                           DvqF16Meta *__restrict__ meta)
{
    const float sb = meta->scale_b;
    float best = 0.0f;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < K; j += gridDim.x * blockDim.x) {
        const float *e = E + (size_t)j * D;
        float s = 0.0f;
        for (int k = 0; k < D; ++k) {
            float v = e[k] * sb;
            float r = v - (float)(_Float16)v;
            s += r * r;
        }
        best = fmaxf(best, s);
    }
    for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if ((threadIdx.x & 63) == 0 && best > 0.0f) {
        float v = sqrtf(best) * 1.001f;
        atomicMax((int *)&meta->etamax, __float_as_int(v));       // positive floats order as ints
    }
}

__device__ __forceinline__ float vmax_raw(float a, float b)
{
    float r;     // plain v_max_f32: no canonicalising pre-ops (fmaxf() adds two per call)
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---------------------------------------------------------------------------------------------
// pass 1
//   workgroup = 4 waves x 32 tokens (consecutive hw positions); two workgroups per CU.
//   A wave keeps its 32 tokens twice in registers: fp32 (D/2 VGPRs, read once from NCHW, reused
//   for z_q so z is never re-read) and scaled fp16 MFMA B fragments (D/4 VGPRs).
//   The fp16 codebook streams through LDS in stages of 2 tiles (64 codes), double-buffered by
//   global->LDS DMA, one barrier per stage; the accumulator of every tile is seeded from LDS with
//   -2^(a+b-1) en_j so the MFMA output is the score itself.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void vq_assign_filter_kernel(
    const float *__restrict__ z, const char *__restrict__ img, const DvqF16Meta *__restrict__ meta,
    const float *__restrict__ en_all, const float *__restrict__ E, const float *__restrict__ mask,
    int HW, int K, long N, float *__restrict__ zq, long long *__restrict__ codes,
    double *__restrict__ partials, int *__restrict__ counters, int *__restrict__ exact_list,
    char *__restrict__ records, int rec_cap)
{
    constexpr int S16 = D / 16;
    constexpr int TILE_BYTES = S16 * 1024;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int CHUNKS_PER_WAVE = S16 / 4;                 // 1-KiB DMA pieces per wave per tile
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // [2][STAGE_BYTES] fp16 tiles | [2 buf][4 wave][2 tile][32] accumulator seeds
    float *seedbuf = (float *)(lds + 2 * STAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int T = dvq_num_tiles(K);
    const int NS = (T + 1) / 2;
    const int meta_ok = meta->ok;
    const float sB = meta->scale_b;

    auto stage = [&](int st, int bufi) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = 2 * st + i;
            if (t < T) {
                const char *src = img + (size_t)t * TILE_BYTES;
                char *dst = lds + bufi * STAGE_BYTES + i * TILE_BYTES;
#pragma unroll
                for (int q = 0; q < CHUNKS_PER_WAVE; ++q) {
                    int chunk = wave * CHUNKS_PER_WAVE + q;
                    glds16(src + chunk * 1024 + lane * 16, dst + chunk * 1024);
                }
            }
        }
    };
    stage(0, 0);

    const long n = ((long)blockIdx.x * 4 + wave) * 32 + c;
    const bool valid = n < N;
    const long nn = valid ? n : N - 1;
    const long bimg = nn / HW;
    const int hw = (int)(nn - bimg * HW);
    const size_t zbase = ((size_t)bimg * D + 8 * h) * HW + hw;   // channel 16s + 8h + j at + (16s+j)*HW
    const float *zp = z + zbase;
    float zf[S16][8];
#pragma unroll
    for (int s = 0; s < S16; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) zf[s][j] = zp[(size_t)(16 * s + j) * HW];

    // exact ATen-order xn: a[m], m = i mod 32 = 16(s&1) + 8h + j for channel i = 16s + 8h + j
    float xn, amax = 0.0f;
    {
        float pa[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float a = sq_rn(zf[p][j]);
#pragma unroll
                for (int s = p + 2; s < S16; s += 2) a = __fadd_rn(a, sq_rn(zf[s][j]));
                pa[p][j] = a;
            }
#pragma unroll
            for (int s = 0; s < S16; ++s) amax = vmax_raw(amax, fabsf(zf[s][j]));
        }
        float t8[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float o0 = __shfl_xor(pa[0][l], 32), o1 = __shfl_xor(pa[1][l], 32);
            float a0 = h == 0 ? pa[0][l] : o0;      // a[l]      (h=0, p=0)
            float a1 = h == 0 ? o0 : pa[0][l];      // a[l+8]    (h=1, p=0)
            float a2 = h == 0 ? pa[1][l] : o1;      // a[l+16]   (h=0, p=1)
            float a3 = h == 0 ? o1 : pa[1][l];      // a[l+24]   (h=1, p=1)
            t8[l] = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
        }
        xn = t8[0];
#pragma unroll
        for (int l = 1; l < 8; ++l) xn = __fadd_rn(xn, t8[l]);
    }
    const bool bad = !(xn < __builtin_inff()) || !meta_ok;      // NaN / Inf / overflowing squares
    // wave-uniform scale from the largest finite magnitude of the wave's 32 tokens
    float sA;
    {
        float am = bad ? 0.0f : amax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
        int a_exp = 0;
        if (am > 0.0f) {
            int e;
            (void)frexpf(am, &e);
            a_exp = 15 - e;                                      // 2^a max|z| in [2^14, 2^15)
        }
        a_exp = a_exp > 100 ? 100 : (a_exp < -100 ? -100 : a_exp);
        sA = ldexpf(1.0f, a_exp);
    }
    f16x8 zh[S16];
    float zeta2 = 0.0f;                      // this lane's share of ||2^a z - zh||^2
#pragma unroll
    for (int s = 0; s < S16; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = zf[s][j] * sA;
            _Float16 hv = (_Float16)v;
            zh[s][j] = hv;
            float r = v - (float)hv;         // exact
            zeta2 = __builtin_fmaf(r, r, zeta2);
        }
    float thr2W;
    {
        zeta2 += __shfl_xor(zeta2, 32);
        const float zeta = sqrtf(zeta2) * 1.001f;
        const float Rh = sqrtf(xn) * 1.00001f;
        const float sAB = sA * sB;
        const float emax = meta->emax, enmax = meta->enmax, etamax = meta->etamax;
        const float zn = sA * Rh + zeta;                 // >= ||zh||
        const float ehn = sB * emax + etamax;            // >= ||eh_j||
        float Wv = zeta * ehn + zn * etamax
                   + GAMMA_P * (zn * ehn + 0.5f * sAB * enmax)
                   + PACK_E * sAB * (Rh * emax + 0.5f * enmax)
                   + sAB * (REF_XN * (xn + enmax) + REF_RE * Rh * emax);
        thr2W = 2.0f * Wv * 1.001f;
    }

    // accumulator seeds -2^(a+b-1) en_j: lane (i = lane>>5, c) owns row c of tile i of the stage;
    // the raw norm is fetched one stage ahead and scaled only when it is written to LDS.
    // Padded codes (>= K) get a huge negative FINITE seed: they never win, and packing the register
    // index into the low mantissa bits cannot turn them into NaNs (it would for -inf).
    const float seed_scale = -0.5f * sA * sB;
    constexpr float SEED_PAD = -3.0e38f;
    auto seed_raw = [&](int st) -> float {
        int code = (2 * st + h) * 32 + c;
        return (code < K) ? en_all[code] : __builtin_inff();
    };
    float *my_seed = seedbuf + wave * 64 + lane;                  // + buf * 256
    my_seed[0] = fmaxf(seed_raw(0) * seed_scale, SEED_PAD);
    float raw_next = (NS > 1) ? seed_raw(1) : 0.0f;

    float m1 = -__builtin_inff(), m2 = -__builtin_inff();
    int t1 = 0;

    for (int st = 0; st < NS; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                    // stage st and its seeds landed; stage st-1 fully consumed
        const int bufi = st & 1;
        if (st + 1 < NS) {
            my_seed[((st + 1) & 1) * 256] = fmaxf(raw_next * seed_scale, SEED_PAD);
            stage(st + 1, (st + 1) & 1);
            if (st + 2 < NS) raw_next = seed_raw(st + 2);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = 2 * st + i;
            if (t < T) {
                const char *tile = lds + bufi * STAGE_BYTES + i * TILE_BYTES + lane * 16;
                const float *seeds = seedbuf + bufi * 256 + wave * 64 + i * 32 + 4 * h;
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 e4 = *(const f32x4 *)(seeds + 8 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[4 * g + q] = e4[q];
                }
                f16x8 a0 = *(const f16x8 *)(tile);
                f16x8 a1 = *(const f16x8 *)(tile + 1024);
#pragma unroll
                for (int s = 0; s < S16; s += 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, zh[s], acc, 0, 0, 0);
                    if (s + 2 < S16) a0 = *(const f16x8 *)(tile + (s + 2) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, zh[s + 1], acc, 0, 0, 0);
                    if (s + 3 < S16) a1 = *(const f16x8 *)(tile + (s + 3) * 1024);
                }
                const float om = m1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float g = __uint_as_float((__float_as_uint(acc[r]) & 0xFFFFFFF0u) | (unsigned)r);
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, g);
                    m1 = vmax_raw(m1, g);
                }
                t1 = (m1 != om) ? t : t1;
            }
        }
    }

    // ---- merge the two lane halves; provisional winner; final / queued / exact-list
    int code;
    float thr;
    bool final_ok;
    {
        const float o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
        const int ot = __shfl_xor(t1, 32);
        const bool other_wins = (o1 > m1) || (o1 == m1 && h == 1);   // both lanes of a token agree
        const float best = other_wins ? o1 : m1;
        const float second = fmaxf(other_wins ? m1 : o1, fmaxf(m2, o2));
        const int wt = other_wins ? ot : t1;
        const int wh = other_wins ? (h ^ 1) : h;
        const int r = (int)(__float_as_uint(best) & 15u);
        code = wt * 32 + (r & 3) + 8 * (r >> 2) + 4 * wh;
        thr = best - thr2W;
        final_ok = (best - second) > thr2W;
    }
    // seeds must stay far from the padding value and from overflow: 2^(a+b-1) max en < 1e37
    const bool seeds_ok = (0.5f * sA * sB * meta->enmax) < 1.0e37f;
    bool hopeless = bad || !seeds_ok || !(code < K) || !(thr == thr);   // -> exact list
    int slot = -1;
    if (valid && !hopeless && !final_ok) {                        // queue for the resolver
        if (h == 0) slot = atomicAdd(&counters[0], 1);
        slot = __shfl(slot, c);                                   // lane c (h = 0) of the same token
        if (slot >= rec_cap) { hopeless = true; slot = -1; }
    }
    if (valid && hopeless && h == 0) {
        int pos = atomicAdd(&counters[1], 1);
        exact_list[pos] = (int)n;
    }
    if (slot >= 0) {
        char *rec = records + (size_t)slot * rec_bytes(D);
#pragma unroll
        for (int s = 0; s < S16; ++s) {
            *(f16x8 *)(rec + (s * 2 + h) * 16) = zh[s];
            f32x4 lo = {zf[s][0], zf[s][1], zf[s][2], zf[s][3]};
            f32x4 hi = {zf[s][4], zf[s][5], zf[s][6], zf[s][7]};
            *(f32x4 *)(rec + D * 2 + (16 * s + 8 * h) * 4) = lo;
            *(f32x4 *)(rec + D * 2 + (16 * s + 8 * h + 4) * 4) = hi;
        }
        if (h == 0) {
            RecMeta rm;
            rm.n = (int)n; rm.xn = xn; rm.thr = thr; rm.seed_scale = seed_scale; rm.prov = code;
            rm.pad[0] = rm.pad[1] = rm.pad[2] = 0;
            *(RecMeta *)(rec + (size_t)D * 6) = rm;
        }
    }
    // ---- provisional (usually final) outputs: code, z_q = z + (e - z), loss term
    float lsum = 0.0f;
    if (valid && !hopeless) {
        if (h == 0) codes[n] = (long long)code;
        if (zq != nullptr || partials != nullptr) {
            const float *ep = E + (size_t)code * D + 8 * h;
            float *zqp = zq ? zq + zbase : nullptr;
            const float m = (mask != nullptr) ? mask[n] : 1.0f;
#pragma unroll
            for (int s0 = 0; s0 < S16; s0 += 4) {        // gathers issued 4 k-steps (8 x 16 B) at a time
                f32x4 eg[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (s0 + q < S16) {
                        eg[q][0] = *(const f32x4 *)(ep + 16 * (s0 + q));
                        eg[q][1] = *(const f32x4 *)(ep + 16 * (s0 + q) + 4);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = s0 + q;
                    if (s < S16) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float e = eg[q][j >> 2][j & 3];
                            float diff = __fsub_rn(e, zf[s][j]);
                            if (zqp != nullptr) zqp[(size_t)(16 * s + j) * HW] = __fadd_rn(zf[s][j], diff);
                            lsum = __fadd_rn(lsum, __fmul_rn(__fmul_rn(diff, diff), m));
                        }
                    }
                }
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
        if (tid == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// resolver: queued tokens, RES_SLOTS per workgroup (wave w owns slots 32w .. 32w+31)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long order_key(float d, int code)
{
    d = d + 0.0f;                                 // -0 -> +0: equal distances tie on the index
    unsigned u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone map, finite d
    return ((unsigned long long)u << 32) | (unsigned)code;
}

template <int D>
__global__ __launch_bounds__(256, 2) void vq_resolve_kernel(
    const char *__restrict__ img, const DvqF16Meta *__restrict__ meta, const float *__restrict__ en_all,
    const float *__restrict__ E, const float *__restrict__ mask, int HW, int K,
    float *__restrict__ zq, long long *__restrict__ codes, double *__restrict__ partials,
    int *__restrict__ counters, int *__restrict__ exact_list, const char *__restrict__ records, int rec_cap)
{
    constexpr int S16 = D / 16;
    constexpr int TILE_BYTES = S16 * 1024;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int CHUNKS_PER_WAVE = S16 / 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // [2][STAGE_BYTES] tiles | cand[RES_CAND] | best[RES_SLOTS] u64 | misc ints
    unsigned *cand = (unsigned *)(lds + 2 * STAGE_BYTES);
    unsigned long long *best = (unsigned long long *)(cand + RES_CAND);
    int *misc = (int *)(best + RES_SLOTS);            // [0] candidate count, [1] rewrite count
    int *rewrite = misc + 4;                          // [RES_SLOTS] slots whose winner changed

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    int total = counters[0];
    total = total < rec_cap ? total : rec_cap;
    const int base = blockIdx.x * RES_SLOTS;
    if (base >= total) {
        if (partials != nullptr && tid == 0) partials[blockIdx.x] = 0.0;
        return;
    }
    const int T = dvq_num_tiles(K);
    const int NS = (T + 1) / 2;

    auto stage = [&](int st, int bufi) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = 2 * st + i;
            if (t < T) {
                const char *src = img + (size_t)t * TILE_BYTES;
                char *dst = lds + bufi * STAGE_BYTES + i * TILE_BYTES;
#pragma unroll
                for (int q = 0; q < CHUNKS_PER_WAVE; ++q) {
                    int chunk = wave * CHUNKS_PER_WAVE + q;
                    glds16(src + chunk * 1024 + lane * 16, dst + chunk * 1024);
                }
            }
        }
    };
    stage(0, 0);
    if (tid < RES_SLOTS) best[tid] = ~0ull;
    if (tid < 4) misc[tid] = 0;

    const int slot_l = wave * 32 + c;                 // local slot of this lane's token column
    const bool live = base + slot_l < total;
    const char *rec = records + (size_t)(live ? base + slot_l : base) * rec_bytes(D);
    f16x8 zh[S16];
#pragma unroll
    for (int s = 0; s < S16; ++s) zh[s] = *(const f16x8 *)(rec + (s * 2 + h) * 16);
    const RecMeta rm = *(const RecMeta *)(rec + (size_t)D * 6);
    const float thr = live ? rm.thr : __builtin_inff();
    const float nss = rm.seed_scale;                  // -2^(a+b-1) of the token's pass-1 wave

    // ---- enumerate: every code whose approximate score reaches best - 2W
    for (int st = 0; st < NS; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int bufi = st & 1;
        if (st + 1 < NS) stage(st + 1, (st + 1) & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = 2 * st + i;
            if (t < T) {
                const char *tile = lds + bufi * STAGE_BYTES + i * TILE_BYTES + lane * 16;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < S16; ++s) {
                    f16x8 a = *(const f16x8 *)(tile + s * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, zh[s], acc, 0, 0, 0);
                }
                unsigned hits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float en = (code < K) ? en_all[code] : __builtin_inff();
                    float g = __builtin_fmaf(en, nss, acc[r]);
                    hits |= (g >= thr) ? (1u << r) : 0u;
                }
                while (hits) {
                    int r = __builtin_ctz(hits);
                    hits &= hits - 1;
                    int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    int pos = atomicAdd(&misc[0], 1);
                    if (pos < RES_CAND) cand[pos] = ((unsigned)slot_l << 20) | (unsigned)code;
                }
            }
        }
    }
    __syncthreads();
    const int ncand_raw = misc[0];
    const bool overflow = ncand_raw > RES_CAND;       // hand the whole group to the exact list
    const int ncand = overflow ? 0 : ncand_raw;

    // ---- exact chains: one thread per (token, candidate)
    for (int i = tid; i < ncand; i += 256) {
        const unsigned pc = cand[i];
        const int sl = (int)(pc >> 20), code = (int)(pc & 0xFFFFFu);
        const char *r2 = records + (size_t)(base + sl) * rec_bytes(D);
        const f32x4 *zv = (const f32x4 *)(r2 + D * 2);
        const f32x4 *ev = (const f32x4 *)(E + (size_t)code * D);
        const float xn = ((const RecMeta *)(r2 + (size_t)D * 6))->xn;
        float acc = 0.0f;
#pragma unroll 4
        for (int q = 0; q < D / 4; ++q) {
            f32x4 a = zv[q], b = ev[q];
            acc = __builtin_fmaf(a[0], b[0], acc);
            acc = __builtin_fmaf(a[1], b[1], acc);
            acc = __builtin_fmaf(a[2], b[2], acc);
            acc = __builtin_fmaf(a[3], b[3], acc);
        }
        float bias = __fadd_rn(xn, en_all[code]);
        float d = __builtin_fmaf(-2.0f, acc, bias);
        atomicMin(&best[sl], order_key(d, code));
    }
    __syncthreads();

    // ---- winners; slots whose winner differs from pass 1 are rewritten
    if (tid < RES_SLOTS && base + tid < total) {
        const char *r2 = records + (size_t)(base + tid) * rec_bytes(D);
        const RecMeta m2 = *(const RecMeta *)(r2 + (size_t)D * 6);
        if (overflow || best[tid] == ~0ull) {
            int pos = atomicAdd(&counters[1], 1);     // cannot resolve here: full exact evaluation;
            exact_list[pos] = m2.n;                   // pass 1's loss term for it is taken back below
            pos = atomicAdd(&misc[1], 1);
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
    for (int i = wave; i < nrew; i += 4) {            // one wave per rewritten token, 4 channels per lane
        const int sl = rewrite[i] >> 20, win = rewrite[i] & 0xFFFFF;
        const bool take_back_only = win == 0xFFFFF;   // token went to the exact list
        const char *r2 = records + (size_t)(base + sl) * rec_bytes(D);
        const RecMeta m2 = *(const RecMeta *)(r2 + (size_t)D * 6);
        const long n = m2.n;
        const long bimg = n / HW;
        const int hw = (int)(n - bimg * HW);
        const float m = (mask != nullptr) ? mask[n] : 1.0f;
        float delta = 0.0f;
        for (int k0 = lane * 4; k0 < D; k0 += 256) {
            f32x4 zv = *(const f32x4 *)(r2 + D * 2 + k0 * 4);
            f32x4 eo = *(const f32x4 *)(E + (size_t)m2.prov * D + k0);
            f32x4 en_ = take_back_only ? eo : *(const f32x4 *)(E + (size_t)win * D + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dn = __fsub_rn(en_[j], zv[j]), dold = __fsub_rn(eo[j], zv[j]);
                if (zq != nullptr && !take_back_only)
                    zq[((size_t)bimg * D + k0 + j) * HW + hw] = __fadd_rn(zv[j], dn);
                float tn = take_back_only ? 0.0f : __fmul_rn(__fmul_rn(dn, dn), m);
                delta += tn - __fmul_rn(__fmul_rn(dold, dold), m);
            }
        }
        if (lane == 0 && !take_back_only) codes[n] = (long long)win;
        dsum += (double)delta;
    }
    if (partials != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        __syncthreads();
        double *red = (double *)lds;
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int dvq_launch_exact_list(const float *z, const float *prep, const float *E, const float *mask,
                          int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                          const int *list, const int *list_count, hipStream_t st);

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static int rec_capacity(long N)
{
    long cap = N / 8;
    if (cap < 4096) cap = 4096;
    if (cap > N) cap = N;
    cap = (cap + RES_SLOTS - 1) / RES_SLOTS * RES_SLOTS;
    return (int)cap;
}

bool dvq_filter_supported(int D, int HW, int K, long N)
{
    (void)HW;
    return (D == 64 || D == 128 || D == 256) && N < (1L << 31) && K < (1 << 20);
}

// ws_extra: [counters 256 B][exact list N ints][records cap * rec_bytes]
size_t dvq_filter_ws_extra_bytes(int D, int HW, int K, long N)
{
    (void)HW; (void)K;
    return 256 + align256((size_t)N * sizeof(int)) + align256((size_t)rec_capacity(N) * rec_bytes(D));
}

int dvq_launch_prep_f16(const float *E, int K, int D, void *prep, hipStream_t st)
{
    char *base = (char *)prep + dvq_prep_f16_offset(K, D);
    base = (char *)(((uintptr_t)base + 255) / 256 * 256);
    DvqF16Meta *meta = (DvqF16Meta *)base;
    _Float16 *img = (_Float16 *)(base + 256);
    const float *en_all = (const float *)((char *)prep + dvq_prep_en_offset(K, D));
    hipLaunchKernelGGL(codebook_meta_kernel, dim3(1), dim3(1024), 0, st, E, K, D, en_all, meta);
    size_t total = (size_t)dvq_num_tiles(K) * (D / 16) * 512;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(codebook_prep_f16_kernel, dim3(blocks), dim3(256), 0, st, E, K, D, meta, img);
    hipLaunchKernelGGL(codebook_eta_kernel, dim3((K + 255) / 256), dim3(256), 0, st, E, K, D, meta);
    return (int)hipGetLastError();
}

// partials layout: [pass 1: ceil(N/128)][resolver: cap/RES_SLOTS][exact list: ceil(N/128)]
int dvq_filter_nparts(long N) { return 2 * (int)((N + 127) / 128) + rec_capacity(N) / RES_SLOTS; }

template <int D>
static int launch_filter(const float *z, const char *img, const DvqF16Meta *meta, const float *en_all,
                         const float *E, const float *mask, int HW, int K, long N, float *zq,
                         long long *codes, double *partials, int *counters, int *exact_list,
                         char *records, int cap, hipStream_t st)
{
    static bool attr_set = false;
    const size_t shmem1 = 4 * (size_t)(D / 16) * 1024 + 2 * 256 * sizeof(float);
    const size_t shmem2 = 4 * (size_t)(D / 16) * 1024 + RES_CAND * 4 + RES_SLOTS * 8 + 16 + RES_SLOTS * 4;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)vq_assign_filter_kernel<D>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem1);
        (void)hipFuncSetAttribute((const void *)vq_resolve_kernel<D>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2);
        attr_set = true;
    }
    const int nb1 = (int)((N + 127) / 128);
    hipLaunchKernelGGL(vq_assign_filter_kernel<D>, dim3(nb1), dim3(256), shmem1, st, z, img, meta,
                       en_all, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, cap);
    hipLaunchKernelGGL(vq_resolve_kernel<D>, dim3(cap / RES_SLOTS), dim3(256), shmem2, st, img, meta,
                       en_all, E, mask, HW, K, zq, codes, partials ? partials + nb1 : nullptr, counters,
                       exact_list, records, cap);
    return (int)hipGetLastError();
}

int dvq_launch_filter(const float *z, const void *prep, const float *E, const float *mask,
                      int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                      void *ws_extra, hipStream_t st)
{
    char *base = (char *)prep + dvq_prep_f16_offset(K, D);
    base = (char *)(((uintptr_t)base + 255) / 256 * 256);
    const DvqF16Meta *meta = (const DvqF16Meta *)base;
    const char *img = base + 256;
    const float *en_all = (const float *)((const char *)prep + dvq_prep_en_offset(K, D));
    int *counters = (int *)ws_extra;
    int *exact_list = (int *)((char *)ws_extra + 256);
    char *records = (char *)ws_extra + 256 + align256((size_t)N * sizeof(int));
    const int cap = rec_capacity(N);
    hipError_t e = hipMemsetAsync(counters, 0, 16, st);
    if (e != hipSuccess) return (int)e;
    int rc;
    switch (D) {
    case 64:  rc = launch_filter<64>(z, img, meta, en_all, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, cap, st); break;
    case 128: rc = launch_filter<128>(z, img, meta, en_all, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, cap, st); break;
    case 256: rc = launch_filter<256>(z, img, meta, en_all, E, mask, HW, K, N, zq, codes, partials, counters, exact_list, records, cap, st); break;
    default:  return -1000;
    }
    if (rc) return rc;
    double *partials3 = partials ? partials + (N + 127) / 128 + cap / RES_SLOTS : nullptr;
    return dvq_launch_exact_list(z, (const float *)prep, E, mask, D, HW, K, N, zq, codes, partials3,
                                 exact_list, counters + 1, st);
}
