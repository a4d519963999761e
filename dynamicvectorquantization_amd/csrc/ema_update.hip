// ema_update.hip -- training-mode codebook statistics for gfx950.
//
// Replaces the dense part of VQEmbedding._update_buffers (reference modules/vector_quantization/
// quantize2_mask.py:66-84): the reference builds a one-hot [K, N] fp32 matrix (1 GiB at K = 1024,
// N = 262144), scatters ones into it, row-sums it and multiplies it with the [N, D] token matrix.
// Here: cluster_size[j] = #tokens with code j and vectors_sum[j, :] = sum of their vectors, straight
// from the NCHW latents.  A workgroup transposes a [64 channel x 64 token] tile through LDS so that a
// wave adds 64 CONSECUTIVE channels of one token to its code's row per instruction: 256 contiguous
// bytes per float-atomic wave-instruction, the shape that runs at the chip's full atomic rate
// (64 lanes hitting 64 different rows would be ~17x slower).  Bound: float-atomic throughput.
// Float atomics add in arrival order: sums agree with the reference to rounding (1e-5), not bit for bit.
// Privatising the sums in LDS ([K][33] fp32 per 32-channel slice, ds_add_f32, one global atomic per non-zero entry at the
// end: 8 M instead of 67 M global atomics at B = 256) was built and measured in round 3: 370 us against this kernel's 288 --
// a ds_add_f32 wave-instruction takes 192 cycles whatever its address pattern (tools/micro/lds_atomic_rate.hip: three
// cycles per lane, 170 G adds/s over the chip, below the 233 G/s the L2 atomics reach here); profiles/archive/r03_ema_lds_negative.json.
#include "dvq_common.h"

// COMBINE: tokens of a tile that chose the same code are summed in LDS first and reach the global sums as ONE row of
// atomics (leader = the first such token; members as a 64-bit mask).  Equal codes inside 64 consecutive positions are the
// rule, not the exception: the 2 x 2 / 4 x 4 copies of a coarse cell carry one code, and a trained codebook's usage is
// skewed -- atomics on one address serialise in L2 (10 % of the tokens on one code: 1314 us without, profiles/
// r02_ema_lds_table_negative_result.txt).  Needs a [K] int table in LDS: K <= 8192 (larger codebooks: plain form).
template <bool COMBINE>
__global__ __launch_bounds__(256) void ema_accumulate_kernel(const float *__restrict__ z,
                                                             const long long *__restrict__ codes, int D, int HW,
                                                             long N, int K, float *__restrict__ cluster_size,
                                                             float *__restrict__ vectors_sum)
{
    __shared__ float tile[64][65];                 // [channel][token], +1 pad: conflict-free transpose
    __shared__ int code_s[64];
    __shared__ unsigned mask_s[64][2];             // COMBINE: members of the leader's class (bit t = token t), 0 for non-leaders
    extern __shared__ int first_s[];               // COMBINE: [K] first token of the tile with this code (64 = none)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tok0 = (long)blockIdx.x * 64;
    if (COMBINE) {
        for (int i = threadIdx.x; i < K; i += 256) first_s[i] = 64;
        if (threadIdx.x < 64) { mask_s[threadIdx.x][0] = 0u; mask_s[threadIdx.x][1] = 0u; }
        __syncthreads();
    }
    {
        const long n = tok0 + threadIdx.x;
        if (threadIdx.x < 64) {
            long long cj = (n < N) ? codes[n] : -1;
            const bool ok = cj >= 0 && cj < K;
            code_s[threadIdx.x] = ok ? (int)cj : -1;
            if (COMBINE) {
                if (ok) atomicMin(&first_s[(int)cj], (int)threadIdx.x);
            } else if (ok) {
                atomicAdd(&cluster_size[cj], 1.0f);
            }
        }
    }
    if (COMBINE) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const int cj = code_s[threadIdx.x];
            if (cj >= 0) atomicOr(&mask_s[first_s[cj]][threadIdx.x >> 5], 1u << (threadIdx.x & 31));
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int cj = code_s[threadIdx.x];
            const unsigned m0 = mask_s[threadIdx.x][0], m1 = mask_s[threadIdx.x][1];
            if (cj >= 0 && (m0 | m1) != 0u) atomicAdd(&cluster_size[cj], (float)(__popc(m0) + __popc(m1)));
        }
    }
    const long n = tok0 + lane;                    // token of this lane while loading
    const long nn = (n < N) ? n : N - 1;
    const long b = nn / HW;
    const int hw = (int)(nn - b * HW);
    const float *zp = z + (size_t)b * D * HW + hw;
    for (int c0 = 0; c0 < D; c0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {             // wave w loads channels c0 + 16w + i: 64 tokens, coalesced
            const int ch = 16 * wave + i;
            tile[ch][lane] = (c0 + ch < D) ? zp[(size_t)(c0 + ch) * HW] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {             // wave w adds tokens 16w + i: lane = channel
            const int tk = 16 * wave + i;
            const int cj = code_s[tk];
            if (COMBINE) {
                unsigned m0 = __builtin_amdgcn_readfirstlane(mask_s[tk][0]), m1 = __builtin_amdgcn_readfirstlane(mask_s[tk][1]);
                if ((m0 | m1) == 0u) continue;     // not a leader (or no valid code)
                float sum = 0.0f;
                while (m0) { const int t = __builtin_ctz(m0); m0 &= m0 - 1; sum += tile[lane][t]; }
                while (m1) { const int t = __builtin_ctz(m1); m1 &= m1 - 1; sum += tile[lane][32 + t]; }
                if (c0 + lane < D) atomicAdd(&vectors_sum[(size_t)cj * D + c0 + lane], sum);
            } else if (cj >= 0 && c0 + lane < D) {
                atomicAdd(&vectors_sum[(size_t)cj * D + c0 + lane], tile[lane][tk]);
            }
        }
    }
}

// Gradient of the commitment loss with respect to the codebook (quantizers that train it by back-propagation: VectorQuantizer2,
// quantize_vqgan.py:290-298; VectorQuantize2 without EMA): g_w[j, :] += c * sum over the tokens t with code j of (z_t - e_j) m_t,
// c = -(g_loss * 2 c' / numel), e = the forward-time codebook.  The reference materialises the [N, D] differences, permutes them
// and index_add_s them; here the EMA kernel's scheme applies: [64 channel x 64 token] tiles transposed through LDS, equal codes of
// a tile summed first (leader + member mask), one row of float atomics per distinct code and tile.  K <= 8192 (the LDS table).
__global__ __launch_bounds__(256) void codebook_grad_kernel(const float *__restrict__ z, const float *__restrict__ E,
                                                            const long long *__restrict__ codes, const float *__restrict__ mask,
                                                            const float *__restrict__ g_loss, float coef_scale,
                                                            int D, int HW, long N, int K, float *__restrict__ gw)
{
    __shared__ float tile[64][65];
    __shared__ int code_s[64];
    __shared__ float m_s[64];
    __shared__ unsigned mask_s[64][2];
    extern __shared__ int first_s[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tok0 = (long)blockIdx.x * 64;
    for (int i = threadIdx.x; i < K; i += 256) first_s[i] = 64;
    if (threadIdx.x < 64) { mask_s[threadIdx.x][0] = 0u; mask_s[threadIdx.x][1] = 0u; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const long n = tok0 + threadIdx.x;
        long long cj = (n < N) ? codes[n] : -1;
        const bool ok = cj >= 0 && cj < K;
        code_s[threadIdx.x] = ok ? (int)cj : -1;
        m_s[threadIdx.x] = (ok && mask != nullptr) ? mask[n] : 1.0f;
        if (ok) atomicMin(&first_s[(int)cj], (int)threadIdx.x);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cj = code_s[threadIdx.x];
        if (cj >= 0) atomicOr(&mask_s[first_s[cj]][threadIdx.x >> 5], 1u << (threadIdx.x & 31));
    }
    const float c = -__fmul_rn(g_loss[0], coef_scale);
    const long n = tok0 + lane;
    const long nn = (n < N) ? n : N - 1;
    const long b = nn / HW;
    const int hw = (int)(nn - b * HW);
    const float *zp = z + (size_t)b * D * HW + hw;
    for (int c0 = 0; c0 < D; c0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = 16 * wave + i;
            tile[ch][lane] = (c0 + ch < D) ? zp[(size_t)(c0 + ch) * HW] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int tk = 16 * wave + i;
            unsigned m0 = __builtin_amdgcn_readfirstlane(mask_s[tk][0]), m1 = __builtin_amdgcn_readfirstlane(mask_s[tk][1]);
            if ((m0 | m1) == 0u) continue;
            const int cj = code_s[tk];
            const float e = (c0 + lane < D) ? E[(size_t)cj * D + c0 + lane] : 0.0f;
            float sum = 0.0f;
            while (m0) { const int t = __builtin_ctz(m0); m0 &= m0 - 1; sum += (tile[lane][t] - e) * m_s[t]; }
            while (m1) { const int t = __builtin_ctz(m1); m1 &= m1 - 1; sum += (tile[lane][32 + t] - e) * m_s[32 + t]; }
            if (c0 + lane < D) atomicAdd(&gw[(size_t)cj * D + c0 + lane], c * sum);
        }
    }
}

int dvq_launch_codebook_grad(const float *z, const float *E, const long long *codes, const float *mask, const float *g_loss,
                             float coef_scale, int D, int HW, long N, int K, float *gw, hipStream_t st)
{
    if (K > 8192) return -1000;
    hipLaunchKernelGGL(codebook_grad_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), (size_t)K * sizeof(int), st, z, E, codes,
                       mask, g_loss, coef_scale, D, HW, N, K, gw);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void ema_zero_kernel(float *__restrict__ a, size_t na, float *__restrict__ b, size_t nb)
{
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = i0; i < na; i += stride) a[i] = 0.0f;
    for (size_t i = i0; i < nb; i += stride) b[i] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// The same statistics with a BIG tile (round 5): 1024 tokens x 32 channels per workgroup, the tile in LDS (128 KiB: one workgroup per
// CU), the tokens of the tile chained by code (LDS atomicExch on a [K] head table), one float-atomic row per (code PRESENT in the
// tile, 32 channels).  The kernel above is bound by the L2 float-atomic rate (67 M adds at B = 256 = 308 us: 0.87 TB/s of the guide's
// 1.3 TB/s for 256-byte contiguous adds) and a 64-token tile has nothing to combine when the codes are spread (62 distinct codes of
// 64); 1024 tokens hold K (1 - e^-1) = 647 distinct codes of K = 1024, so 37 % of the adds disappear -- more with the 2 x 2 / 4 x 4
// copies of coarse cells and a skewed codebook.  Needs HW % 4 == 0 (16-byte loads along the tokens), D % 32 == 0, K <= 4096.
// ---------------------------------------------------------------------------------------------
#ifndef EMA_BIG
#define EMA_BIG 1                // 0: the 64-token kernel everywhere (A/B)
#endif
#define EMA_BT 1024              // tokens per tile
#define EMA_BC 32                // channels per tile
#define EMA_BSTR (EMA_BT + 4)    // floats per channel row in LDS
#ifndef EMA_BTHREADS
#define EMA_BTHREADS 1024         // 16 waves: the chain walk below is a chain of dependent LDS reads, it lives on waves in flight
#endif
__global__ __launch_bounds__(EMA_BTHREADS) void ema_accumulate_big_kernel(const float *__restrict__ z, const long long *__restrict__ codes,
                                                                 int D, int HW, long N, int K, float *__restrict__ cluster_size,
                                                                 float *__restrict__ vectors_sum)
{
    constexpr int NT = EMA_BTHREADS, NWV = NT / 64, PPW = 128 / NWV;     // 128 wave-pieces of the tile, PPW per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = (float *)smem;                                         // [EMA_BC][EMA_BSTR]
    short *nxt = (short *)(smem + (size_t)EMA_BC * EMA_BSTR * 4);        // [EMA_BT] next token of the same code, -1 = end
    int *head = (int *)((char *)nxt + EMA_BT * 2);                       // [K] last token of the tile with this code, -1 = none
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // persistent: a workgroup walks items (token block, 32-channel slice) with stride gridDim.x, and the NEXT item's tile is on
    // its way into registers while this item's chains are walked (one workgroup per CU: nobody else would hide the loads)
    const int nslice = D / EMA_BC;
    const long nitems = ((N + EMA_BT - 1) / EMA_BT) * nslice;
    f32x4 v[PPW];
    auto fetch = [&](long item) {
        const long tok0 = (item / nslice) * EMA_BT;
        const int c0 = (int)(item % nslice) * EMA_BC;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;                                // piece: channel q / 4, tokens (q % 4) * 256 + 4 lane ..
            const int ch = q >> 2, t4 = (q & 3) * 256 + 4 * lane;
            long n = tok0 + t4;
            n = n < N ? n : (N - 4 > 0 ? N - 4 : 0);                     // (HW % 4 == 0: N is a multiple of 4)
            const long b = n / HW;
            const int hw = (int)(n - b * HW);
            v[i] = __builtin_nontemporal_load((const f32x4 *)(z + ((size_t)b * D + c0 + ch) * HW + hw));
        }
    };
    long item = blockIdx.x;
    if (item < nitems) fetch(item);
    for (; item < nitems; item += gridDim.x) {
        const long tok0 = (item / nslice) * EMA_BT;
        const int c0 = (int)(item % nslice) * EMA_BC;
        for (int i = tid; i < K; i += NT) head[i] = -1;
        __syncthreads();                                                 // head[] is initialised (and the previous item's walk is over)
        for (int t = tid; t < EMA_BT; t += NT) {
            const long n = tok0 + t;
            const long long cj = (n < N) ? codes[n] : -1;
            short nx = (short)-2;                                        // not in any chain
            if (cj >= 0 && cj < K) nx = (short)atomicExch(&head[(int)cj], t);
            nxt[t] = nx;
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;
            const int ch = q >> 2, t4 = (q & 3) * 256 + 4 * lane;
            *(f32x4 *)(tile + ch * EMA_BSTR + t4) = v[i];
        }
        __syncthreads();
        if (item + gridDim.x < nitems) fetch(item + gridDim.x);
        // one (code, channel) per lane: 2 codes x 32 channels per wave-instruction; walk the code's chain, ONE atomic per present
        // code.  Four codes per lane at a time: the walks are chains of dependent LDS reads (head -> value, next -> ...),
        // interleaved they overlap each other's latency.
        const int ch = lane & 31, hsel = lane >> 5;
        const float *row = tile + ch * EMA_BSTR;
        constexpr int STEP = 2 * NWV;
        const bool count_here = (item % nslice) == 0;
        for (int k0 = 2 * wave + hsel; k0 < K; k0 += 4 * STEP) {
            int t[4], cnt[4];
            float sum[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * STEP;
                t[u] = k < K ? head[k] : -1;
                sum[u] = 0.0f;
                cnt[u] = 0;
            }
            while (t[0] >= 0 || t[1] >= 0 || t[2] >= 0 || t[3] >= 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t[u] >= 0) { sum[u] += row[t[u]]; t[u] = nxt[t[u]]; ++cnt[u]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * STEP;
                if (cnt[u] > 0) {
                    atomicAdd(&vectors_sum[(size_t)k * D + c0 + ch], sum[u]);
                    if (count_here && ch == 0) atomicAdd(&cluster_size[k], (float)cnt[u]);
                }
            }
        }
        __syncthreads();                                                 // tile / head / nxt are free for the next item
    }
}

int dvq_launch_ema_accumulate(const float *z, const long long *codes, int D, int HW, long N, int K,
                              float *cluster_size, float *vectors_sum, hipStream_t st)
{
    // zeroed by a kernel, not hipMemsetAsync: memset nodes misbehave under hipGraph replay on ROCm 7.2
    {
        size_t n = (size_t)K * D;
        int blocks = (int)((n / 4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(ema_zero_kernel, dim3(blocks), dim3(256), 0, st, cluster_size, (size_t)K, vectors_sum, n);
    }
    if (K <= 4096 && (HW & 3) == 0 && (D & 31) == 0 && N >= 64 * EMA_BT && EMA_BIG) {
        static unsigned long long done = 0;
        const size_t shm = (size_t)EMA_BC * EMA_BSTR * 4 + EMA_BT * 2 + (size_t)K * 4;
        int rc = dvq_allow_dynamic_lds((const void *)ema_accumulate_big_kernel, (int)shm, &done);
        if (rc) return rc;
        int ncu = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const long nitems = ((N + EMA_BT - 1) / EMA_BT) * (D / EMA_BC);
        const unsigned grid = (unsigned)(nitems < ncu ? nitems : (ncu > 0 ? ncu : 256));
        hipLaunchKernelGGL(ema_accumulate_big_kernel, dim3(grid), dim3(EMA_BTHREADS), shm, st, z,
                           codes, D, HW, N, K, cluster_size, vectors_sum);
    } else if (K <= 8192)
        hipLaunchKernelGGL(ema_accumulate_kernel<true>, dim3((unsigned)((N + 63) / 64)), dim3(256), (size_t)K * sizeof(int), st, z,
                           codes, D, HW, N, K, cluster_size, vectors_sum);
    else
        hipLaunchKernelGGL(ema_accumulate_kernel<false>, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, z, codes, D, HW, N, K,
                           cluster_size, vectors_sum);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Dead-code restart: which K of the n input vectors replace dead codes.  The reference takes `torch.randperm(n_vectors)[:K]`
// (quantize2_mask.py:93-96): on the GPU that is a sort of n = 262 144 keys, ~115 us of a 1-ms training step, to keep 1 024 of them.
// The first K entries of a uniform random permutation are K draws without replacement; drawing independently and keeping first
// occurrences is the same distribution.  One workgroup: 2K counter-based draws (splitmix64 of seed + i, multiply-high into
// [0, n)), an LDS hash table that keeps for every value its smallest draw index, a block scan over the "first occurrence" flags,
// the first K survivors in draw order.  Fewer than K distinct values among 2K draws (n >= 16 K: never in practice) leaves the
// missing slots at their own index.  K <= 2048.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(1024) void restart_pick_kernel(unsigned long long seed, long long n, int k, int hbits,
                                                            long long *__restrict__ out)
{
    extern __shared__ unsigned long long tab[];              // [1 << hbits] (value << 32 | smallest draw index), ~0 = empty
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = 1 << hbits, M = 2 * k;
    for (int i = tid; i < H; i += 1024) tab[i] = ~0ull;
    for (int i = tid; i < k; i += 1024) out[i] = i;          // (a slot no survivor reaches keeps a valid index)
    __syncthreads();
    const int per = (M + 1023) / 1024;                       // consecutive draws per thread: draw order = thread order
    unsigned d[4];
    bool keep[4];
    for (int j = 0; j < per; ++j) {
        const int i = tid * per + j;
        d[j] = (unsigned)__umul64hi(splitmix64(seed + (unsigned long long)i), (unsigned long long)n);
        keep[j] = false;
        if (i >= M) continue;
        const unsigned long long packed = ((unsigned long long)d[j] << 32) | (unsigned)i;
        unsigned slot = (unsigned)(splitmix64(d[j]) >> 40) & (H - 1);
        for (;;) {
            const unsigned long long cur = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == ~0ull) {
                if (atomicCAS(&tab[slot], ~0ull, packed) == ~0ull) break;
            } else if ((unsigned)(cur >> 32) == d[j]) {
                atomicMin(&tab[slot], packed);
                break;
            } else {
                slot = (slot + 1) & (H - 1);
            }
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int j = 0; j < per; ++j) {
        const int i = tid * per + j;
        if (i >= M) continue;
        unsigned slot = (unsigned)(splitmix64(d[j]) >> 40) & (H - 1);
        while ((unsigned)(tab[slot] >> 32) != d[j]) slot = (slot + 1) & (H - 1);
        keep[j] = (unsigned)tab[slot] == (unsigned)i;         // this draw is the value's first occurrence
        cnt += keep[j];
    }
    int incl = cnt;                                           // block-wide exclusive scan of cnt
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int rank = base + incl - cnt;
    for (int j = 0; j < per; ++j)
        if (keep[j]) { if (rank < k) out[rank] = (long long)d[j]; ++rank; }
}

int dvq_launch_restart_pick(unsigned long long seed, long long n, int k, long long *out, hipStream_t st)
{
    if (k < 1 || k > 2048 || n < 1 || n > 0xFFFFFFFFll) return -1000;
    int hbits = 3;
    while ((1 << hbits) < 8 * k) ++hbits;                    // table = 4 x the draws
    static unsigned long long done = 0;
    const size_t shm = (size_t)(1 << hbits) * sizeof(unsigned long long);
    int rc = dvq_allow_dynamic_lds((const void *)restart_pick_kernel, (int)shm, &done);
    if (rc) return rc;
    hipLaunchKernelGGL(restart_pick_kernel, dim3(1), dim3(1024), shm, st, seed, n, k, hbits, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Everything of the training-mode codebook update AFTER the statistics, as one kernel (quantize2_mask.py:89-115): the two EMA
// updates (`cluster_size_ema.mul_(decay).add_(..., alpha = 1 - decay)`, the same for `embed_ema`), the dead-code restart (codes
// whose updated count fell below 1 take a restart vector and count 1) and `_update_embedding` (n = sum of the counts,
// weight = embed_ema / (n (count + eps) / (n + K eps))).  As torch ops these were ~20 launches of 4-5 us on K x D = 1 MiB of data
// -- 65 us of kernels and as much again in gaps per 0.8-ms training step (profiles/r05_train_step.json).
// One wave per code row.  Every workgroup computes n itself from the OLD counts (4 KiB of reads): the new counts therefore go to a
// separate array (`cluster_size_out`; the caller copies it over the buffer afterwards) -- written in place, a fast workgroup's
// new count would enter a slow one's sum decayed twice.  embed_ema is updated in place (a row has one owner).
// Restart rows: from `restart_rows` [K, D] (data-parallel runs broadcast rank 0's), or gathered here from the NCHW latents at
// token index pick[j].  fp32 operation order as the reference's expressions; the sum n is accumulated in double (the reference's
// fp32 tree differs by ~1e-7 relative: tolerance parity, 1e-5).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_update_kernel(const float *__restrict__ stats_sum, const float *__restrict__ stats_count,
                                                         float decay, float alpha, float eps, int K, int D,
                                                         const float *__restrict__ cs_old, float *__restrict__ cs_new,
                                                         float *__restrict__ embed_ema, float *__restrict__ weight, int restart,
                                                         const float *__restrict__ restart_rows, const float *__restrict__ z, int HW,
                                                         const long long *__restrict__ pick)
{
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto new_count = [&](int j, bool &dead) -> float {
        const float c = __fadd_rn(__fmul_rn(cs_old[j], decay), __fmul_rn(alpha, stats_count[j]));
        dead = restart != 0 && c < 1.0f;
        return dead ? 1.0f : c;
    };
    double s = 0.0;
    for (int j = tid; j < K; j += 256) { bool dd; s += (double)new_count(j, dd); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float n = (float)((red[0] + red[1]) + (red[2] + red[3]));
    const float denom = __fadd_rn(n, __fmul_rn((float)K, eps));
    const int j = blockIdx.x * 4 + wave;
    if (j >= K) return;
    bool dead;
    const float c = new_count(j, dead);
    if (lane == 0) cs_new[j] = c;
    const float norm = __fmul_rn(n, __fadd_rn(c, eps)) / denom;
    const float *rr = nullptr;
    size_t rstride = 1;
    if (dead) {
        if (restart == 1) rr = restart_rows + (size_t)j * D;
        else { const long long p = pick[j]; const long long b = p / HW; rr = z + ((size_t)b * D) * HW + (size_t)(p - b * HW); rstride = (size_t)HW; }
    }
    for (int ch = lane; ch < D; ch += 64) {
        const size_t i = (size_t)j * D + ch;
        float e = __fadd_rn(__fmul_rn(embed_ema[i], decay), __fmul_rn(alpha, stats_sum[i]));
        if (dead) e = rr[(size_t)ch * rstride];
        embed_ema[i] = e;
        weight[i] = e / norm;
    }
}

int dvq_launch_ema_update(const float *stats_sum, const float *stats_count, float decay, float eps, int K, int D,
                          const float *cs_old, float *cs_new, float *embed_ema, float *weight, int restart, const float *restart_rows,
                          const float *z, int HW, const long long *pick, hipStream_t st)
{
    const float alpha = (float)(1.0 - (double)decay);        // what `alpha = 1 - self.decay` (Python doubles) becomes as an fp32 scalar
    hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, st, stats_sum, stats_count, decay, alpha, eps, K,
                       D, cs_old, cs_new, embed_ema, weight, restart, restart_rows, z, HW, pick);
    return (int)hipGetLastError();
}
