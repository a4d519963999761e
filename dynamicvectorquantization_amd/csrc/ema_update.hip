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
// The same statistics with a BIG tile whose tokens are SORTED by code (N >= 65 536 tokens, K <= 4096, HW % 4 == 0, D % 16 == 0).
// The kernel above is bound by the L2 float-atomic rate (67 M adds at B = 256 = 308 us) and a 64-token tile has nothing to combine
// when the codes are spread.  Round 5 took 1024 tokens x 32 channels (128 KiB of LDS, one 16-wave workgroup per CU) with the tile's
// tokens chained by code (LDS atomicExch on a head table): 647 distinct codes of K = 1024 per tile, 37 % fewer adds, 205 us.
// Round 6: 2048 tokens x 16 channels (the same 128 KiB): 885 distinct codes -- 57 % of the adds disappear -- and a (code, 16
// channels) row of atomics is one full 64-byte request.  With linked chains that shape ran 192 -> 127 us on uniform codes but
// 208 -> 311 us with 10 % of the tokens on one code (a 205-link chain is 205 dependent LDS round trips on ONE lane group while the
// workgroup waits), so the tile's tokens are counting-sorted by code in LDS instead (count with a returning LDS atomic = rank inside
// the code, exclusive scan over the K counts, scatter), once per token block and reused by the slices a workgroup walks of it: a
// code's tokens are a contiguous index range, read four at a time (two LDS round trips per four tokens), and a code with more than
// EMS_LONG tokens in the tile is summed by ALL lane groups and added up in LDS -- ONE row of atomics per (hot code, item).
// Measured, B = 256, K = 1024 (tools/ema_time.py, same box; round-5 kernel -> this one): uniform codes 194 -> 126 us, dual-grain
// codes 168 -> 127, 10 % of the tokens on one code 209 -> 123, 50 % 664 -> 117, one code for everything 1230 -> 77.
// ---------------------------------------------------------------------------------------------
#define EMS_BT 2048
#define EMS_BC 16
#define EMS_STR (EMS_BT + 4)
static_assert(EMS_BC == 16 && EMS_BT % 256 == 0 && EMS_BT * EMS_BC == 32768, "the walk's lane mapping (four 16-channel groups per wave) and the 128 load pieces assume this tile");
#define EMS_LONG 128             // a code with more tokens than this in a tile is summed by ALL lane groups, 32 tokens each
__global__ __launch_bounds__(1024) void ema_accumulate_sorted_kernel(const float *__restrict__ z, const long long *__restrict__ codes,
                                                                    int D, int HW, long N, int K, float *__restrict__ cluster_size,
                                                                    float *__restrict__ vectors_sum)
{
    constexpr int NT = 1024, NWV = NT / 64, PPW = 128 / NWV, PPC = EMS_BT / 256, CPI = 64 / EMS_BC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = (float *)smem;                                         // [EMS_BC][EMS_STR]
    short *order = (short *)(smem + (size_t)EMS_BC * EMS_STR * 4);       // [EMS_BT] the tile's tokens, by code
    short *rank = order + EMS_BT;                                        // [EMS_BT] rank of a token among the tile's tokens of its code
    short *code_s = rank + EMS_BT;                                       // [EMS_BT] code, -1 = none
    int *start = (int *)(code_s + EMS_BT);                               // [K + 1] counts, then offsets into order[]
    __shared__ int wtot[NWV];
    __shared__ int nlong, longk[EMS_BT / EMS_LONG];                      // codes with more than EMS_LONG tokens in the tile
    __shared__ float red[NWV * EMS_BC];                                  // their per-wave partial sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nslice = D / EMS_BC;
    const long nitems = ((N + EMS_BT - 1) / EMS_BT) * nslice;            // item = (token block, 16-channel slice), token-block major
    const long per = (nitems + gridDim.x - 1) / gridDim.x;
    const long item0 = (long)blockIdx.x * per, item1 = item0 + per < nitems ? item0 + per : nitems;
    f32x4 v[PPW];
    // the slices of a token block in an order rotated by the block's index (a bijection per block, so the workgroups that share a
    // block still cover disjoint slices): workgroups walk their blocks in step, and without the rotation every one of them would
    // be adding into the SAME 16 channels of the code rows at the same time (atomics on one address serialise)
    auto slice_of = [&](long item) -> int { return (int)((item % nslice + 5 * (item / nslice)) % nslice); };
    auto fetch = [&](long item) {
        const long tok0 = (item / nslice) * EMS_BT;
        const int c0 = slice_of(item) * EMS_BC;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;                                // piece: channel q / PPC, tokens (q % PPC) * 256 + 4 lane ..
            const int ch = q / PPC, t4 = (q % PPC) * 256 + 4 * lane;
            long n = tok0 + t4;
            n = n < N ? n : (N - 4 > 0 ? N - 4 : 0);                     // (HW % 4 == 0: N is a multiple of 4)
            const long b = n / HW;
            const int hw = (int)(n - b * HW);
            v[i] = __builtin_nontemporal_load((const f32x4 *)(z + ((size_t)b * D + c0 + ch) * HW + hw));
        }
    };
    if (item0 < item1) fetch(item0);
    long sorted_tb = -1;
    for (long item = item0; item < item1; ++item) {
        const long tb = item / nslice;
        const long tok0 = tb * EMS_BT;
        const int c0 = slice_of(item) * EMS_BC;
        if (tb != sorted_tb) {                                           // (the previous item's walk ended with a barrier)
            for (int i = tid; i <= K; i += NT) start[i] = 0;
            if (tid == 0) nlong = 0;
            __syncthreads();
            for (int t = tid; t < EMS_BT; t += NT) {
                const long n = tok0 + t;
                const long long cj = (n < N) ? codes[n] : -1;
                const bool ok = cj >= 0 && cj < K;
                code_s[t] = ok ? (short)cj : (short)-1;
                rank[t] = ok ? (short)atomicAdd(&start[(int)cj], 1) : (short)0;
            }
            __syncthreads();
            // exclusive scan of the K counts in place (a thread owns four consecutive entries); start[K] = the tile's valid tokens
            {
                int a[4], tot = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * tid + j;
                    a[j] = k < K ? start[k] : 0;
                    tot += a[j];
                    if (a[j] > EMS_LONG) longk[atomicAdd(&nlong, 1)] = k;     // (at most EMS_BT / EMS_LONG of them)
                }
                int inc = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
                if (lane == 63) wtot[wave] = inc;
                __syncthreads();
                int base = inc - tot;
                for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int k = 4 * tid + j; if (k < K) start[k] = base; base += a[j]; }
                if (tid == NT - 1) start[K] = base;                      // (4 * 1024 >= K: the last thread's running sum is the total)
            }
            __syncthreads();
            for (int t = tid; t < EMS_BT; t += NT) {
                const int c = code_s[t];
                if (c >= 0) order[start[c] + rank[t]] = (short)t;
            }
            sorted_tb = tb;
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;
            const int ch = q / PPC, t4 = (q % PPC) * 256 + 4 * lane;
            *(f32x4 *)(tile + ch * EMS_STR + t4) = v[i];
        }
        __syncthreads();                                                 // tile and order[] are complete
        if (item + 1 < item1) fetch(item + 1);                           // the next tile is on its way while this one is walked
        // lane = (code of the instruction's four, channel); four instructions' codes per lane at a time (independent LDS loads)
        const int ch = lane % EMS_BC, hsel = lane / EMS_BC;
        const float *row = tile + ch * EMS_STR;
        constexpr int STEP = CPI * NWV;
        const bool count_here = c0 == 0;
        for (int k0 = CPI * wave + hsel; k0 < K; k0 += 4 * STEP) {
            int s[4], e[4], n0[4];
            float sum[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * STEP;
                s[u] = k < K ? start[k] : 0;
                e[u] = k < K ? start[k + 1] : 0;
                n0[u] = e[u] - s[u];
                if (n0[u] > EMS_LONG) { e[u] = s[u]; n0[u] = 0; }     // a long run: everybody's work, below
                sum[u] = 0.0f;
            }
            // four tokens of a code per step: their four index reads are independent of each other, and so are the four value reads
            // behind them (two LDS round trips per four tokens, however long the run)
            while (s[0] < e[0] || s[1] < e[1] || s[2] < e[2] || s[3] < e[3]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (s[u] < e[u]) {
                        const int last = e[u] - 1;
                        const int i1 = s[u] + 1 < last ? s[u] + 1 : last, i2 = s[u] + 2 < last ? s[u] + 2 : last,
                                  i3 = s[u] + 3 < last ? s[u] + 3 : last;
                        const int t0 = order[s[u]], t1 = order[i1], t2 = order[i2], t3 = order[i3];
                        const float r0 = row[t0], r1 = row[t1], r2 = row[t2], r3 = row[t3];
                        const int left = e[u] - s[u];
                        sum[u] += (r0 + (left > 1 ? r1 : 0.0f)) + ((left > 2 ? r2 : 0.0f) + (left > 3 ? r3 : 0.0f));
                        s[u] += 4;
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * STEP;
                if (n0[u] > 0) {
                    atomicAdd(&vectors_sum[(size_t)k * D + c0 + ch], sum[u]);
                    if (count_here && ch == 0) atomicAdd(&cluster_size[k], (float)n0[u]);
                }
            }
        }
        // the long runs (a hot code of a collapsing codebook: more than EMS_LONG of the tile's 2048 tokens): every lane group sums 32-token
        // pieces of the run, the workgroup adds the pieces up in LDS, ONE row of atomics per (code, item) -- atomics on one address
        // serialise at the memory side, and with a hot code every workgroup of the launch is adding into the same row
        const int nl = nlong;
        for (int li = 0; li < nl; ++li) {
            const int k = longk[li];
            const int s0 = start[k], e0 = start[k + 1];
            float sm = 0.0f;
            for (int p0 = s0 + 32 * (CPI * wave + hsel); p0 < e0; p0 += 32 * CPI * NWV) {
                const int p1 = p0 + 32 < e0 ? p0 + 32 : e0;
                for (int i = p0; i < p1; i += 4) {
                    const int last = p1 - 1;
                    const int i1 = i + 1 < last ? i + 1 : last, i2 = i + 2 < last ? i + 2 : last, i3 = i + 3 < last ? i + 3 : last;
                    const int t0 = order[i], t1 = order[i1], t2 = order[i2], t3 = order[i3];
                    const float r0 = row[t0], r1 = row[t1], r2 = row[t2], r3 = row[t3];
                    const int left = p1 - i;
                    sm += (r0 + (left > 1 ? r1 : 0.0f)) + ((left > 2 ? r2 : 0.0f) + (left > 3 ? r3 : 0.0f));
                }
            }
            sm += __shfl_xor(sm, 16);                                    // the wave's four lane groups (EMS_BC = 16 channels each)
            sm += __shfl_xor(sm, 32);
            if (lane < EMS_BC) red[wave * EMS_BC + lane] = sm;
            __syncthreads();
            if (tid < EMS_BC) {
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) t += red[w * EMS_BC + tid];
                atomicAdd(&vectors_sum[(size_t)k * D + c0 + tid], t);
                if (count_here && tid == 0) atomicAdd(&cluster_size[k], (float)(e0 - s0));
            }
            __syncthreads();                                             // red[] is free again
        }
        __syncthreads();                                                 // tile / order / start are free for the next item
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
    if (K <= 4096 && (HW & 3) == 0 && (D & 15) == 0 && N >= 32 * EMS_BT) {
        static unsigned long long done_s = 0;
        const size_t shm = (size_t)EMS_BC * EMS_STR * 4 + (size_t)EMS_BT * 6 + ((size_t)K + 1) * 4;
        int rc = dvq_allow_dynamic_lds((const void *)ema_accumulate_sorted_kernel, (int)shm, &done_s);
        if (rc) return rc;
        int ncu = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const long nitems = ((N + EMS_BT - 1) / EMS_BT) * (D / EMS_BC);
        const unsigned grid = (unsigned)(nitems < ncu ? nitems : (ncu > 0 ? ncu : 256));
        hipLaunchKernelGGL(ema_accumulate_sorted_kernel, dim3(grid), dim3(1024), shm, st, z, codes, D, HW, N, K, cluster_size,
                           vectors_sum);
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
