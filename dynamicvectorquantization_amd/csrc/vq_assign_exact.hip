// vq_assign_exact.hip -- bit-exact nearest-codebook assignment on the fp32 matrix cores (gfx950).
//
// Replaces VQEmbedding.compute_distances + find_nearest_embedding + embed and the
// VectorQuantize2.forward glue around them (reference modules/vector_quantization/
// quantize2_mask.py:29-55,157-191; quantize_vqgan.py:271-312).
//
// Arithmetic contract (what torch-CPU produces, pinned by oracle/ + tests/golden):
//   dot   = sequential k = 0..D-1 fp32 FMA chain from 0.  v_mfma_f32_32x32x2_f32 IS that chain,
//           two k per instruction (D = fma(a1,b1, fma(a0,b0, C)), one rounding per product), so
//           D/2 chained MFMAs give 32x32 bit-exact dots.
//   xn,en = ATen-order sum of squares (32 strided partials, fixed combine order)
//   d     = fl(fl(xn + en) - 2 dot);  argmin first index, NaN = minimum
//   z_q   = fl(z + fl(e - z))
//
// Mapping: MFMA rows (A operand) = 32 codes, columns (B operand) = 32 tokens; lane (c, h) holds
// token column c and, in the accumulator, 16 code rows {(r&3) + 8(r>>2) + 4h}, so the running
// argmin is lane-local and only the two lane halves are merged at the end.
// A wave keeps its 32 tokens' D channels in registers (D/2 VGPRs, read straight from NCHW: lanes
// 0-31 / 32-63 read two 128-B runs of consecutive tokens per load) and streams the codebook as
// prepared 32-code LDS tile images (double-buffered global->LDS DMA, one barrier per tile).
// Compute-bound on the fp32 MFMA rate (2*K*D flop per token).
#include "dvq_filter.h"
#include <type_traits>

// ROUTED: token = output position of a routed batch (dvq_filter.h: DvqRouted), read from the encoder branch that
// won its cell.
// FOLDCONV (list mode of the filter path behind a 1x1 quant_conv -- folded into the codebook, vq_fold.hip, or computed in pass 1's
// prologue): z / the branches hold the conv's INPUT; a wave first
// computes h = W x + bias for its 32 tokens (qconv.hip's split-fp16 arithmetic on the matrix cores, weight fragments straight
// from the L2-resident images; bit-identical to dvq_qconv_f32 and to the resolver's h) and moves the accumulators into the
// (even / odd channel per lane half) layout of zr by one cross-half exchange per register pair.  A rare path: one workgroup
// per CU (512 registers), no attempt at overlap.
template <int D, bool LIST, bool ROUTED, bool FOLDCONV = false>
__global__ __launch_bounds__(256, FOLDCONV ? 1 : 2) void vq_assign_exact_kernel(
    const float *__restrict__ z, const float *__restrict__ tiles, const float *__restrict__ E,
    const float *__restrict__ mask, int HW, int K, long N,
    float *__restrict__ zq, long long *__restrict__ codes, double *__restrict__ partials,
    const int *__restrict__ list, const int *__restrict__ list_count,
    DvqLossTail tail, const DvqRouted rv, const DvqConv cv)
{
    constexpr int S = D / 2;                         // MFMA steps (2 k each)
    constexpr int TILE_FLOATS = 32 * D + 64;
    constexpr int CHUNKS_PER_WAVE = (32 * D * 4 / 1024) / 4;   // 1-KiB DMA pieces per wave per tile
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * TILE_FLOATS (+ 4 doubles)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    // tokens: the dense range [0, N) (one 128-token chunk per block), or (pass 2 of the filter path)
    // the entries of a work list, which a small grid walks in 128-entry chunks
    const int cnt = LIST ? *list_count : 0;
    // list mode: only the blocks that have list entries take part (block 0 always does, it may have to
    // finalize on its own) -- an empty list costs one block, not a grid of ticket atomics
    int nwork = 1;
    if (LIST) {
        nwork = (cnt + 127) / 128;
        nwork = nwork < (int)gridDim.x ? nwork : (int)gridDim.x;
        nwork = nwork > 1 ? nwork : 1;
        if ((int)blockIdx.x >= nwork) return;
    }
    double block_sum = 0.0;
    long chunk = blockIdx.x;
    for (int it = 0; LIST ? chunk * 128 < cnt : it < 1; ++it, chunk += gridDim.x) {   // dense: exactly once
    long n = (chunk * 4 + wave) * 32 + c;
    bool valid = n < N;
    long nn = valid ? n : N - 1;
    if (LIST) {
        valid = n < cnt;
        n = list[valid ? n : 0];
        nn = n;
    }
    const float *zp;                                   // channel h of the token; channel k = 2s + h at zp + 2s*stride
    int stride, rep = 1, Wout = 0, HWo = HW;
    if (ROUTED) {
        const DvqTok tk = dvq_routed_lookup(rv, nn);          // output position -> address in the branch that won its cell
        valid = valid && tk.valid;
        stride = tk.stride;
        zp = tk.src + (size_t)h * stride;
        n = nn = tk.n;
        Wout = rv.Wout;
        HWo = rv.HWout;
    } else {
        const long b = nn / HW;
        stride = HW;
        zp = z + ((size_t)b * D + h) * HW + (size_t)(nn - b * HW);
    }
    const long bout = nn / HWo;
    const size_t zqbase = ((size_t)bout * D + h) * HWo + (size_t)(nn - bout * HWo);

    float zr[S];
    if constexpr (FOLDCONV) {
        constexpr int S16 = D / 16, T8 = D / 32;
        constexpr int QIMG = S16 * 1024, QTILE = 2 * QIMG + 256;
        const float *xb = zp - (size_t)h * stride;            // channel 0 of the token
        f16x8 xh[S16], xl[S16];
        float unscale;
        {
            float xf[S16][8];
            float amax = 0.0f;
#pragma unroll
            for (int s = 0; s < S16; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xf[s][j] = xb[(size_t)(16 * s + 8 * h + j) * stride];
                    amax = vmax_abs(amax, xf[s][j]);
                }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            int ea = 0;
            if (amax > 0.0f && amax < __builtin_inff()) { int e; (void)frexpf(amax, &e); ea = 14 - e; }
            ea = ea > 100 ? 100 : (ea < -100 ? -100 : ea);
            const float sa = ldexpf(1.0f, ea);
            unscale = ldexpf(cv.meta->inv_scale_w, -ea);
#pragma unroll
            for (int s = 0; s < S16; ++s) {
                u32x4 ph, pl;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const float v0 = xf[s][2 * j2] * sa, v1 = xf[s][2 * j2 + 1] * sa;
                    const f32x2 vv = {v0, v1};
                    const f16x2 hh = __builtin_convertvector(vv, f16x2);
                    const f32x2 rr = {v0 - (float)hh[0], v1 - (float)hh[1]};
                    const f16x2 ll = __builtin_convertvector(rr, f16x2);
                    ph[j2] = __builtin_bit_cast(unsigned, hh);
                    pl[j2] = __builtin_bit_cast(unsigned, ll);
                }
                xh[s] = __builtin_bit_cast(f16x8, ph);
                xl[s] = __builtin_bit_cast(f16x8, pl);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t8 = 0; t8 < T8; ++t8) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < S16; ++s) {
                const char *wt = cv.wimg + (size_t)t8 * QTILE + s * 1024 + lane * 16;
                const f16x8 ah = *(const f16x8 *)wt, al = *(const f16x8 *)(wt + QIMG);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[s], acc, 0, 0, 0);     // small terms first (qconv.hip)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[s], acc, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // at most four k-steps of weight fragments in flight
            }
            // register r of lane half h holds channel 32 t8 + 16 (r >> 3) + 8 h + (r & 7); zr[s] of lane half h is channel
            // 2 s + h: of each register pair (j = 2m, 2m + 1) a lane keeps the channel of its own parity and hands the other
            // one to its partner lane (lane ^ 32), whose channel of that parity sits in the same pair 8 channels away
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int r0 = 8 * a + 2 * m, ch0 = 32 * t8 + 16 * a + 8 * h + 2 * m;
                    const float v0 = __builtin_fmaf(acc[r0], unscale, cv.bias[ch0]);
                    const float v1 = __builtin_fmaf(acc[r0 + 1], unscale, cv.bias[ch0 + 1]);
                    const float keep = h ? v1 : v0, send = h ? v0 : v1;
                    const float recv = __shfl_xor(send, 32);
                    zr[16 * t8 + 8 * a + m] = h ? recv : keep;
                    zr[16 * t8 + 8 * a + 4 + m] = h ? keep : recv;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the conv-fused op's test form: the h this kernel scores replaces what pass 1 wrote for the token)
        if (cv.h_all && cv.h_buf != nullptr && valid) {
            float *hp = cv.h_buf + zqbase;
#pragma unroll
            for (int s = 0; s < S; ++s) hp[(size_t)2 * s * HWo] = zr[s];
        }
    } else {
#pragma unroll
    for (int s = 0; s < S; ++s) zr[s] = zp[(size_t)2 * s * stride];
    }

    auto stage = [&](int t, float *buf) {
        const char *src = (const char *)(tiles + (size_t)t * TILE_FLOATS);
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_WAVE; ++i) {
            int chunk = wave * CHUNKS_PER_WAVE + i;
            glds16(src + chunk * 1024 + lane * 16, (char *)buf + chunk * 1024);
        }
        if (wave == 0) glds4(src + 32 * D * 4 + lane * 4, (char *)buf + 32 * D * 4);
    };

    const int T = dvq_num_tiles(K);
    if (LIST) __syncthreads();                // (next chunk) everyone is done with the buffers
    stage(0, lds);

    // ---- xn: ATen-order sum of squares of this token (both lanes of a token get the same value)
    float xn;
    {
        float p[16], o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float a = sq_rn(zr[q]);
#pragma unroll
            for (int j = 1; j < S / 16; ++j) a = __fadd_rn(a, sq_rn(zr[q + 16 * j]));
            p[q] = a;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = __shfl_xor(p[q], 32);
        float t[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            // a[m], m = l + 8g: even m lives in the h = 0 lane's p[m/2], odd m in the h = 1 lane's
            float a4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int m = l + 8 * g;
                float mine = p[m >> 1], other = o[m >> 1];
                a4[g] = ((m & 1) == h) ? mine : other;
            }
            t[l] = __fadd_rn(__fadd_rn(__fadd_rn(a4[0], a4[1]), a4[2]), a4[3]);
        }
        xn = t[0];
#pragma unroll
        for (int l = 1; l < 8; ++l) xn = __fadd_rn(xn, t[l]);
    }

    float best = __builtin_inff();
    int bidx = 0x7fffffff;

    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // tile t landed; everyone is done with tile t-1
        float *buf = lds + (t & 1) * TILE_FLOATS;
        if (t + 1 < T) stage(t + 1, lds + ((t + 1) & 1) * TILE_FLOATS);

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float *ap = buf + c * 8 + h * 4;
#pragma unroll
        for (int kg = 0; kg < D / 8; ++kg) {
            f32x4 a = *(const f32x4 *)(ap + kg * 256);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], zr[kg * 4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], zr[kg * 4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], zr[kg * 4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], zr[kg * 4 + 3], acc, 0, 0, 0);
        }
        const float *entile = buf + 32 * D + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 en4 = *(const f32x4 *)(entile + 8 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int r = g * 4 + q;
                int code = t * 32 + q + 8 * g + 4 * h;
                float bias = __fadd_rn(xn, en4[q]);
                float d = __builtin_fmaf(-2.0f, acc[r], bias);   // fl(bias - 2 dot), 2 dot exact
                bool take = argmin_take(d, best) && (code < K);
                best = take ? d : best;
                bidx = take ? code : bidx;
            }
        }
    }

    {   // merge the two lane halves of each token
        float ob = __shfl_xor(best, 32);
        int oi = __shfl_xor(bidx, 32);
        argmin_merge(best, bidx, ob, oi);
    }
    const int code = (bidx == 0x7fffffff) ? 0 : bidx;    // every distance +inf -> index 0
    if (valid && h == 0) {
        if (!ROUTED) {
            codes[n] = (long long)code;
        } else {
            for (int ry = 0; ry < rep; ++ry)
                for (int rx = 0; rx < rep; ++rx) codes[n + (long)ry * Wout + rx] = (long long)code;
        }
    }

    // ---- z_q = z + (e - z), loss partial sum((e - z)^2 * m)
    float lsum = 0.0f;
    if ((zq != nullptr || partials != nullptr) && valid) {
        const float *ep = E + (size_t)code * D + h;
        const float m = (mask != nullptr) ? mask[nn] : 1.0f;
        // the zq test is a scalar branch on the kernel argument, taken once (inside the loop, on the
        // per-lane pointer, it turned every store into its own exec-masked branch)
        auto finish = [&](auto store_tag) {
            constexpr bool STORE = decltype(store_tag)::value;
            float *zqp = STORE ? zq + zqbase : nullptr;
#pragma unroll
            for (int s = 0; s < S; ++s) {          // fully unrolled: zr[] must stay in registers
                float e = ep[2 * s];
                float diff = __fsub_rn(e, zr[s]);
                if (STORE) {
                    const float v = __fadd_rn(zr[s], diff);
                    if (!ROUTED) {
                        zqp[(size_t)2 * s * HWo] = v;
                    } else {
                        for (int ry = 0; ry < rep; ++ry)
                            for (int rx = 0; rx < rep; ++rx) zqp[(size_t)2 * s * HWo + (size_t)ry * Wout + rx] = v;
                    }
                }
                lsum = __fadd_rn(lsum, __fmul_rn(__fmul_rn(diff, diff), m));
            }
        };
        if (zq != nullptr) finish(std::true_type{});
        else finish(std::false_type{});
        if (ROUTED) lsum *= (float)(rep * rep);
    }
    block_sum += (double)lsum;
    }   // chunk loop
    if (partials != nullptr) {
        double ds = block_sum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off);
        __syncthreads();                       // all waves are done with the tile buffers
        double *red = (double *)lds;
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (tid == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    if (LIST && (tail.loss != nullptr || tail.counters != nullptr)) {
        // fused loss finalize: the block that takes the last ticket sums every partial of the op
        // (fixed order -> the result does not depend on which block that is)
        int *flag = (int *)lds + 16;
        __syncthreads();
        if (tid == 0) {
            if (nwork == 1) {
                *flag = 1;
            } else {
                __threadfence();
                int old = atomicAdd(tail.ticket, 1);
                *flag = (old == nwork - 1);
            }
        }
        __syncthreads();
        if (*flag && tid < DVQ_QSHARDS && tail.counters != nullptr) {
            // bookkeeping for dvq_vq_assign_fallback_count_offset, and the counter block goes back to zero: this is the op's last
            // workgroup (every workgroup that takes part has drawn its ticket, the others only ever read the list count, and
            // with either value -- the count or the zero written here -- they leave at once), so the NEXT op on this workspace
            // needs no zero kernel (DVQ_MODE_WS_CLEAN)
            int v = tail.counters[DVQ_QCOUNT0 + tid];
            v = v < tail.shard_cap ? v : tail.shard_cap;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            tail.counters[DVQ_QCOUNT0 + tid] = 0;
            if (tid == 0) {
                tail.counters[DVQ_C_QUEUED] = v;
                tail.counters[DVQ_C_NEXACT] = cnt;
                tail.counters[DVQ_C_EXACT] = 0;
                *tail.ticket = 0;
            }
        }
        if (*flag && tail.loss != nullptr) {
            if (nwork > 1) __threadfence();    // (an empty list -- the usual case -- has nobody else's stores to wait for: the fence
                                               // alone was ~2 us of this kernel's 5)
            // partials of the earlier kernels are plain memory by now; this kernel's own need
            // device-coherent loads (they were written by blocks on other XCDs)
            const int nprev = tail.nparts - (int)gridDim.x;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int i = tid;
            for (; i + 768 < nprev; i += 1024) {
                s0 += tail.partials[i]; s1 += tail.partials[i + 256];
                s2 += tail.partials[i + 512]; s3 += tail.partials[i + 768];
            }
            for (; i < nprev; i += 256) s0 += tail.partials[i];
            for (i = nprev + tid; i < nprev + nwork; i += 256)     // only the blocks that took part wrote theirs
                s1 += __hip_atomic_load(tail.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double s = (s0 + s1) + (s2 + s3);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            double *red = (double *)lds + 16;
            __syncthreads();
            if (lane == 0) red[wave] = s;
            __syncthreads();
            if (tid == 0) {
                float mean = (float)(((red[0] + red[1]) + (red[2] + red[3])) * tail.inv_numel);
                tail.loss[0] = mean;
                tail.loss[1] = __fadd_rn(__fmul_rn(tail.beta, mean), mean);
            }
        }
    }
}

// loss[0] = mean, loss[1] = fl(fl(beta*mean) + mean)   (quantize2_mask.py:175, quantize_vqgan.py:291)
__global__ void vq_loss_finalize_kernel(const double *__restrict__ partials, int nparts,
                                        double inv_numel, float beta, float *__restrict__ loss)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float mean = (float)(red[0] * inv_numel);
        loss[0] = mean;
        loss[1] = __fadd_rn(__fmul_rn(beta, mean), mean);
    }
}

// ---------------------------------------------------------------------------------------------
// codebook prep: f32 tile images + exact-order norms
// ---------------------------------------------------------------------------------------------
// One workgroup per CPW codes of a tile (8: a K = 1024 codebook is 128 workgroups; 32 from 256 tiles on).  A thread owns octets of
// channels of one code: coalesced row reads, 32-byte image stores.  The squared norms keep the ATen order of the oracle
// (oracle/dvq_oracle.c: dvq_oracle_sumsq): 32 accumulators a[m] = sum over k0 of e[k0 + m]^2 in k0 order, then
// ((a[l] + a[l+8]) + a[l+16]) + a[l+24] summed over l = 0..7 in order -- through LDS, one (code, m) pair per thread.
// Round 6: the scan the fp16 section needs (max |e|, max norm, finiteness: codebook_meta_partial_kernel until now, a second pass
// over the codebook) rides along: a workgroup parks (max |e|, max en, bad, 0) in the tile's 32 floats of padding, slot
// 32 D + 32 + 4 * (workgroup within the tile); no kernel reads the padding of the image.  The prep is rebuilt at every training step.
template <int CPW>
__global__ __launch_bounds__(256) void codebook_prep_f32_kernel(const float *__restrict__ E, int K, int D,
                                                                float *__restrict__ tiles, float *__restrict__ en_all,
                                                                float *__restrict__ etamax)
{
    extern __shared__ float sq[];                            // [CPW][D] squares, then [CPW][32] partial sums
    __shared__ float s_amax[4];
    __shared__ int s_bad[4];
    constexpr int SUBS = 32 / CPW;
    const int t = blockIdx.x / SUBS, sub = blockIdx.x % SUBS, tid = threadIdx.x;
    float *tile = tiles + (size_t)t * dvq_tile_floats(D);
    const int KG = D / 8;
    float amax = 0.0f;
    int bad = 0;
    if (blockIdx.x == 0 && tid == 0) *etamax = 0.0f;         // the fp16 section's residual bound: raised by codebook_prep_f16_kernel
    for (int u = tid; u < CPW * KG; u += 256) {
        const int cl = u / KG, kg = u - cl * KG, c = sub * CPW + cl;
        const int code = t * 32 + c;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (code < K) ? E[(size_t)code * D + 8 * kg + j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float av = fabsf(e[j]);
            bad |= !(av < __builtin_inff());
            amax = fmaxf(amax, av);
            sq[cl * D + 8 * kg + j] = sq_rn(e[j]);
        }
        const f32x4 o0 = {e[0], e[2], e[4], e[6]}, o1 = {e[1], e[3], e[5], e[7]};     // p -> k = 8 kg + 2 (p & 3) + (p >> 2)
        *(f32x4 *)(tile + kg * 256 + c * 8) = o0;
        *(f32x4 *)(tile + kg * 256 + c * 8 + 4) = o1;
    }
    for (int off = 32; off > 0; off >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, off));
        bad |= __shfl_xor(bad, off);
    }
    if ((tid & 63) == 0) { s_amax[tid >> 6] = amax; s_bad[tid >> 6] = bad; }
    __syncthreads();
    float a4[(CPW * 32 + 255) / 256];
#pragma unroll
    for (int i = 0; i < (CPW * 32 + 255) / 256; ++i) {
        const int pr = tid + 256 * i, cl = pr >> 5, m = pr & 31;
        float a = 0.0f;
        if (cl < CPW)
            for (int k0 = 0; k0 < D; k0 += 32) a = __fadd_rn(a, sq[cl * D + k0 + m]);
        a4[i] = a;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (CPW * 32 + 255) / 256; ++i)
        if (tid + 256 * i < CPW * 32) sq[tid + 256 * i] = a4[i];
    __syncthreads();
    if (tid < 64) {                                          // wave 0: lane cl < CPW finishes a code; then the workgroup's partial maxima
        float v = 0.0f;
        int wbad = 0;
        if (tid < CPW) {
            const float *a = sq + tid * 32;
            float sum = 0.0f;
            for (int l = 0; l < 8; ++l) {
                const float tl = __fadd_rn(__fadd_rn(__fadd_rn(a[l], a[l + 8]), a[l + 16]), a[l + 24]);
                sum = (l == 0) ? tl : __fadd_rn(sum, tl);
            }
            const int c = sub * CPW + tid, code = t * 32 + c;
            v = (code < K) ? sum : 0.0f;
            tile[32 * D + c] = v;
            en_all[t * 32 + c] = v;
            wbad = (code < K) && !(v < __builtin_inff());
        }
        float enmax = v;                                     // (a NaN norm: fmaxf drops it, the flag keeps it)
        for (int off = 32; off > 0; off >>= 1) {
            enmax = fmaxf(enmax, __shfl_xor(enmax, off));
            wbad |= __shfl_xor(wbad, off);
        }
        if (tid == 0) {
            const f32x4 part = {fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3])), enmax,
                                (wbad | s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) ? 1.0f : 0.0f, 0.0f};
            *(f32x4 *)(tile + 32 * D + 32 + 4 * sub) = part;
        }
        if (sub == 0 && tid >= 4 * SUBS && tid < 32) tile[32 * D + 32 + tid] = 0.0f;      // the rest of the padding
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers (called from dvq_abi.hip)
// ---------------------------------------------------------------------------------------------
int dvq_prep_codes_per_workgroup(int K) { return dvq_num_tiles(K) >= 256 ? 32 : 8; }

int dvq_launch_prep_f32(const float *E, int K, int D, void *prep, hipStream_t st)
{
    float *tiles = (float *)prep;
    float *en_all = (float *)((char *)prep + dvq_prep_en_offset(K, D));
    char *f16 = (char *)prep + dvq_prep_f16_offset(K, D);
    f16 = (char *)(((uintptr_t)f16 + 255) / 256 * 256);
    float *etamax = (float *)(f16 + 20);                     // DvqF16Meta::etamax (dvq_filter.h)
    const int T = dvq_num_tiles(K);
    if (dvq_prep_codes_per_workgroup(K) == 8)
        hipLaunchKernelGGL(codebook_prep_f32_kernel<8>, dim3(T * 4), dim3(256), 8 * D * sizeof(float), st, E, K, D, tiles, en_all, etamax);
    else
        hipLaunchKernelGGL(codebook_prep_f32_kernel<32>, dim3(T), dim3(256), 32 * D * sizeof(float), st, E, K, D, tiles, en_all, etamax);
    return (int)hipGetLastError();
}

template <int D, bool ROUTED>
static int launch_exact(const float *z, const float *tiles, const float *E, const float *mask,
                        int HW, int K, long N, float *zq, long long *codes, double *partials,
                        const int *list, const int *list_count, DvqLossTail tail, const DvqRouted &rv,
                        hipStream_t st, const DvqConv *fold_conv)
{
    static unsigned long long done_dense = 0, done_list = 0, done_fold = 0;
    const size_t shmem = 2 * (32 * D + 64) * sizeof(float);
    int rc = dvq_allow_dynamic_lds((const void *)vq_assign_exact_kernel<D, false, ROUTED>, (int)shmem, &done_dense);
    if (rc) return rc;
    rc = dvq_allow_dynamic_lds((const void *)vq_assign_exact_kernel<D, true, ROUTED>, (int)shmem, &done_list);
    if (rc) return rc;
    int blocks = (int)((N + 127) / 128);
    const DvqConv nocv = {};
    if (list != nullptr && fold_conv != nullptr) {
        rc = dvq_allow_dynamic_lds((const void *)vq_assign_exact_kernel<D, true, ROUTED, true>, (int)shmem, &done_fold);
        if (rc) return rc;
        if (blocks > DVQ_EXACT_LIST_BLOCKS) blocks = DVQ_EXACT_LIST_BLOCKS;
        hipLaunchKernelGGL((vq_assign_exact_kernel<D, true, ROUTED, true>), dim3(blocks), dim3(256), shmem, st,
                           z, tiles, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, rv, *fold_conv);
    } else if (list != nullptr) {
        if (blocks > DVQ_EXACT_LIST_BLOCKS) blocks = DVQ_EXACT_LIST_BLOCKS;
        hipLaunchKernelGGL((vq_assign_exact_kernel<D, true, ROUTED>), dim3(blocks), dim3(256), shmem, st,
                           z, tiles, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, rv, nocv);
    } else {
        hipLaunchKernelGGL((vq_assign_exact_kernel<D, false, ROUTED>), dim3(blocks), dim3(256), shmem, st,
                           z, tiles, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, rv, nocv);
    }
    return (int)hipGetLastError();
}

static const DvqRouted kNoRoute = {};

// pass 2 of the filter path: the tokens listed in list[0 .. *list_count); rv != nullptr: routed token ids
int dvq_launch_exact_list(const float *z, const float *prep, const float *E, const float *mask,
                          int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                          const int *list, const int *list_count, DvqLossTail tail, const DvqRouted *rv,
                          hipStream_t st, const DvqConv *fold_conv)
{
    if (rv != nullptr) {
        switch (D) {
        case 64:  return launch_exact<64, true>(z, prep, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, *rv, st, fold_conv);
        case 128: return launch_exact<128, true>(z, prep, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, *rv, st, fold_conv);
        case 256: return launch_exact<256, true>(z, prep, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, *rv, st, fold_conv);
        default:  return -1000;
        }
    }
    switch (D) {
    case 64:  return launch_exact<64, false>(z, prep, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, kNoRoute, st, fold_conv);
    case 128: return launch_exact<128, false>(z, prep, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, kNoRoute, st, fold_conv);
    case 256: return launch_exact<256, false>(z, prep, E, mask, HW, K, N, zq, codes, partials, list, list_count, tail, kNoRoute, st, fold_conv);
    default:  return -1000;
    }
}

// every token by the exact chain; rv != nullptr: the unique tokens of a routed batch (N = worst-case
// token count = B * HWout; ids beyond the last slot are invalid)
int dvq_launch_exact(const float *z, const float *prep, const float *E, const float *mask,
                     int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                     const DvqRouted *rv, hipStream_t st)
{
    const DvqLossTail none = {nullptr, nullptr, nullptr, 0, 0.0, 0.0f, nullptr, 0};
    return dvq_launch_exact_list(z, prep, E, mask, D, HW, K, N, zq, codes, partials, nullptr, nullptr, none, rv, st, nullptr);
}

int dvq_launch_loss_finalize(const double *partials, int nparts, double inv_numel, float beta,
                             float *loss, hipStream_t st)
{
    hipLaunchKernelGGL(vq_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nparts,
                       inv_numel, beta, loss);
    return (int)hipGetLastError();
}
