// qconv.hip -- the 1x1 quant_conv of the stage-1 models (reference models/stage1_dynamic/dqvae_dual_feat.py:34,66,
// dqvae_triple_feat.py:39,75, models/stage1/vqgan.py:42,70: nn.Conv2d(z_channels, embed_dim, 1)) with the router
// select fused in: h[b, :, y, x] = W src(b, y, x) + bias, where src is the vector of the encoder branch that won the
// position's coarse cell (EncoderDual.py:134-149 / EncoderTriple.py:148-176).  h_dual / h_triple is never written;
// indices, codebook_mask and the router's int64 gate come out as by-products, so the op replaces
// "route select -> quant_conv" and its output feeds the dense assign.
//
// A 1x1 conv is a [tokens, D] x [D, D] GEMM.  It runs on the fp16 matrix cores at fp32 grade: both operands
// are split x = hi + lo (hi = fp16(2^a x), lo = fp16(2^a x - hi); a per token, a power of two so the scaling is
// exact; the weight likewise with one scale for the tensor) and the product is hi*hi + hi*lo + lo*hi with fp32
// accumulation: the dropped lo*lo term is 2^-22 of |x||w|.  Three MFMAs at 16x the fp32-MFMA rate.  Summation
// order differs from MIOpen / oneDNN: equal to the reference's conv within 1e-5 relative to |x| |w|, not bit for
// bit -- the bit-exact contract of the assign starts at its input, so codes downstream
// are "exact given this h"; tests report the match rate against the conv-then-quantize order in fp64.
//
// Mapping (as the assign kernels): output channels are the MFMA rows (A operand, 32 per tile, streamed through a
// double-buffered LDS image), output positions the columns; a wave owns 32 consecutive positions of a row, so
// every store instruction writes 128-B runs of one output channel row.
#include "dvq_filter.h"

// prep buffer: [meta 256 B][tile t < D/32: hi image S16 KiB | lo image S16 KiB | bias of the tile's 32 rows + pad (256 B)]
//              [bias in channel order, D floats]   (QconvMeta, qconv_tile_bytes, qconv_row_channel: dvq_filter.h)
size_t dvq_qconv_prep_bytes_impl(int D) { return 256 + (size_t)(D / 32) * qconv_tile_bytes(D) + (size_t)D * sizeof(float); }

__global__ __launch_bounds__(1024) void qconv_meta_kernel(const float *__restrict__ Wt, int D, QconvMeta *__restrict__ meta)
{
    __shared__ float s_max[1024];
    __shared__ int s_bad[1024];
    float amax = 0.0f;
    int bad = 0;
    for (int i = threadIdx.x; i < D * D; i += 1024) {
        const float v = fabsf(Wt[i]);
        bad |= !(v < __builtin_inff());
        amax = fmaxf(amax, v);
    }
    s_max[threadIdx.x] = amax;
    s_bad[threadIdx.x] = bad;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + w]);
            s_bad[threadIdx.x] |= s_bad[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int b = 0;
        if (s_max[0] > 0.0f) {
            int e;
            (void)frexpf(s_max[0], &e);
            b = 14 - e;
        }
        if (b > 100 || b < -100) s_bad[0] = 1;
        meta->ok = s_bad[0] ? 0 : 1;
        meta->b_exp = b;
        meta->scale_w = ldexpf(1.0f, s_bad[0] ? 0 : b);
        meta->inv_scale_w = ldexpf(1.0f, s_bad[0] ? 0 : -b);
    }
}

// image of tile t, k-step s, lane l, j < 8: W[32t + qconv_row_channel(l & 31)][16s + 8(l >> 5) + j]  (the A fragment of the
// MFMA).  The MFMA rows of a tile are a PERMUTATION of its 32 output channels, chosen so that accumulator register r of lane
// half h holds channel 32t + 16(r >> 3) + 8h + (r & 7): the (k-step, 8 channels per lane half) layout in which the assign's
// pass 1 keeps its latents -- with the conv fused into pass 1 (CONV form) the accumulators ARE those registers.
__global__ __launch_bounds__(256) void qconv_prep_kernel(const float *__restrict__ Wt, const float *__restrict__ bias,
                                                         int D, const QconvMeta *__restrict__ meta, char *__restrict__ img)
{
    const float sw = meta->scale_w;
    const int S16 = D / 16;
    const size_t per = (size_t)S16 * 512;                    // halves per image
    const size_t tile_b = qconv_tile_bytes(D);
    const size_t total = (size_t)(D / 32) * (per + 32);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / (per + 32));
        const int r = (int)(i - (size_t)t * (per + 32));
        char *tile = img + (size_t)t * tile_b;
        if (r < (int)per) {
            const int s = r >> 9, lane = (r >> 3) & 63, j = r & 7;
            const int o = t * 32 + qconv_row_channel(lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
            const float v = Wt[(size_t)o * D + k] * sw;
            const _Float16 hi = (_Float16)v;
            ((_Float16 *)tile)[r] = hi;
            ((_Float16 *)(tile + per * 2))[r] = (_Float16)(v - (float)hi);
        } else {
            const int q = r - (int)per;
            ((float *)(tile + per * 4))[q] = (bias != nullptr) ? bias[t * 32 + qconv_row_channel(q)] : 0.0f;
        }
    }
    float *bias_c = (float *)(img + (size_t)(D / 32) * tile_b);            // channel order, for the CONV form of pass 1
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) bias_c[i] = (bias != nullptr) ? bias[i] : 0.0f;
}

template <int D, bool SEL>
__global__ __launch_bounds__(256, 2) void qconv_kernel(
    const float *__restrict__ x, const DvqRouted rv, const char *__restrict__ img, const QconvMeta *__restrict__ meta,
    int HW, long N, float *__restrict__ hout)
{
    constexpr int S16 = D / 16;
    constexpr int IMG = S16 * 1024;                          // one image (hi or lo)
    constexpr int TILE = 2 * IMG + 256;
    constexpr int T = D / 32;
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 x TILE

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    const long n_raw = ((long)tile_id * 4 + wave) * 32 + c;
    const int n = (n_raw < N) ? (int)n_raw : -1;
    const int nn = (n >= 0) ? n : (int)(N - 1);
    const int b = nn / HW, pos = nn - b * HW;

    auto stage = [&](int t, char *buf) {                     // TILE bytes = 2*S16 + ... 1-KiB pieces over 4 waves
        const char *src = img + (size_t)t * TILE;
        constexpr int PIECES = 2 * S16;
        for (int i = wave; i < PIECES; i += 4) glds16(src + i * 1024 + lane * 16, buf + i * 1024);
        if (wave == 0) glds4(src + 2 * IMG + lane * 4, buf + 2 * IMG);
    };
    const float *zp;
    size_t st;
    if (SEL) {
        const int y = pos / rv.Wout, xx = pos - y * rv.Wout;
        const int SC = rv.sub[rv.G - 1];
        const size_t cell = (size_t)b * rv.hc * rv.wc + (y / SC) * rv.wc + xx / SC;
        // (the gate is read before the first LDS-DMA is in flight: an ordinary load whose value is used while one is
        // makes hipcc drain the whole vector-memory queue)
        const DvqGateRaw graw = dvq_gate_fetch(rv.gate, rv.gate_mode, rv.G, cell);
        const int g = dvq_gate_reduce(graw, rv.gate_mode, rv.G, rv.thr);
        const int rep_g = rv.rep[g];
        if (n >= 0 && h == 0 && rv.cmask_out != nullptr) {
            rv.cmask_out[n] = 1.0f / (float)(rep_g * rep_g);
            if (y % SC == 0 && xx % SC == 0) {
                rv.indices_out[cell] = g;
                if (rv.gate_mode == 2 && rv.gate_out != nullptr) {
                    const float e = graw.f[0];
                    longlong2 gg; gg.x = (e <= rv.thr) ? 1 : 0; gg.y = (e > rv.thr) ? 1 : 0;
                    *(longlong2 *)(rv.gate_out + 2 * cell) = gg;
                }
            }
        }
        int stride_l;
        zp = dvq_dense_source(rv, b, y, xx, g, stride_l) + (size_t)8 * h * stride_l;
        st = (size_t)stride_l;
    } else {
        zp = x + ((size_t)b * D + 8 * h) * HW + pos;
        st = (size_t)HW;
    }
    stage(0, lds);
    float xf[S16][8];
#pragma unroll
    for (int s = 0; s < S16; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[s][j] = zp[(size_t)(16 * s + j) * st];
    // per-token power-of-two scale: 2^a max|x| in [2^13, 2^14)
    float amax = 0.0f;
#pragma unroll
    for (int s = 0; s < S16; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = vmax_abs(amax, xf[s][j]);
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    int ea = 0;
    if (amax > 0.0f && amax < __builtin_inff()) { int e; (void)frexpf(amax, &e); ea = 14 - e; }
    ea = ea > 100 ? 100 : (ea < -100 ? -100 : ea);
    const float sa = ldexpf(1.0f, ea);
    const float unscale = ldexpf(meta->inv_scale_w, -ea);
    f16x8 xh[S16], xl[S16];
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
    float *hp = hout + ((size_t)b * D + 8 * h) * HW + pos;   // register r = 4 g4 + q of lane half h: channel 32t + 16(r >> 3) + 8h + (r & 7)

    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // tile t landed; everyone is done with tile t-1
        const char *buf = lds + (t & 1) * TILE;
        if (t + 1 < T) stage(t + 1, lds + ((t + 1) & 1) * TILE);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < S16; ++s) {
            const f16x8 ah = *(const f16x8 *)(buf + s * 1024 + lane * 16);
            const f16x8 al = *(const f16x8 *)(buf + IMG + s * 1024 + lane * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[s], acc, 0, 0, 0);     // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[s], acc, 0, 0, 0);
        }
        if (n >= 0) {
            const float *bias = (const float *)(buf + 2 * IMG) + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 b4 = *(const f32x4 *)(bias + 8 * g4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    hp[(size_t)(32 * t + 16 * (g4 >> 1) + 4 * (g4 & 1) + q) * HW] = __builtin_fmaf(acc[4 * g4 + q], unscale, b4[q]);
            }
        }
    }
}

int dvq_launch_qconv_prep(const float *Wt, const float *bias, int D, void *prep, hipStream_t st)
{
    QconvMeta *meta = (QconvMeta *)prep;
    char *img = (char *)prep + 256;
    hipLaunchKernelGGL(qconv_meta_kernel, dim3(1), dim3(1024), 0, st, Wt, D, meta);
    hipLaunchKernelGGL(qconv_prep_kernel, dim3(64), dim3(256), 0, st, Wt, bias, D, meta, img);
    return (int)hipGetLastError();
}

template <int D, bool SEL>
static int launch_qconv(const float *x, const DvqRouted &rv, const void *prep, int HW, long N, float *hout, hipStream_t st)
{
    static unsigned long long done = 0;
    const size_t shmem = 2 * (2 * (size_t)(D / 16) * 1024 + 256);
    int rc = dvq_allow_dynamic_lds((const void *)qconv_kernel<D, SEL>, (int)shmem, &done);
    if (rc) return rc;
    hipLaunchKernelGGL((qconv_kernel<D, SEL>), dim3((unsigned)((N + 127) / 128)), dim3(256), shmem, st, x, rv,
                       (const char *)prep + 256, (const QconvMeta *)prep, HW, N, hout);
    return (int)hipGetLastError();
}

// x != nullptr: dense input [B, D, HW]; else the select fused in through rv (dense form, gate given)
int dvq_launch_qconv(const float *x, const DvqRouted *rv, const void *prep, int D, int HW, long N, float *hout,
                     hipStream_t st)
{
    const DvqRouted none = {};
    if (x != nullptr) {
        switch (D) {
        case 64:  return launch_qconv<64, false>(x, none, prep, HW, N, hout, st);
        case 128: return launch_qconv<128, false>(x, none, prep, HW, N, hout, st);
        case 256: return launch_qconv<256, false>(x, none, prep, HW, N, hout, st);
        default:  return -1000;
        }
    }
    switch (D) {
    case 64:  return launch_qconv<64, true>(nullptr, *rv, prep, HW, N, hout, st);
    case 128: return launch_qconv<128, true>(nullptr, *rv, prep, HW, N, hout, st);
    case 256: return launch_qconv<256, true>(nullptr, *rv, prep, HW, N, hout, st);
    default:  return -1000;
    }
}
