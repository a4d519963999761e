// vq_fold.hip -- the 1x1 quant_conv folded into the codebook: preparation of the folded code images.
//
// The stage-1 models put a pointwise conv between the routing tail and the quantizer, h = W x + bias
// (models/stage1_dynamic/dqvae_dual_feat.py:34,66, dqvae_triple_feat.py:39,75, models/stage1/vqgan.py:42,70).  The nearest code
// of h maximises  h.e_j - en_j/2 = x.(W^T e_j) + (bias.e_j - en_j/2),  so pass 1 of the filter path can score the conv's INPUT
// against E' = E W with no conv at all: stage 2's tokenisation keeps the codes only
// (models/stage2_dynamic/dqtransformer_uncond_entropy.py:166-171,182).  This file builds what that needs, once per
// (codebook, conv weight): E' and the seed constants in float64, the two fp16 tile images of E' in the layouts of
// vq_assign_filter.hip (32x32x16 order for the resolver, 16x16x32 order for pass 1), and the constants of the bound W'
// (dvq_filter.h: DvqFoldMeta / dvq_fold_threshold).  Every constant is an UPPER bound, rounded up.
//
// Buffer: [DvqFoldMeta, 256 B][image "32"][image "16"]  (offsets as in the codebook prep's f16 section)
//         [scratch: E' as K x D doubles | per-code statistics 4 doubles | per-column statistics of W^T W, 2 doubles]
#include "dvq_filter.h"

static size_t fold_img_bytes(int K, int D)
{
    const size_t tile = (size_t)(D / 16) * 1024 + 256;
    return ((size_t)dvq_num_tiles(K) * tile + 255) / 256 * 256;
}
static size_t fold_scratch_offset(int K, int D) { return 256 + 2 * fold_img_bytes(K, D); }
size_t dvq_fold_prep_bytes_impl(int K, int D)
{
    return fold_scratch_offset(K, D) + ((size_t)K * D + 4 * (size_t)K + 2 * (size_t)D + 8) * sizeof(double) + 256;
}

// E'[j][k] = sum_o E[j][o] W[o][k] and A[j][k] = sum_o |E[j][o]| |W[o][k]| in float64; per code: max |e'|, ||e'||^2, ||A_j||^2,
// bias.e_j.  One workgroup per code, one thread per input channel k (W rows are read coalesced, E[j][o] is a broadcast).
__global__ void fold_gemm_kernel(const float *__restrict__ E, const float *__restrict__ Wt, const float *__restrict__ bias,
                                 int K, int D, double *__restrict__ Ep, double *__restrict__ stat)
{
    __shared__ double red[4][256];
    __shared__ float erow[256];
    const int j = blockIdx.x, k = threadIdx.x;
    if (k < D) erow[k] = E[(size_t)j * D + k];
    __syncthreads();
    double acc = 0.0, aab = 0.0;
    if (k < D)
        for (int o = 0; o < D; ++o) {
            const double e = (double)erow[o], w = (double)Wt[(size_t)o * D + k];
            acc += e * w;
            aab += fabs(e) * fabs(w);
        }
    if (k < D) Ep[(size_t)j * D + k] = acc;
    red[0][k] = (k < D) ? fabs(acc) : 0.0;
    red[1][k] = (k < D) ? acc * acc : 0.0;
    red[2][k] = (k < D) ? aab * aab : 0.0;
    red[3][k] = (k < D && bias != nullptr) ? (double)bias[k] * (double)erow[k] : 0.0;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (k < w) {
            red[0][k] = fmax(red[0][k], red[0][k + w]);
            red[1][k] += red[1][k + w];
            red[2][k] += red[2][k + w];
            red[3][k] += red[3][k + w];
        }
        __syncthreads();
    }
    if (k == 0) {
        stat[4 * (size_t)j + 0] = red[0][0];
        stat[4 * (size_t)j + 1] = red[1][0];
        stat[4 * (size_t)j + 2] = red[2][0];
        stat[4 * (size_t)j + 3] = red[3][0];
    }
}

// column i of G = W^T W: sum_k |G_ik| (Gershgorin: lambda_max(G) <= the largest of them) and G_ii (their sum is ||W||_F^2)
__global__ void fold_gram_kernel(const float *__restrict__ Wt, int D, double *__restrict__ cstat)
{
    __shared__ double red[256];
    const int i = blockIdx.x, k = threadIdx.x;
    double g = 0.0;
    if (k < D)
        for (int o = 0; o < D; ++o) g += (double)Wt[(size_t)o * D + i] * (double)Wt[(size_t)o * D + k];
    red[k] = (k < D) ? fabs(g) : 0.0;
    if (k == i) cstat[2 * i + 1] = g;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (k < w) red[k] += red[k + w];
        __syncthreads();
    }
    if (k == 0) cstat[2 * i] = red[0];
}

__device__ __forceinline__ float up_f32(double v) { return (float)(v * (1.0 + 1.0e-6)); }    // an upper bound survives the cast

__global__ __launch_bounds__(1024) void fold_meta_kernel(const double *__restrict__ stat, const double *__restrict__ cstat,
                                                         const float *__restrict__ bias, const float *__restrict__ en_all,
                                                         const DvqF16Meta *__restrict__ cbmeta, int K, int D,
                                                         DvqFoldMeta *__restrict__ meta)
{
    __shared__ double r0[1024], r1[1024], r2[1024], r3[1024];
    const int t = threadIdx.x;
    double amax = 0.0, n2 = 0.0, q2 = 0.0, cabs = 0.0;
    int bad = 0;
    for (int j = t; j < K; j += 1024) {
        const double a = stat[4 * (size_t)j], b = stat[4 * (size_t)j + 1], c = stat[4 * (size_t)j + 2];
        const double cj = stat[4 * (size_t)j + 3] - 0.5 * (double)en_all[j];
        bad |= !(a < 1.0e300) || !(b < 1.0e300) || !(c < 1.0e300) || !(fabs(cj) < 1.0e300);
        amax = fmax(amax, a); n2 = fmax(n2, b); q2 = fmax(q2, c); cabs = fmax(cabs, fabs(cj));
    }
    r0[t] = amax; r1[t] = n2; r2[t] = q2; r3[t] = cabs;
    __shared__ int sbad[1024];
    sbad[t] = bad;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (t < w) {
            r0[t] = fmax(r0[t], r0[t + w]); r1[t] = fmax(r1[t], r1[t + w]);
            r2[t] = fmax(r2[t], r2[t + w]); r3[t] = fmax(r3[t], r3[t + w]);
            sbad[t] |= sbad[t + w];
        }
        __syncthreads();
    }
    if (t == 0) {
        double gmax = 0.0, fro2 = 0.0, bn2 = 0.0;
        for (int i = 0; i < D; ++i) {
            gmax = fmax(gmax, cstat[2 * i]);
            fro2 += cstat[2 * i + 1];
            if (bias != nullptr) bn2 += (double)bias[i] * (double)bias[i];
        }
        bad = sbad[0] | !(gmax < 1.0e300) | !(fro2 < 1.0e300) | !(bn2 < 1.0e300) | !cbmeta->ok;
        int b = 0;
        if (r0[0] > 0.0) {
            int e;
            (void)frexp(r0[0], &e);                   // amax = m 2^e, m in [0.5, 1)
            b = 15 - e;                               // 2^b amax in [2^14, 2^15)
        }
        if (b > 100 || b < -100) bad = 1;
        const double sb = ldexp(1.0, bad ? 0 : b);
        const double seedmax = sb * r3[0];
        if (!(seedmax < 1.0e37) || !(sqrt(r1[0]) < 1.0e30) || !(sqrt(fmin(gmax, fro2)) < 1.0e15)) bad = 1;
        meta->ok = bad ? 0 : 1;
        meta->b_exp = b;
        meta->scale_b = (float)sb;
        meta->emax = up_f32(sqrt(r1[0]));
        meta->enmax = cbmeta->enmax;
        meta->etamax = 0.0f;                          // filled by fold_image_kernel
        meta->seedmax = bad ? 0.0f : up_f32(seedmax);
        meta->qmax = up_f32(sqrt(r2[0]));
        meta->sigma = up_f32(sqrt(fmin(gmax, fro2)));
        meta->bnorm = up_f32(sqrt(bn2));
        meta->emax0 = cbmeta->emax;
    }
}

// both tile images (layouts: codebook_prep_f16_kernel / codebook_prep_f16x_kernel of vq_assign_filter.hip) and the
// residual norm eta' = max_j || 2^b' e'_j - fp16(2^b' e'_j) ||.  One workgroup per code tile.
__global__ __launch_bounds__(256) void fold_image_kernel(const double *__restrict__ Ep, const double *__restrict__ stat,
                                                         const float *__restrict__ en_all, int K, int D,
                                                         DvqFoldMeta *__restrict__ meta, char *__restrict__ img32,
                                                         char *__restrict__ img16)
{
    __shared__ double eta2[32][8];
    const double sb = (double)meta->scale_b;
    const int S16 = D / 16, S32 = D / 32;
    const int img_halves = S16 * 512;
    const size_t tile_bytes = (size_t)img_halves * 2 + 256;
    const int t = blockIdx.x;
    char *t32 = img32 + (size_t)t * tile_bytes, *t16 = img16 + (size_t)t * tile_bytes;
    for (int r = threadIdx.x; r < img_halves; r += 256) {
        {   // 32x32x16 order: [s < D/16][lane][j]: code 32t + (lane & 31), k = 16 s + 8 (lane >> 5) + j
            const int s = r >> 9, lane = (r >> 3) & 63, j = r & 7;
            const int cl = lane & 31, code = t * 32 + cl, k = 16 * s + 8 * (lane >> 5) + j;
            const double v = (code < K) ? Ep[(size_t)code * D + k] * sb : 0.0;
            ((_Float16 *)t32)[r] = (_Float16)(float)v;
            (void)cl;
        }
        {   // 16x16x32 order: fragment F = c2 (D/32) + s', lane, j: code 32t + 16 c2 + (lane & 15), k = 32 s' + 8 (lane >> 4) + j
            const int F = r >> 9, lane = (r >> 3) & 63, j = r & 7;
            const int c2 = F / S32, sp = F - c2 * S32;
            const int code = t * 32 + 16 * c2 + (lane & 15), k = 32 * sp + 8 * (lane >> 4) + j;
            const double v = (code < K) ? Ep[(size_t)code * D + k] * sb : 0.0;
            ((_Float16 *)t16)[r] = (_Float16)(float)v;
        }
    }
    if (threadIdx.x < 64) {
        const int q = threadIdx.x, code = t * 32 + q;
        float v = 0.0f;
        if (q < 32) {
            v = DVQ_SEED_PAD;
            if (code < K) {
                const double cj = stat[4 * (size_t)code + 3] - 0.5 * (double)en_all[code];
                v = fmaxf((float)(sb * cj), DVQ_SEED_PAD);
            }
        }
        ((float *)(t32 + (size_t)img_halves * 2))[q] = v;
        ((float *)(t16 + (size_t)img_halves * 2))[q] = v;
    }
    {   // residual norms in a fixed order (the bound, and with it which tokens get resolved, is the same in every run)
        const int cl = threadIdx.x >> 3, part = threadIdx.x & 7, code = t * 32 + cl;
        double a = 0.0;
        if (code < K)
            for (int k = part; k < D; k += 8) {
                const double v = Ep[(size_t)code * D + k] * sb;
                const double res = v - (double)(float)(_Float16)(float)v;
                a += res * res;
            }
        eta2[cl][part] = a;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double a = 0.0;
        for (int i = 0; i < 8; ++i) a += eta2[threadIdx.x][i];
        if (a > 0.0) {
            const float v = (float)(sqrt(a) * 1.001);
            atomicMax((int *)&meta->etamax, __float_as_int(v));      // positive floats order as ints
        }
    }
}

// codebook [K, D], its prep (exact norms en_j and meta), conv weight [D, D] (row = output channel), bias nullable [D]
int dvq_launch_fold_prep(const float *E, int K, int D, const void *cbprep, const float *Wt, const float *bias, void *fprep,
                         hipStream_t st)
{
    DvqFoldMeta *meta = (DvqFoldMeta *)fprep;
    char *img32 = (char *)fprep + 256, *img16 = img32 + fold_img_bytes(K, D);
    double *Ep = (double *)((char *)fprep + fold_scratch_offset(K, D));
    double *stat = Ep + (size_t)K * D, *cstat = stat + 4 * (size_t)K;
    const float *en_all = (const float *)((const char *)cbprep + dvq_prep_en_offset(K, D));
    const char *cb16 = (const char *)cbprep + dvq_prep_f16_offset(K, D);
    cb16 = (const char *)(((uintptr_t)cb16 + 255) / 256 * 256);
    hipLaunchKernelGGL(fold_gemm_kernel, dim3(K), dim3(256), 0, st, E, Wt, bias, K, D, Ep, stat);
    hipLaunchKernelGGL(fold_gram_kernel, dim3(D), dim3(256), 0, st, Wt, D, cstat);
    hipLaunchKernelGGL(fold_meta_kernel, dim3(1), dim3(1024), 0, st, stat, cstat, bias, en_all, (const DvqF16Meta *)cb16, K, D, meta);
    hipLaunchKernelGGL(fold_image_kernel, dim3(dvq_num_tiles(K)), dim3(256), 0, st, Ep, stat, en_all, K, D, meta, img32, img16);
    return (int)hipGetLastError();
}
