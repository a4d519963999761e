/*
 * dvq_oracle.c -- CPU restatement of the DQ-VAE vector-quantization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.  The
 * product path (dynamicvectorquantization_amd/, libdvq.so) never links,
 * imports or executes anything from oracle/.
 *
 * Parity status: PINNED.  The reference is pure Python/PyTorch and its
 * arithmetic lives in torch-CPU (MKL sgemm + ATen reductions); it has no
 * tests of its own for this path.  This restatement is pinned by golden
 * vectors captured from the imported reference in the build container
 * (oracle/gen_golden.py -> tests/golden/), torch 2.10.0 CPU, see
 * tests/test_oracle_golden.py.
 *
 * Arithmetic orders restated here (all fp32, every rounding explicit):
 *   dot(z,e)   strictly sequential k = 0..D-1 fused-multiply-add chain from 0
 *              (what torch.addmm / einsum produce on CPU for D <= 512):
 *              modules/vector_quantization/quantize2_mask.py:41-46,
 *              quantize_vqgan.py:280-282
 *   sumsq(v)   ATen vectorised inner reduction: 32 partial sums a[i%32] of the
 *              rounded squares, combined ((a[l]+a[l+8])+a[l+16])+a[l+24] per
 *              l < 8, then l = 0..7 summed left to right:
 *              quantize2_mask.py:39-40, quantize_vqgan.py:280-281
 *   distance   d = fl(fl(xn + en) - 2*dot)            quantize2_mask.py:41
 *   argmin     first index wins ties; NaN is the minimum, first NaN wins:
 *              quantize2_mask.py:53, quantize_vqgan.py:284
 *   z_q        fl(z + fl(e - z))                      quantize2_mask.py:182
 *   loss terms (e - z)^2 * mask, summed (here in double; the reference's f32
 *              mean is compared at 1e-5 relative)      quantize2_mask.py:172-179
 *   router select / codebook mask                     EncoderDual.py:134-149,
 *                                                     EncoderTriple.py:148-176
 *   entropy gate                                      RouterDual.py:53-57
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -fopenmp; FMA only where
 * written with fmaf()).
 */
#include <math.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define DVQ_ORACLE_VERSION 1

int dvq_oracle_version(void) { return DVQ_ORACLE_VERSION; }

/* ATen-order sum of squares of v[0], v[stride], ... (D elements).
 * Pinned for D % 32 == 0 and D <= 512 (beyond that ATen cascades). */
float dvq_oracle_sumsq(const float *v, int D, long stride)
{
    float a[32];
    for (int m = 0; m < 32; ++m) a[m] = 0.0f;
    for (int i = 0; i < D; ++i) {
        float x = v[(long)i * stride];
        float sq = x * x;            /* rounded square, no contraction */
        a[i & 31] = a[i & 31] + sq;
    }
    float t[8];
    for (int l = 0; l < 8; ++l)
        t[l] = ((a[l] + a[l + 8]) + a[l + 16]) + a[l + 24];
    float s = t[0];
    for (int l = 1; l < 8; ++l) s = s + t[l];
    return s;
}

void dvq_oracle_codebook_norms(const float *E, int K, int D, float *en)
{
    for (int j = 0; j < K; ++j) en[j] = dvq_oracle_sumsq(E + (long)j * D, D, 1);
}

/* sequential-k FMA chain, z strided (NCHW token), e contiguous */
static inline float chain_dot(const float *z, long zs, const float *e, int D)
{
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) acc = fmaf(z[(long)k * zs], e[k], acc);
    return acc;
}

/* torch CPU argmin semantics over candidates visited in ascending index:
 * take if d < best, or d is NaN and best is not. */
static inline int take_min(float d, float best)
{
    return (d < best) || (d != d && best == best);
}

/*
 * Full dense distances for one token (debug / golden support).
 * z: token base pointer, channel stride zs.
 */
void dvq_oracle_token_distances(const float *z, long zs, const float *E,
                                const float *en, int D, int K, float *d_out)
{
    float xn = dvq_oracle_sumsq(z, D, zs);
    for (int j = 0; j < K; ++j) {
        float dot = chain_dot(z, zs, E + (long)j * D, D);
        float bias = xn + en[j];
        d_out[j] = bias - 2.0f * dot;   /* 2*dot exact; one rounding */
    }
}

/*
 * VectorQuantize2.forward / VectorQuantizer2.forward, eval mode.
 *   z      [B, D, HW] f32 (NCHW with HW = H*W); HW == 1 gives the flat [N, D] case
 *   E      [K, D] f32 (codebook rows; the caller passes weight[:-1] for VQEmbedding)
 *   mask   nullable, [B, HW] f32 (codebook_mask [B,1,H,W])
 *   zq     nullable, [B, D, HW] f32
 *   codes  [B, HW] int64
 *   sqerr  nullable: sum over all elements of (e - z)^2 * mask, double
 *   dmin   nullable: [B, HW] winning distance (f32, reference arithmetic)
 * Returns 0.
 */
int dvq_oracle_vq_assign_nchw(const float *z, const float *E, const float *mask,
                              int B, int D, int HW, int K,
                              float *zq, int64_t *codes, double *sqerr,
                              float *dmin)
{
    /* Register tile: TB tokens x JB codes advance together through k.  Every (token, code) output
     * is still its own strictly sequential k = 0..D-1 fmaf chain (vfmadd231ps is the same fused
     * operation per lane), so the tiling changes speed only, never a bit of the result. */
    enum { JB = 32, TB = 4, DMAX = 1024 };
    if (D > DMAX) return -3;
    float *en = (float *)malloc(sizeof(float) * (size_t)K);
    /* transposed copy ET[k][j] so JB independent chains advance together */
    float *ET = (float *)malloc(sizeof(float) * (size_t)K * (size_t)D);
    if (!en || !ET) { free(en); free(ET); return -1; }
    dvq_oracle_codebook_norms(E, K, D, en);
    for (int j = 0; j < K; ++j)
        for (int k = 0; k < D; ++k) ET[(long)k * K + j] = E[(long)j * D + k];
    long N = (long)B * HW;
    double total = 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total)
    for (long n0 = 0; n0 < N; n0 += TB) {
        int tn = (N - n0 < TB) ? (int)(N - n0) : TB;
        float zc[TB][DMAX];
        float xn[TB], best[TB];
        long bi[TB];
        for (int t = 0; t < TB; ++t) {
            long n = n0 + (t < tn ? t : tn - 1);          /* ragged tail: repeat the last token */
            long b = n / HW, hw = n % HW;
            const float *zt = z + b * (long)D * HW + hw;
            xn[t] = dvq_oracle_sumsq(zt, D, HW);
            for (int k = 0; k < D; ++k) zc[t][k] = zt[(long)k * HW];
            best[t] = 0.0f;
            bi[t] = -1;
        }
        for (int j0 = 0; j0 < K; j0 += JB) {
            int jn = K - j0 < JB ? K - j0 : JB;
            float acc[TB][JB];
            if (jn == JB) {
#if defined(__AVX2__) && defined(__FMA__)
                __m256 a[TB][JB / 8];
                for (int t = 0; t < TB; ++t)
                    for (int v = 0; v < JB / 8; ++v) a[t][v] = _mm256_setzero_ps();
                for (int k = 0; k < D; ++k) {
                    const float *et = ET + (long)k * K + j0;
                    __m256 e0 = _mm256_loadu_ps(et), e1 = _mm256_loadu_ps(et + 8);
                    __m256 e2 = _mm256_loadu_ps(et + 16), e3 = _mm256_loadu_ps(et + 24);
                    for (int t = 0; t < TB; ++t) {
                        __m256 zk = _mm256_set1_ps(zc[t][k]);
                        a[t][0] = _mm256_fmadd_ps(zk, e0, a[t][0]);
                        a[t][1] = _mm256_fmadd_ps(zk, e1, a[t][1]);
                        a[t][2] = _mm256_fmadd_ps(zk, e2, a[t][2]);
                        a[t][3] = _mm256_fmadd_ps(zk, e3, a[t][3]);
                    }
                }
                for (int t = 0; t < TB; ++t)
                    for (int v = 0; v < JB / 8; ++v) _mm256_storeu_ps(&acc[t][8 * v], a[t][v]);
#else
                for (int t = 0; t < TB; ++t)
                    for (int j = 0; j < JB; ++j) acc[t][j] = 0.0f;
                for (int k = 0; k < D; ++k) {
                    const float *__restrict et = ET + (long)k * K + j0;
                    for (int t = 0; t < TB; ++t) {
                        float zk = zc[t][k];
                        for (int j = 0; j < JB; ++j) acc[t][j] = fmaf(zk, et[j], acc[t][j]);
                    }
                }
#endif
            } else {
                for (int t = 0; t < TB; ++t)
                    for (int j = 0; j < JB; ++j) acc[t][j] = 0.0f;
                for (int k = 0; k < D; ++k) {
                    const float *et = ET + (long)k * K + j0;
                    for (int t = 0; t < TB; ++t) {
                        float zk = zc[t][k];
                        for (int j = 0; j < jn; ++j) acc[t][j] = fmaf(zk, et[j], acc[t][j]);
                    }
                }
            }
            for (int t = 0; t < tn; ++t)
                for (int j = 0; j < jn; ++j) {
                    float bias = xn[t] + en[j0 + j];
                    float d = bias - 2.0f * acc[t][j];   /* 2*dot exact; one rounding */
                    if (bi[t] < 0 || take_min(d, best[t])) { best[t] = d; bi[t] = j0 + j; }
                }
        }
        for (int t = 0; t < tn; ++t) {
            long n = n0 + t;
            long b = n / HW, hw = n % HW;
            codes[n] = bi[t];
            if (dmin) dmin[n] = best[t];
            const float *e = E + bi[t] * (long)D;
            float m = mask ? mask[n] : 1.0f;
            double accd = 0.0;
            for (int k = 0; k < D; ++k) {
                float diff = e[k] - zc[t][k];
                if (zq) zq[b * (long)D * HW + (long)k * HW + hw] = zc[t][k] + diff;
                float sq = diff * diff;
                float w = sq * m;
                accd += (double)w;
            }
            total += accd;
        }
    }
    if (sqerr) *sqerr = total;
    free(en);
    free(ET);
    return 0;
}

/* nn.Embedding gather: out[n, :] = E[idx[n], :]   (quantize2_mask.py:130-132, 207-210) */
int dvq_oracle_embed_gather(const float *E, int K, int D, const int64_t *idx,
                            long N, float *out)
{
    for (long n = 0; n < N; ++n) {
        int64_t j = idx[n];
        if (j < 0 || j >= K) return -2;
        memcpy(out + n * D, E + j * (long)D, sizeof(float) * (size_t)D);
    }
    return 0;
}

/*
 * DualGrainFixedEntropyRouter.forward (RouterDual.py:53-57):
 *   gate[..., 0] = entropy <= thr, gate[..., 1] = entropy > thr, int64.
 * thr is the python float from the JSON; torch compares an f32 tensor with a
 * python scalar in f32 (the scalar is cast to the tensor dtype).
 */
void dvq_oracle_entropy_gate(const float *entropy, long n, double thr, int64_t *gate)
{
    float t = (float)thr;
    for (long i = 0; i < n; ++i) {
        gate[2 * i + 0] = (entropy[i] <= t) ? 1 : 0;
        gate[2 * i + 1] = (entropy[i] > t) ? 1 : 0;
    }
}

/* first-max argmax over G gate values (torch.argmax: first index on ties, NaN is max) */
static inline int argmax_f32(const float *g, int G)
{
    int bi = 0;
    float best = g[0];
    for (int i = 1; i < G; ++i) {
        float v = g[i];
        if ((v > best) || (v != v && best == best)) { best = v; bi = i; }
    }
    return bi;
}
static inline int argmax_i64(const int64_t *g, int G)
{
    int bi = 0;
    int64_t best = g[0];
    for (int i = 1; i < G; ++i)
        if (g[i] > best) { best = g[i]; bi = i; }
    return bi;
}

/*
 * Routing tail of DualGrainEncoder.forward, eval mode (EncoderDual.py:134-149).
 *   gate      [B, hc, wc, 2]  (f32 logits if gate_is_i64 == 0, else int64)
 *   h_coarse  [B, C, hc, wc], h_fine [B, C, 2hc, 2wc]
 *   h_out     [B, C, 2hc, 2wc]; indices [B, hc, wc] int64; cmask [B, 1, 2hc, 2wc] f32
 */
void dvq_oracle_route_select_dual(const void *gate, int gate_is_i64,
                                  const float *h_coarse, const float *h_fine,
                                  int B, int C, int hc, int wc,
                                  float *h_out, int64_t *indices, float *cmask)
{
    int H = 2 * hc, W = 2 * wc;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < hc; ++y)
            for (int x = 0; x < wc; ++x) {
                long cell = ((long)b * hc + y) * wc + x;
                int g = gate_is_i64 ? argmax_i64((const int64_t *)gate + cell * 2, 2)
                                    : argmax_f32((const float *)gate + cell * 2, 2);
                indices[cell] = g;
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    long cell = ((long)b * hc + y / 2) * wc + x / 2;
                    long o = (((long)b * C + c) * H + y) * W + x;
                    long ic = (((long)b * C + c) * hc + y / 2) * wc + x / 2;
                    h_out[o] = indices[cell] == 0 ? h_coarse[ic] : h_fine[o];
                }
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long cell = ((long)b * hc + y / 2) * wc + x / 2;
                cmask[((long)b * H + y) * W + x] = indices[cell] == 0 ? 0.25f : 1.0f;
            }
}

/*
 * Routing tail of TripleGrainEncoder.forward, eval mode (EncoderTriple.py:148-176).
 *   gate [B, hc, wc, 3]; h_coarse [B,C,hc,wc]; h_median [B,C,2hc,2wc]; h_fine [B,C,4hc,4wc]
 *   0 -> coarse (mask 0.0625), 1 -> median (0.25), 2 -> fine (1.0)
 */
void dvq_oracle_route_select_triple(const void *gate, int gate_is_i64,
                                    const float *h_coarse, const float *h_median,
                                    const float *h_fine,
                                    int B, int C, int hc, int wc,
                                    float *h_out, int64_t *indices, float *cmask)
{
    int H = 4 * hc, W = 4 * wc, hm = 2 * hc, wm = 2 * wc;
    for (long cell = 0; cell < (long)B * hc * wc; ++cell)
        indices[cell] = gate_is_i64 ? argmax_i64((const int64_t *)gate + cell * 3, 3)
                                    : argmax_f32((const float *)gate + cell * 3, 3);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    long cell = ((long)b * hc + y / 4) * wc + x / 4;
                    long o = (((long)b * C + c) * H + y) * W + x;
                    long ic = (((long)b * C + c) * hc + y / 4) * wc + x / 4;
                    long im = (((long)b * C + c) * hm + y / 2) * wm + x / 2;
                    int64_t g = indices[cell];
                    h_out[o] = g == 0 ? h_coarse[ic] : (g == 1 ? h_median[im] : h_fine[o]);
                }
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long cell = ((long)b * hc + y / 4) * wc + x / 4;
                int64_t g = indices[cell];
                cmask[((long)b * H + y) * W + x] = g == 0 ? 0.0625f : (g == 1 ? 0.25f : 1.0f);
            }
}
