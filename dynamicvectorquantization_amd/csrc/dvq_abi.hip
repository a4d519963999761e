// dvq_abi.hip -- the extern "C" boundary of libdvq.so (see include/dvq.h).
// Argument validation + launch only: no allocation, no synchronisation, no torch types.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dvq.h"
#include "dvq_filter.h"

#define DVQ_VERSION 600   // 0.6.0 (include/dvq.h lists what each version changed)
#define DVQ_ROUTE_MAX_CELLS_ABI 1024   // = DVQ_ROUTE_MAX_CELLS (dvq_filter.h)

static thread_local char g_err[512] = "";

void dvq_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// launchers implemented next to their kernels
int dvq_launch_prep_f32(const float *E, int K, int D, void *prep, hipStream_t st);
int dvq_launch_prep_f16(const float *E, int K, int D, void *prep, hipStream_t st);
int dvq_launch_exact(const float *z, const float *prep, const float *E, const float *mask,
                     int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                     const DvqRouted *rv, hipStream_t st);
int dvq_launch_filter(const float *z, const void *prep, const float *E, const float *mask,
                      int D, int HW, int K, long N, float *zq, long long *codes, double *partials,
                      void *ws_extra, bool pass1_only, bool force_wide, float *loss, float beta,
                      const DvqRouted *rv, hipStream_t st, const DvqConv *cv, const DvqFold *fd, bool ws_clean);
int dvq_launch_routed(int G, int gate_mode, const void *gate, float thr, const float *h_coarse,
                      const float *h_median, const float *h_fine, const void *prep, const float *E,
                      int B, int D, int hc, int wc, int K, float beta, float *zq, long long *codes,
                      float *loss, long long *indices, float *cmask, long long *gate_out,
                      double *partials, void *ws_extra, bool exact, bool pass1_only, hipStream_t st, const DvqConv *cv,
                      const DvqFold *fd, bool ws_clean);
size_t dvq_fold_prep_bytes_impl(int K, int D);
int dvq_launch_fold_prep(const float *E, int K, int D, const void *cbprep, const float *Wt, const float *bias, void *fprep,
                         hipStream_t st);
size_t dvq_filter_ws_extra_bytes(int D, int HW, int K, long N);
bool dvq_filter_supported(int D, int HW, int K, long N);
int dvq_launch_loss_finalize(const double *partials, int nparts, double inv_numel, float beta,
                             float *loss, hipStream_t st);
int dvq_launch_route_select(int G, int gate_mode, const void *gate, const float *h_coarse,
                            const float *h_median, const float *h_fine, int B, int C, int hc, int wc,
                            float *h_out, long long *indices, float *cmask, float thr, long long *gate_out,
                            hipStream_t st);
int dvq_launch_entropy_gate(const float *ent, long n, float thr, long long *gate, hipStream_t st);
int dvq_launch_vq_backward_z(const float *z, const float *E, const long long *codes, const float *mask, const float *g_zq,
                             const float *g_loss, float coef_scale, int D, int HW, int K, long N, float *gz, hipStream_t st);
int dvq_launch_codebook_grad(const float *z, const float *E, const long long *codes, const float *mask, const float *g_loss,
                             float coef_scale, int D, int HW, long N, int K, float *gw, hipStream_t st);
int dvq_launch_embed_gather(const float *E, int K, int D, const long long *idx, long n, float *out,
                            hipStream_t st);

int dvq_launch_permute_count(const long long *grain, int B, int ncell, int *counts, int *maxes, hipStream_t st);
int dvq_launch_permute_forward(const long long *codes, const long long *grain, int B, int hc, int wc,
                               int row_first, int Lc, int Lf, const long long *special,
                               long long *cc, long long *cp, long long *cs, long long *fc, long long *fp,
                               long long *fs, hipStream_t st);
int dvq_launch_permute_backward(const long long *cc, const long long *fc, const long long *cp,
                                const long long *fp, int B, int Lc, int Lf, int hc, int wc,
                                long long cpos_eos, long long fpos_eos, long long *target, hipStream_t st);
int dvq_permute_max_cells(void);
int dvq_launch_ema_accumulate(const float *z, const long long *codes, int D, int HW, long N, int K,
                              float *cluster_size, float *vectors_sum, hipStream_t st);
int dvq_launch_entropy_map(const float *img, int B, int H, int W, float *out, hipStream_t st);
size_t dvq_router_gate_ws_bytes(int nb, int B, int C, int hc, int wc, int groups, int Hid);
size_t dvq_router_gate_prep_bytes_impl(int nb, int C, int Hid);
int dvq_launch_router_gate_prepare(const float *W1, int nb, int C, int Hid, void *prep, hipStream_t st);
int dvq_launch_restart_pick(unsigned long long seed, long long n, int k, long long *out, hipStream_t st);
int dvq_launch_ema_update(const float *stats_sum, const float *stats_count, float decay, float eps, int K, int D,
                          const float *cs_old, float *cs_new, float *embed_ema, float *weight, int restart, const float *restart_rows,
                          const float *z, int HW, const long long *pick, hipStream_t st);
int dvq_launch_router_gate_prepare_norm(const float *const *gn_w, const float *const *gn_b, int nb, int C, int Hid, void *prep,
                                        hipStream_t st);
int dvq_launch_router_gate(int nb, const float *const *h, const float *const *gn_w, const float *const *gn_b,
                           int B, int C, int hc, int wc, int groups, float eps,
                           const float *W1, const float *b1, const float *W2, const float *b2,
                           int Hid, int act, const void *w1_prep, float *gate, void *ws, hipStream_t st);

size_t dvq_qconv_prep_bytes_impl(int D);
int dvq_launch_qconv_prep(const float *Wt, const float *bias, int D, void *prep, hipStream_t st);
int dvq_launch_qconv(const float *x, const DvqRouted *rv, const void *prep, int D, int HW, long N, float *hout,
                     hipStream_t st);
int dvq_launch_filter_scores_debug(const float *tokens, int n, const void *prep, int D, int K, float *G,
                                   float *thr2W, float *xn, float *scale_b_out, hipStream_t st, int fold = 0);
size_t dvq_xch_bytes(long cpi, long gpi, int b_max, int num_codes);
int dvq_launch_xch_pack(const long long *codes, const long long *grain, const float *loss, double numel, int b_local,
                        int b_max, long cpi, long gpi, int num_codes, void *buf, hipStream_t st);
int dvq_launch_xch_unpack(const void *gathered, int world, int global_batch, long cpi, long gpi, int num_codes,
                          long long *codes, long long *grain, float *mean, hipStream_t st);

static int hip_rc(int rc, const char *what)
{
    if (rc == 0) return DVQ_OK;
    dvq_set_error("%s: HIP error %d (%s)", what, rc, hipGetErrorString((hipError_t)rc));
    return DVQ_EHIP;
}

static bool dim_ok(int D) { return D == 64 || D == 128 || D == 256; }

// loss partials: pass 1 / exact blocks (<= 2 ceil(N/128)), resolver blocks (<= max(32, ceil(N/128)))
static size_t partials_bytes_for(long N)
{
    size_t count = 6 * (size_t)((N + 127) / 128) + 160;    // resolver: one per 32 queued-token slots
    return (count * sizeof(double) + 255) / 256 * 256;
}

extern "C" {

int dvq_version(void) { return DVQ_VERSION; }
const char *dvq_last_error_string(void) { return g_err; }

size_t dvq_codebook_prep_bytes(int K, int D)
{
    if (K <= 0 || D <= 0) return 0;
    // f32 tile images + norms + f16 section (images of the filter kernel: 2 B per element + 16 B
    // of per-code metadata), rounded up generously so the layout can grow without an ABI change
    size_t f32 = dvq_prep_f16_offset(K, D);
    size_t f16 = 2 * ((size_t)dvq_num_tiles(K) * 32 * ((size_t)D * 2 + 16) + 4096);   // two tile images (32x32x16 / 16x16x32 order)
    return ((f32 + f16 + 255) / 256) * 256;
}

int dvq_codebook_prepare_f32(const float *codebook, int K, int D, void *prep, size_t prep_bytes,
                             void *stream)
{
    if (!codebook || !prep || K <= 0) { dvq_set_error("dvq_codebook_prepare_f32: null pointer or K <= 0"); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("dvq_codebook_prepare_f32: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", D); return DVQ_EUNSUPPORTED; }
    if (prep_bytes < dvq_codebook_prep_bytes(K, D)) { dvq_set_error("dvq_codebook_prepare_f32: prep buffer %zu < %zu bytes", prep_bytes, dvq_codebook_prep_bytes(K, D)); return DVQ_EWORKSPACE; }
    if (((uintptr_t)prep & 255) != 0) { dvq_set_error("dvq_codebook_prepare_f32: prep must be 256-byte aligned"); return DVQ_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    int rc = dvq_launch_prep_f32(codebook, K, D, prep, st);
    if (rc) return hip_rc(rc, "codebook_prep_f32");
    rc = dvq_launch_prep_f16(codebook, K, D, prep, st);
    return hip_rc(rc, "codebook_prep_f16");
}

size_t dvq_vq_assign_workspace_bytes(int B, int D, int HW, int K, int mode)
{
    if (B <= 0 || HW <= 0 || K <= 0 || D <= 0) return 0;
    mode &= ~DVQ_MODE_WS_CLEAN;
    long N = (long)B * HW;
    size_t partials = partials_bytes_for(N);
    size_t extra = (mode == DVQ_MODE_FILTER || mode == DVQ_MODE_FILTER_PASS1 || mode == DVQ_MODE_FILTER_WIDE) ? dvq_filter_ws_extra_bytes(D, HW, K, N) : 0;
    return partials + extra + 256;
}

size_t dvq_vq_assign_fallback_count_offset(int B, int D, int HW, int K)
{
    (void)D; (void)K;
    if (B <= 0 || HW <= 0) return 0;
    long N = (long)B * HW;
    return partials_bytes_for(N);
}

int dvq_vq_assign_nchw_f32(const float *z, const float *codebook, const void *prep,
                           const float *mask, int B, int D, int HW, int K, float beta,
                           float *zq, int64_t *codes, float *loss,
                           void *ws, size_t ws_bytes, int mode, void *stream)
{
    if (!z || !codebook || !prep || !codes) { dvq_set_error("dvq_vq_assign_nchw_f32: null pointer"); return DVQ_EINVAL; }
    if (B <= 0 || HW <= 0 || K <= 0) { dvq_set_error("dvq_vq_assign_nchw_f32: B=%d HW=%d K=%d must be positive", B, HW, K); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("dvq_vq_assign_nchw_f32: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", D); return DVQ_EUNSUPPORTED; }
    const bool ws_clean = (mode & DVQ_MODE_WS_CLEAN) != 0;
    mode &= ~DVQ_MODE_WS_CLEAN;
    const bool pass1_only = (mode == DVQ_MODE_FILTER_PASS1);
    const bool force_wide = (mode == DVQ_MODE_FILTER_WIDE);
    if (pass1_only || force_wide) mode = DVQ_MODE_FILTER;
    if (mode != DVQ_MODE_EXACT && mode != DVQ_MODE_FILTER) { dvq_set_error("dvq_vq_assign_nchw_f32: unknown mode %d", mode); return DVQ_EINVAL; }
    if ((size_t)B * HW * D >= ((size_t)1 << 40)) { dvq_set_error("dvq_vq_assign_nchw_f32: tensor too large"); return DVQ_EUNSUPPORTED; }
    const long N = (long)B * HW;
    if (loss && !ws) { dvq_set_error("dvq_vq_assign_nchw_f32: loss requested without workspace"); return DVQ_EINVAL; }
    if ((loss || mode == DVQ_MODE_FILTER) && (!ws || ws_bytes < dvq_vq_assign_workspace_bytes(B, D, HW, K, mode))) {
        dvq_set_error("dvq_vq_assign_nchw_f32: workspace %zu < %zu bytes", ws_bytes, dvq_vq_assign_workspace_bytes(B, D, HW, K, mode));
        return DVQ_EWORKSPACE;
    }
    if (ws && ((uintptr_t)ws & 255) != 0) { dvq_set_error("dvq_vq_assign_nchw_f32: workspace must be 256-byte aligned"); return DVQ_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    double *partials = loss ? (double *)ws : nullptr;
    size_t partials_bytes = partials_bytes_for(N);
    int nparts;
    int rc;
    if (mode == DVQ_MODE_FILTER && dvq_filter_supported(D, HW, K, N)) {
        rc = dvq_launch_filter(z, prep, codebook, mask, D, HW, K, N, zq, (long long *)codes, partials,
                               (char *)ws + partials_bytes, pass1_only, force_wide, loss, beta, nullptr, st, nullptr, nullptr, ws_clean);
        return hip_rc(rc, "vq_assign_filter");     // the loss finalize is fused into its last kernel
    } else {
        rc = dvq_launch_exact(z, (const float *)prep, codebook, mask, D, HW, K, N, zq,
                              (long long *)codes, partials, nullptr, st);
        if (rc) return hip_rc(rc, "vq_assign_exact");
        nparts = (int)((N + 127) / 128);
    }
    if (loss) {
        rc = dvq_launch_loss_finalize(partials, nparts, 1.0 / ((double)N * D), beta, loss, st);
        if (rc) return hip_rc(rc, "vq_loss_finalize");
    }
    return DVQ_OK;
}

// row-major latents: the same op on [N, D, 1] (the filter path reads / writes a token's row with 16-byte accesses when HW == 1)
int dvq_vq_assign_flat_f32(const float *z, const float *codebook, const void *prep, const float *mask,
                           int64_t N, int D, int K, float beta, float *zq, int64_t *codes, float *loss,
                           void *ws, size_t ws_bytes, int mode, void *stream)
{
    if (N <= 0 || N >= ((int64_t)1 << 31)) { dvq_set_error("dvq_vq_assign_flat_f32: N=%lld out of range", (long long)N); return DVQ_EINVAL; }
    return dvq_vq_assign_nchw_f32(z, codebook, prep, mask, (int)N, D, 1, K, beta, zq, codes, loss, ws, ws_bytes, mode, stream);
}

// ---- the 1x1 quant_conv fused into the assign (filter mode; D = 256) --------------------------------------------------
static int conv_desc(const char *fn, const void *qconv_prep, float *h_buf, int h_all, int D, DvqConv *cv)
{
    if (!qconv_prep) { dvq_set_error("%s: qconv_prep is required", fn); return DVQ_EINVAL; }
    if (h_all && !h_buf) { dvq_set_error("%s: h_all needs an h_buf", fn); return DVQ_EINVAL; }
    if (D != 256) { dvq_set_error("%s: the fused conv exists for D = 256 (got %d): use dvq_qconv_f32 + dvq_vq_assign_nchw_f32", fn, D); return DVQ_EUNSUPPORTED; }
    if (((uintptr_t)qconv_prep & 255) != 0 || ((uintptr_t)h_buf & 3) != 0) { dvq_set_error("%s: qconv_prep must be 256-byte aligned", fn); return DVQ_EINVAL; }
    cv->meta = (const QconvMeta *)qconv_prep;
    cv->wimg = (const char *)qconv_prep + 256;
    cv->bias = (const float *)((const char *)qconv_prep + 256 + (size_t)(D / 32) * qconv_tile_bytes(D));
    cv->h_buf = h_all ? h_buf : nullptr;                     // (without h_all the op touches no h_buf: nullable since 0.6.0)
    cv->h_all = h_all ? 1 : 0;
    return DVQ_OK;
}

int dvq_vq_assign_qconv_f32(const float *x, const void *qconv_prep, const float *codebook, const void *prep,
                            const float *mask, int B, int D, int HW, int K, float beta,
                            float *zq, int64_t *codes, float *loss, float *h_buf, int h_all,
                            void *ws, size_t ws_bytes, int mode, void *stream)
{
    const char *fn = "dvq_vq_assign_qconv_f32";
    if (!x || !codebook || !prep || !codes) { dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL; }
    if (B <= 0 || HW <= 0 || K <= 0) { dvq_set_error("%s: B=%d HW=%d K=%d must be positive", fn, B, HW, K); return DVQ_EINVAL; }
    const bool ws_clean = (mode & DVQ_MODE_WS_CLEAN) != 0;
    mode &= ~DVQ_MODE_WS_CLEAN;
    const bool pass1_only = (mode == DVQ_MODE_FILTER_PASS1);
    if (mode != DVQ_MODE_FILTER && !pass1_only) { dvq_set_error("%s: mode %d (the conv is fused into the filter path only)", fn, mode); return DVQ_EINVAL; }
    DvqConv cv;
    int rc = conv_desc(fn, qconv_prep, h_buf, h_all, D, &cv);
    if (rc) return rc;
    const long N = (long)B * HW;
    if (N >= (1L << 31) || (size_t)N * D >= ((size_t)1 << 40)) { dvq_set_error("%s: tensor too large", fn); return DVQ_EUNSUPPORTED; }
    if (!dvq_filter_supported(D, HW, K, N)) { dvq_set_error("%s: shape unsupported by the filter path (D=%d HW=%d K=%d: K < 2^20, D * HW < 2^29)", fn, D, HW, K); return DVQ_EUNSUPPORTED; }
    if (!ws || ws_bytes < dvq_vq_assign_workspace_bytes(B, D, HW, K, DVQ_MODE_FILTER)) {
        dvq_set_error("%s: workspace %zu < %zu bytes", fn, ws_bytes, dvq_vq_assign_workspace_bytes(B, D, HW, K, DVQ_MODE_FILTER));
        return DVQ_EWORKSPACE;
    }
    if (((uintptr_t)ws & 255) != 0) { dvq_set_error("%s: workspace must be 256-byte aligned", fn); return DVQ_EINVAL; }
    double *partials = loss ? (double *)ws : nullptr;
    rc = dvq_launch_filter(x, prep, codebook, mask, D, HW, K, N, zq, (long long *)codes, partials,
                           (char *)ws + partials_bytes_for(N), pass1_only, false, loss, beta, nullptr, (hipStream_t)stream, &cv, nullptr, ws_clean);
    return hip_rc(rc, fn);
}

// ---- the conv folded into the codebook (vq_fold.hip; filter path, codes [+ z_q := e[code]], no loss) -------------------------
static int fold_desc(const char *fn, const void *qconv_prep, const void *fold_prep, int D, DvqFold *fd)
{
    if (!qconv_prep || !fold_prep) { dvq_set_error("%s: qconv_prep and fold_prep are required", fn); return DVQ_EINVAL; }
    if (((uintptr_t)qconv_prep & 255) != 0 || ((uintptr_t)fold_prep & 255) != 0) { dvq_set_error("%s: qconv_prep / fold_prep must be 256-byte aligned", fn); return DVQ_EINVAL; }
    fd->fprep = (const char *)fold_prep;
    fd->cv.meta = (const QconvMeta *)qconv_prep;
    fd->cv.wimg = (const char *)qconv_prep + 256;
    fd->cv.bias = (const float *)((const char *)qconv_prep + 256 + (size_t)(D / 32) * qconv_tile_bytes(D));
    fd->cv.h_buf = nullptr;
    fd->cv.h_all = 0;
    return DVQ_OK;
}

size_t dvq_fold_prep_bytes(int K, int D)
{
    if (K <= 0 || !dim_ok(D)) return 0;
    return (dvq_fold_prep_bytes_impl(K, D) + 255) / 256 * 256;
}

int dvq_fold_prepare_f32(const float *codebook, int K, int D, const void *codebook_prep, const float *conv_weight,
                         const float *conv_bias, void *fold_prep, size_t fold_prep_bytes, void *stream)
{
    const char *fn = "dvq_fold_prepare_f32";
    if (!codebook || !codebook_prep || !conv_weight || !fold_prep || K <= 0) { dvq_set_error("%s: null pointer or K <= 0", fn); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("%s: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", fn, D); return DVQ_EUNSUPPORTED; }
    if (fold_prep_bytes < dvq_fold_prep_bytes(K, D)) { dvq_set_error("%s: fold_prep buffer %zu < %zu bytes", fn, fold_prep_bytes, dvq_fold_prep_bytes(K, D)); return DVQ_EWORKSPACE; }
    if (((uintptr_t)fold_prep & 255) != 0 || ((uintptr_t)codebook_prep & 255) != 0) { dvq_set_error("%s: prep buffers must be 256-byte aligned", fn); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_fold_prep(codebook, K, D, codebook_prep, conv_weight, conv_bias, fold_prep, (hipStream_t)stream), fn);
}

int dvq_vq_assign_fold_f32(const float *x, const void *qconv_prep, const void *fold_prep, const float *codebook,
                           const void *prep, int B, int D, int HW, int K, float *zq, int64_t *codes,
                           void *ws, size_t ws_bytes, int mode, void *stream)
{
    const char *fn = "dvq_vq_assign_fold_f32";
    const bool ws_clean = (mode & DVQ_MODE_WS_CLEAN) != 0;
    mode &= ~DVQ_MODE_WS_CLEAN;
    if (!x || !codebook || !prep || !codes) { dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL; }
    if (B <= 0 || HW <= 0 || K <= 0) { dvq_set_error("%s: B=%d HW=%d K=%d must be positive", fn, B, HW, K); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("%s: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", fn, D); return DVQ_EUNSUPPORTED; }
    DvqFold fd;
    int rc = fold_desc(fn, qconv_prep, fold_prep, D, &fd);
    if (rc) return rc;
    const long N = (long)B * HW;
    if (N >= (1L << 31) || (size_t)N * D >= ((size_t)1 << 40)) { dvq_set_error("%s: tensor too large", fn); return DVQ_EUNSUPPORTED; }
    if (!dvq_filter_supported(D, HW, K, N)) { dvq_set_error("%s: shape unsupported by the filter path (D=%d HW=%d K=%d: K < 2^20, D * HW < 2^29)", fn, D, HW, K); return DVQ_EUNSUPPORTED; }
    if (!ws || ws_bytes < dvq_vq_assign_workspace_bytes(B, D, HW, K, DVQ_MODE_FILTER)) {
        dvq_set_error("%s: workspace %zu < %zu bytes", fn, ws_bytes, dvq_vq_assign_workspace_bytes(B, D, HW, K, DVQ_MODE_FILTER));
        return DVQ_EWORKSPACE;
    }
    if (((uintptr_t)ws & 255) != 0) { dvq_set_error("%s: workspace must be 256-byte aligned", fn); return DVQ_EINVAL; }
    if (mode != DVQ_MODE_FILTER && mode != DVQ_MODE_FILTER_PASS1) { dvq_set_error("%s: mode %d (the fold is a form of the filter path)", fn, mode); return DVQ_EINVAL; }
    rc = dvq_launch_filter(x, prep, codebook, nullptr, D, HW, K, N, zq, (long long *)codes, nullptr,
                           (char *)ws + partials_bytes_for(N), mode == DVQ_MODE_FILTER_PASS1, false, nullptr, 0.0f, nullptr,
                           (hipStream_t)stream, nullptr, &fd, ws_clean);
    return hip_rc(rc, fn);
}

// token ids of a routed batch: at most one per output position
static long routed_ids(int nb, int B, long N) { (void)nb; (void)B; return N; }

static int routed_dims(int nb, int hc, int wc, int *SC)
{
    *SC = (nb == 2) ? 2 : 4;
    return (nb == 2 || nb == 3) && hc > 0 && wc > 0 && (long)hc * wc <= DVQ_ROUTE_MAX_CELLS_ABI;
}

size_t dvq_vq_assign_routed_workspace_bytes(int num_branches, int B, int D, int hc, int wc, int K, int mode)
{
    int SC;
    if (B <= 0 || D <= 0 || K <= 0 || !routed_dims(num_branches, hc, wc, &SC)) return 0;
    const int HWout = SC * hc * SC * wc;
    const long N = (long)B * HWout;
    (void)mode;                                          // sized for the filter path in every mode
    return partials_bytes_for(routed_ids(num_branches, B, N)) + dvq_filter_ws_extra_bytes(D, HWout, K, N) + 256;
}

size_t dvq_vq_assign_routed_fallback_count_offset(int num_branches, int B, int D, int hc, int wc, int K)
{
    int SC;
    (void)D; (void)K;
    if (B <= 0 || !routed_dims(num_branches, hc, wc, &SC)) return 0;
    return partials_bytes_for(routed_ids(num_branches, B, (long)B * SC * hc * SC * wc));
}

static int routed_common(const char *fn, int nb, const void *gate, int gate_kind, float threshold,
                         const float *h_coarse, const float *h_median, const float *h_fine,
                         const float *codebook, const void *prep, int B, int D, int hc, int wc, int K, float beta,
                         float *zq, int64_t *codes, float *loss, int64_t *indices, float *cmask, int64_t *gate_out,
                         void *ws, size_t ws_bytes, int mode, void *stream,
                         const void *qconv_prep = nullptr, float *h_buf = nullptr, int h_all = 0, bool conv = false,
                         const void *fold_prep = nullptr)
{
    if (!gate || !h_coarse || !h_fine || !codebook || !prep || !codes || !indices || !cmask || (nb == 3 && !h_median)) {
        dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL;
    }
    if (B <= 0 || hc <= 0 || wc <= 0 || K <= 0) { dvq_set_error("%s: B=%d hc=%d wc=%d K=%d must be positive", fn, B, hc, wc, K); return DVQ_EINVAL; }
    if (gate_kind != DVQ_GATE_F32 && gate_kind != DVQ_GATE_I64 && gate_kind != DVQ_GATE_ENTROPY) { dvq_set_error("%s: gate_kind %d", fn, gate_kind); return DVQ_EINVAL; }
    if (gate_kind == DVQ_GATE_ENTROPY && nb != 2) { dvq_set_error("%s: the entropy gate is a dual-granularity router", fn); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("%s: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", fn, D); return DVQ_EUNSUPPORTED; }
    int SC;
    if (!routed_dims(nb, hc, wc, &SC)) { dvq_set_error("%s: hc*wc=%ld exceeds %d coarse cells", fn, (long)hc * wc, DVQ_ROUTE_MAX_CELLS_ABI); return DVQ_EUNSUPPORTED; }
    const bool ws_clean = (mode & DVQ_MODE_WS_CLEAN) != 0;
    mode &= ~DVQ_MODE_WS_CLEAN;
    const bool pass1_only = (mode == DVQ_MODE_FILTER_PASS1);
    if (pass1_only) mode = DVQ_MODE_FILTER;
    if (mode != DVQ_MODE_EXACT && mode != DVQ_MODE_FILTER) { dvq_set_error("%s: unknown mode %d", fn, mode); return DVQ_EINVAL; }
    DvqConv cv;
    if (conv) {
        if (mode != DVQ_MODE_FILTER) { dvq_set_error("%s: mode %d (the conv is fused into the filter path only)", fn, mode); return DVQ_EINVAL; }
        const int rcv = conv_desc(fn, qconv_prep, h_buf, h_all, D, &cv);
        if (rcv) return rcv;
    }
    DvqFold fd;
    if (fold_prep != nullptr) {
        if (mode != DVQ_MODE_FILTER) { dvq_set_error("%s: mode %d (the fold is a form of the filter path)", fn, mode); return DVQ_EINVAL; }
        const int rcf = fold_desc(fn, qconv_prep, fold_prep, D, &fd);
        if (rcf) return rcf;
    }
    const long N = (long)B * SC * hc * SC * wc;
    if (N >= (1L << 31) || (size_t)N * D >= ((size_t)1 << 40) || B > 32768) { dvq_set_error("%s: tensor too large", fn); return DVQ_EUNSUPPORTED; }
    if (!dvq_filter_supported(D, SC * hc * SC * wc, K, N)) { dvq_set_error("%s: shape unsupported by the filter path (D=%d HW=%d K=%d: K < 2^20, D * HW < 2^29)", fn, D, SC * hc * SC * wc, K); return DVQ_EUNSUPPORTED; }
    if (!ws || ws_bytes < dvq_vq_assign_routed_workspace_bytes(nb, B, D, hc, wc, K, mode)) {
        dvq_set_error("%s: workspace %zu < %zu bytes", fn, ws_bytes, dvq_vq_assign_routed_workspace_bytes(nb, B, D, hc, wc, K, mode));
        return DVQ_EWORKSPACE;
    }
    if (((uintptr_t)ws & 255) != 0) { dvq_set_error("%s: workspace must be 256-byte aligned", fn); return DVQ_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    double *partials = loss ? (double *)ws : nullptr;
    const size_t pbytes = partials_bytes_for(routed_ids(nb, B, N));
    const int gmode = (gate_kind == DVQ_GATE_ENTROPY) ? 2 : (gate_kind == DVQ_GATE_I64 ? 1 : 0);
    int rc = dvq_launch_routed(nb, gmode, gate, threshold, h_coarse, h_median, h_fine, prep, codebook, B, D, hc, wc, K,
                               beta, zq, (long long *)codes, loss, (long long *)indices, cmask, (long long *)gate_out,
                               partials, (char *)ws + pbytes, mode == DVQ_MODE_EXACT, pass1_only, st, conv ? &cv : nullptr,
                               fold_prep != nullptr ? &fd : nullptr, ws_clean);
    return hip_rc(rc, fn);                                   // the loss finalize is part of the op in both modes
}

int dvq_vq_assign_routed_dual_f32(const void *gate, int gate_kind, float threshold,
                                  const float *h_coarse, const float *h_fine,
                                  const float *codebook, const void *prep,
                                  int B, int D, int hc, int wc, int K, float beta,
                                  float *zq, int64_t *codes, float *loss,
                                  int64_t *indices, float *cmask, int64_t *gate_out,
                                  void *ws, size_t ws_bytes, int mode, void *stream)
{
    return routed_common("dvq_vq_assign_routed_dual_f32", 2, gate, gate_kind, threshold, h_coarse, nullptr, h_fine,
                         codebook, prep, B, D, hc, wc, K, beta, zq, codes, loss, indices, cmask, gate_out, ws, ws_bytes,
                         mode, stream);
}

int dvq_vq_assign_routed_triple_f32(const void *gate, int gate_kind,
                                    const float *h_coarse, const float *h_median, const float *h_fine,
                                    const float *codebook, const void *prep,
                                    int B, int D, int hc, int wc, int K, float beta,
                                    float *zq, int64_t *codes, float *loss,
                                    int64_t *indices, float *cmask,
                                    void *ws, size_t ws_bytes, int mode, void *stream)
{
    if (!h_median) { dvq_set_error("dvq_vq_assign_routed_triple_f32: null h_median"); return DVQ_EINVAL; }
    return routed_common("dvq_vq_assign_routed_triple_f32", 3, gate, gate_kind, 0.0f, h_coarse, h_median, h_fine,
                         codebook, prep, B, D, hc, wc, K, beta, zq, codes, loss, indices, cmask, nullptr, ws, ws_bytes,
                         mode, stream);
}

int dvq_vq_assign_routed_qconv_dual_f32(const void *gate, int gate_kind, float threshold,
                                        const float *h_coarse, const float *h_fine, const void *qconv_prep,
                                        const float *codebook, const void *prep,
                                        int B, int D, int hc, int wc, int K, float beta,
                                        float *zq, int64_t *codes, float *loss,
                                        int64_t *indices, float *cmask, int64_t *gate_out, float *h_buf, int h_all,
                                        void *ws, size_t ws_bytes, int mode, void *stream)
{
    return routed_common("dvq_vq_assign_routed_qconv_dual_f32", 2, gate, gate_kind, threshold, h_coarse, nullptr, h_fine,
                         codebook, prep, B, D, hc, wc, K, beta, zq, codes, loss, indices, cmask, gate_out, ws, ws_bytes,
                         mode, stream, qconv_prep, h_buf, h_all, true);
}

int dvq_vq_assign_routed_qconv_triple_f32(const void *gate, int gate_kind,
                                          const float *h_coarse, const float *h_median, const float *h_fine,
                                          const void *qconv_prep, const float *codebook, const void *prep,
                                          int B, int D, int hc, int wc, int K, float beta,
                                          float *zq, int64_t *codes, float *loss,
                                          int64_t *indices, float *cmask, float *h_buf, int h_all,
                                          void *ws, size_t ws_bytes, int mode, void *stream)
{
    if (!h_median) { dvq_set_error("dvq_vq_assign_routed_qconv_triple_f32: null h_median"); return DVQ_EINVAL; }
    return routed_common("dvq_vq_assign_routed_qconv_triple_f32", 3, gate, gate_kind, 0.0f, h_coarse, h_median, h_fine,
                         codebook, prep, B, D, hc, wc, K, beta, zq, codes, loss, indices, cmask, nullptr, ws, ws_bytes,
                         mode, stream, qconv_prep, h_buf, h_all, true);
}

int dvq_vq_assign_routed_fold_dual_f32(const void *gate, int gate_kind, float threshold,
                                       const float *h_coarse, const float *h_fine, const void *qconv_prep,
                                       const void *fold_prep, const float *codebook, const void *prep,
                                       int B, int D, int hc, int wc, int K, float *zq, int64_t *codes,
                                       int64_t *indices, float *cmask, int64_t *gate_out,
                                       void *ws, size_t ws_bytes, int mode, void *stream)
{
    const char *fn = "dvq_vq_assign_routed_fold_dual_f32";
    if (!fold_prep) { dvq_set_error("%s: null fold_prep", fn); return DVQ_EINVAL; }
    return routed_common(fn, 2, gate, gate_kind, threshold, h_coarse, nullptr, h_fine, codebook, prep, B, D, hc, wc, K, 0.0f,
                         zq, codes, nullptr, indices, cmask, gate_out, ws, ws_bytes, mode, stream, qconv_prep,
                         nullptr, 0, false, fold_prep);
}

int dvq_vq_assign_routed_fold_triple_f32(const void *gate, int gate_kind,
                                         const float *h_coarse, const float *h_median, const float *h_fine,
                                         const void *qconv_prep, const void *fold_prep, const float *codebook, const void *prep,
                                         int B, int D, int hc, int wc, int K, float *zq, int64_t *codes,
                                         int64_t *indices, float *cmask, void *ws, size_t ws_bytes, int mode, void *stream)
{
    const char *fn = "dvq_vq_assign_routed_fold_triple_f32";
    if (!h_median || !fold_prep) { dvq_set_error("%s: null h_median / fold_prep", fn); return DVQ_EINVAL; }
    return routed_common(fn, 3, gate, gate_kind, 0.0f, h_coarse, h_median, h_fine, codebook, prep, B, D, hc, wc, K, 0.0f,
                         zq, codes, nullptr, indices, cmask, nullptr, ws, ws_bytes, mode, stream, qconv_prep,
                         nullptr, 0, false, fold_prep);
}

int dvq_vq_backward_nchw_f32(const float *z, const float *codebook, const int64_t *codes, const float *mask,
                             const float *g_zq, const float *g_loss, float coef_scale,
                             int B, int D, int HW, int K, float *g_z, void *stream)
{
    const char *fn = "dvq_vq_backward_nchw_f32";
    if (!z || !codebook || !codes || !g_z) { dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL; }
    if (!g_zq && !g_loss) { dvq_set_error("%s: neither g_zq nor g_loss given", fn); return DVQ_EINVAL; }
    if (B <= 0 || HW <= 0 || K <= 0 || D <= 0) { dvq_set_error("%s: sizes must be positive", fn); return DVQ_EINVAL; }
    if (D % 16 != 0) { dvq_set_error("%s: D=%d must be a multiple of 16", fn, D); return DVQ_EUNSUPPORTED; }
    if ((((uintptr_t)codebook) & 15) != 0) { dvq_set_error("%s: codebook must be 16-byte aligned", fn); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_vq_backward_z(z, codebook, (const long long *)codes, mask, g_zq, g_loss, coef_scale, D, HW, K,
                                           (long)B * HW, g_z, (hipStream_t)stream), fn);
}

int dvq_vq_backward_codebook_nchw_f32(const float *z, const float *codebook, const int64_t *codes, const float *mask,
                                      const float *g_loss, float coef_scale, int B, int D, int HW, int K,
                                      float *g_weight, void *stream)
{
    const char *fn = "dvq_vq_backward_codebook_nchw_f32";
    if (!z || !codebook || !codes || !g_loss || !g_weight) { dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL; }
    if (B <= 0 || HW <= 0 || K <= 0 || D <= 0) { dvq_set_error("%s: sizes must be positive", fn); return DVQ_EINVAL; }
    if (K > 8192) { dvq_set_error("%s: K=%d > 8192 (index_add the differences instead)", fn, K); return DVQ_EUNSUPPORTED; }
    return hip_rc(dvq_launch_codebook_grad(z, codebook, (const long long *)codes, mask, g_loss, coef_scale, D, HW, (long)B * HW, K,
                                           g_weight, (hipStream_t)stream), fn);
}

int dvq_embed_gather_f32(const float *codebook, int K, int D, const int64_t *idx, int64_t n,
                         float *out, void *stream)
{
    if (!codebook || !idx || !out) { dvq_set_error("dvq_embed_gather_f32: null pointer"); return DVQ_EINVAL; }
    if (K <= 0 || D <= 0 || n < 0) { dvq_set_error("dvq_embed_gather_f32: bad sizes"); return DVQ_EINVAL; }
    if (D % 4 != 0) { dvq_set_error("dvq_embed_gather_f32: D=%d must be a multiple of 4", D); return DVQ_EUNSUPPORTED; }
    if (n == 0) return DVQ_OK;
    return hip_rc(dvq_launch_embed_gather(codebook, K, D, (const long long *)idx, (long)n, out, (hipStream_t)stream), "embed_gather");
}

int dvq_entropy_gate_f32(const float *entropy, int64_t n, float thr, int64_t *gate, void *stream)
{
    if (!entropy || !gate) { dvq_set_error("dvq_entropy_gate_f32: null pointer"); return DVQ_EINVAL; }
    if (n < 0) { dvq_set_error("dvq_entropy_gate_f32: n < 0"); return DVQ_EINVAL; }
    if (n == 0) return DVQ_OK;
    return hip_rc(dvq_launch_entropy_gate(entropy, (long)n, thr, (long long *)gate, (hipStream_t)stream), "entropy_gate");
}

static int route_args_ok(const char *fn, const void *gate, int gate_dtype, const float *a, const float *b,
                         int B, int C, int hc, int wc, float *h_out, int64_t *indices, float *cmask)
{
    if (!gate || !a || !b || !h_out || !indices || !cmask) { dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL; }
    if (gate_dtype != DVQ_GATE_F32 && gate_dtype != DVQ_GATE_I64) { dvq_set_error("%s: gate_dtype %d", fn, gate_dtype); return DVQ_EINVAL; }
    if (B <= 0 || C <= 0 || hc <= 0 || wc <= 0) { dvq_set_error("%s: sizes must be positive", fn); return DVQ_EINVAL; }
    return DVQ_OK;
}

int dvq_route_select_dual_f32(const void *gate, int gate_dtype, const float *h_coarse,
                              const float *h_fine, int B, int C, int hc, int wc,
                              float *h_out, int64_t *indices, float *cmask, void *stream)
{
    int rc = route_args_ok("dvq_route_select_dual_f32", gate, gate_dtype, h_coarse, h_fine, B, C, hc, wc, h_out, indices, cmask);
    if (rc) return rc;
    return hip_rc(dvq_launch_route_select(2, gate_dtype == DVQ_GATE_I64, gate, h_coarse, nullptr, h_fine, B, C, hc, wc,
                                          h_out, (long long *)indices, cmask, 0.0f, nullptr, (hipStream_t)stream), "route_select_dual");
}

int dvq_route_select_dual_entropy_f32(const float *entropy, float threshold, const float *h_coarse,
                                      const float *h_fine, int B, int C, int hc, int wc,
                                      float *h_out, int64_t *indices, float *cmask, int64_t *gate_out,
                                      void *stream)
{
    int rc = route_args_ok("dvq_route_select_dual_entropy_f32", entropy, DVQ_GATE_F32, h_coarse, h_fine, B, C, hc, wc, h_out, indices, cmask);
    if (rc) return rc;
    return hip_rc(dvq_launch_route_select(2, 2, entropy, h_coarse, nullptr, h_fine, B, C, hc, wc,
                                          h_out, (long long *)indices, cmask, threshold, (long long *)gate_out,
                                          (hipStream_t)stream), "route_select_dual_entropy");
}

int dvq_route_select_triple_f32(const void *gate, int gate_dtype, const float *h_coarse,
                                const float *h_median, const float *h_fine, int B, int C, int hc,
                                int wc, float *h_out, int64_t *indices, float *cmask, void *stream)
{
    int rc = route_args_ok("dvq_route_select_triple_f32", gate, gate_dtype, h_coarse, h_fine, B, C, hc, wc, h_out, indices, cmask);
    if (rc) return rc;
    if (!h_median) { dvq_set_error("dvq_route_select_triple_f32: null h_median"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_route_select(3, gate_dtype == DVQ_GATE_I64, gate, h_coarse, h_median, h_fine, B, C, hc, wc,
                                          h_out, (long long *)indices, cmask, 0.0f, nullptr, (hipStream_t)stream), "route_select_triple");
}

int dvq_ema_accumulate_nchw_f32(const float *z, const int64_t *codes, int B, int D, int HW, int K,
                                float *cluster_size, float *vectors_sum, void *stream)
{
    if (!z || !codes || !cluster_size || !vectors_sum) { dvq_set_error("dvq_ema_accumulate_nchw_f32: null pointer"); return DVQ_EINVAL; }
    if (B <= 0 || D <= 0 || HW <= 0 || K <= 0) { dvq_set_error("dvq_ema_accumulate_nchw_f32: sizes must be positive"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_ema_accumulate(z, (const long long *)codes, D, HW, (long)B * HW, K, cluster_size, vectors_sum,
                                            (hipStream_t)stream), "ema_accumulate");
}

int dvq_restart_pick_i64(uint64_t seed, int64_t n, int k, int64_t *out, void *stream)
{
    if (!out) { dvq_set_error("dvq_restart_pick_i64: null pointer"); return DVQ_EINVAL; }
    if (k < 1 || k > 2048 || n < 16 * (int64_t)k || n > 0xFFFFFFFFll) {
        dvq_set_error("dvq_restart_pick_i64: k=%d n=%lld (1 <= k <= 2048, 16 k <= n < 2^32)", k, (long long)n); return DVQ_EUNSUPPORTED;
    }
    return hip_rc(dvq_launch_restart_pick((unsigned long long)seed, (long long)n, k, (long long *)out, (hipStream_t)stream), "restart_pick");
}

int dvq_ema_update_f32(const float *stats_sum, const float *stats_count, float decay, float eps, int K, int D,
                       const float *cluster_size_ema, float *cluster_size_out, float *embed_ema, float *weight,
                       int restart, const float *restart_rows, const float *z, int B, int HW, const int64_t *pick, void *stream)
{
    if (!stats_sum || !stats_count || !cluster_size_ema || !cluster_size_out || !embed_ema || !weight) { dvq_set_error("dvq_ema_update_f32: null pointer"); return DVQ_EINVAL; }
    if (K <= 0 || D <= 0) { dvq_set_error("dvq_ema_update_f32: sizes must be positive"); return DVQ_EINVAL; }
    if (cluster_size_out == cluster_size_ema) { dvq_set_error("dvq_ema_update_f32: cluster_size_out must not alias cluster_size_ema (every workgroup sums the OLD counts)"); return DVQ_EINVAL; }
    if (restart < 0 || restart > 2) { dvq_set_error("dvq_ema_update_f32: restart must be 0, 1 or 2"); return DVQ_EINVAL; }
    if (restart == 1 && !restart_rows) { dvq_set_error("dvq_ema_update_f32: restart = 1 needs restart_rows"); return DVQ_EINVAL; }
    if (restart == 2 && (!z || !pick || B <= 0 || HW <= 0)) { dvq_set_error("dvq_ema_update_f32: restart = 2 needs z [B, D, HW] and pick [K]"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_ema_update(stats_sum, stats_count, decay, eps, K, D, cluster_size_ema, cluster_size_out, embed_ema, weight,
                                        restart, restart_rows, z, HW, (const long long *)pick, (hipStream_t)stream), "ema_update");
}

size_t dvq_router_gate_workspace_bytes(int num_branches, int B, int C, int hc, int wc, int num_groups, int hidden)
{
    if (num_branches < 2 || num_branches > 3 || B <= 0 || C <= 0 || hc <= 0 || wc <= 0 || num_groups < 0 || hidden < 0) return 0;
    return dvq_router_gate_ws_bytes(num_branches, B, C, hc, wc, num_groups, hidden > 0 ? hidden : 32);
}

size_t dvq_router_gate_prep_bytes(int num_branches, int C, int hidden)
{
    if (num_branches < 2 || num_branches > 3 || C <= 0 || hidden <= 0) return 0;
    return dvq_router_gate_prep_bytes_impl(num_branches, C, hidden);
}

int dvq_router_gate_prepare_f32(const float *w1, int nb, int C, int hidden, void *w1_prep, size_t w1_prep_bytes, void *stream)
{
    const char *fn = "dvq_router_gate_prepare_f32";
    if (nb != 2 && nb != 3) { dvq_set_error("%s: num_branches=%d (2 or 3)", fn, nb); return DVQ_EINVAL; }
    if (!w1 || !w1_prep || C <= 0 || hidden <= 0) { dvq_set_error("%s: null pointer or non-positive size", fn); return DVQ_EINVAL; }
    if (C % 8 != 0 || nb * C > 1280) { dvq_set_error("%s: C=%d unsupported (C %% 8 == 0, num_branches*C <= 1280)", fn, C); return DVQ_EUNSUPPORTED; }
    if (w1_prep_bytes < dvq_router_gate_prep_bytes(nb, C, hidden)) { dvq_set_error("%s: buffer %zu < %zu bytes", fn, w1_prep_bytes, dvq_router_gate_prep_bytes(nb, C, hidden)); return DVQ_EWORKSPACE; }
    if (((uintptr_t)w1_prep & 255) != 0) { dvq_set_error("%s: buffer must be 256-byte aligned", fn); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_router_gate_prepare(w1, nb, C, hidden, w1_prep, (hipStream_t)stream), "router_gate_prepare");
}

int dvq_router_gate_prepare_norm_f32(int nb, int C, int hidden, const float *gn_w_coarse, const float *gn_b_coarse,
                                     const float *gn_w_median, const float *gn_b_median,
                                     const float *gn_w_fine, const float *gn_b_fine, void *w1_prep, size_t w1_prep_bytes, void *stream)
{
    const char *fn = "dvq_router_gate_prepare_norm_f32";
    if (nb != 2 && nb != 3) { dvq_set_error("%s: num_branches=%d (2 or 3)", fn, nb); return DVQ_EINVAL; }
    if (!w1_prep || !gn_w_coarse || !gn_b_coarse || !gn_w_fine || !gn_b_fine || (nb == 3 && (!gn_w_median || !gn_b_median)) || C <= 0 || hidden <= 0) {
        dvq_set_error("%s: null pointer or non-positive size", fn); return DVQ_EINVAL;
    }
    if (w1_prep_bytes < dvq_router_gate_prep_bytes(nb, C, hidden)) { dvq_set_error("%s: buffer %zu < %zu bytes", fn, w1_prep_bytes, dvq_router_gate_prep_bytes(nb, C, hidden)); return DVQ_EWORKSPACE; }
    const float *w[3] = {gn_w_coarse, nb == 3 ? gn_w_median : gn_w_fine, gn_w_fine};
    const float *b[3] = {gn_b_coarse, nb == 3 ? gn_b_median : gn_b_fine, gn_b_fine};
    return hip_rc(dvq_launch_router_gate_prepare_norm(w, b, nb, C, hidden, w1_prep, (hipStream_t)stream), fn);
}

int dvq_router_gate_f32(int nb, const float *h_coarse, const float *h_median, const float *h_fine,
                        int B, int C, int hc, int wc, int num_groups, float eps,
                        const float *gn_w_coarse, const float *gn_b_coarse,
                        const float *gn_w_median, const float *gn_b_median,
                        const float *gn_w_fine, const float *gn_b_fine,
                        const float *w1, const float *b1, const float *w2, const float *b2,
                        int hidden, int activation, const void *w1_prep, float *gate, void *ws, size_t ws_bytes,
                        void *stream)
{
    const char *fn = "dvq_router_gate_f32";
    if (nb != 2 && nb != 3) { dvq_set_error("%s: num_branches=%d (2 or 3)", fn, nb); return DVQ_EINVAL; }
    if (!h_coarse || !h_fine || !w2 || !b2 || !gate) { dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL; }
    if ((nb == 3) != (h_median != nullptr)) { dvq_set_error("%s: h_median must be given exactly when num_branches == 3", fn); return DVQ_EINVAL; }
    if (B <= 0 || C <= 0 || hc <= 0 || wc <= 0) { dvq_set_error("%s: sizes must be positive", fn); return DVQ_EINVAL; }
    if (activation != DVQ_ACT_NONE && activation != DVQ_ACT_SILU && activation != DVQ_ACT_RELU) { dvq_set_error("%s: unknown activation %d", fn, activation); return DVQ_EINVAL; }
    if (activation != DVQ_ACT_NONE && (!w1 || !b1 || hidden <= 0)) { dvq_set_error("%s: hidden layer requested without w1 / b1 / hidden", fn); return DVQ_EINVAL; }
    if (num_groups < 0 || (num_groups > 0 && C % num_groups != 0)) { dvq_set_error("%s: C=%d is not divisible by num_groups=%d", fn, C, num_groups); return DVQ_EINVAL; }
    if (num_groups > 0 && (!gn_w_coarse || !gn_b_coarse || !gn_w_fine || !gn_b_fine || (nb == 3 && (!gn_w_median || !gn_b_median)))) {
        dvq_set_error("%s: GroupNorm affine parameters missing", fn); return DVQ_EINVAL;
    }
    if (C % 8 != 0 || nb * C > 1280) { dvq_set_error("%s: C=%d unsupported (C %% 8 == 0, num_branches*C <= 1280)", fn, C); return DVQ_EUNSUPPORTED; }
    if (num_groups > 0 && ((size_t)(C / num_groups) * hc * wc) % 4 != 0) { dvq_set_error("%s: group size not a multiple of 4 floats", fn); return DVQ_EUNSUPPORTED; }
    if (!ws || ws_bytes < dvq_router_gate_workspace_bytes(nb, B, C, hc, wc, num_groups, hidden)) {
        dvq_set_error("%s: workspace %zu < %zu bytes", fn, ws_bytes, dvq_router_gate_workspace_bytes(nb, B, C, hc, wc, num_groups, hidden));
        return DVQ_EWORKSPACE;
    }
    if (((uintptr_t)ws & 255) != 0) { dvq_set_error("%s: workspace must be 256-byte aligned", fn); return DVQ_EINVAL; }
    const float *h[3], *gw[3], *gb[3];
    if (nb == 2) {
        h[0] = h_coarse; h[1] = h_fine; h[2] = nullptr;
        gw[0] = gn_w_coarse; gw[1] = gn_w_fine; gw[2] = nullptr;
        gb[0] = gn_b_coarse; gb[1] = gn_b_fine; gb[2] = nullptr;
    } else {
        h[0] = h_coarse; h[1] = h_median; h[2] = h_fine;
        gw[0] = gn_w_coarse; gw[1] = gn_w_median; gw[2] = gn_w_fine;
        gb[0] = gn_b_coarse; gb[1] = gn_b_median; gb[2] = gn_b_fine;
    }
    return hip_rc(dvq_launch_router_gate(nb, h, gw, gb, B, C, hc, wc, num_groups, eps, w1, b1, w2, b2,
                                         activation == DVQ_ACT_NONE ? 32 : hidden, activation, w1_prep, gate, ws,
                                         (hipStream_t)stream), "router_gate");
}

int dvq_entropy_map_f32(const float *images, int B, int H, int W, int patch, float *out, void *stream)
{
    if (!images || !out) { dvq_set_error("dvq_entropy_map_f32: null pointer"); return DVQ_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0) { dvq_set_error("dvq_entropy_map_f32: sizes must be positive"); return DVQ_EINVAL; }
    if (patch != 16 || H % 16 != 0 || W % 16 != 0) { dvq_set_error("dvq_entropy_map_f32: patch=%d H=%d W=%d (patch 16, H and W multiples of 16)", patch, H, W); return DVQ_EUNSUPPORTED; }
    if ((long)B * (H / 16) * (W / 16) >= (1L << 30)) { dvq_set_error("dvq_entropy_map_f32: %ld patches (at most 2^30 per call)", (long)B * (H / 16) * (W / 16)); return DVQ_EUNSUPPORTED; }
    return hip_rc(dvq_launch_entropy_map(images, B, H, W, out, (hipStream_t)stream), "entropy_map");
}

static int perm_dims_ok(const char *fn, int B, int hc, int wc)
{
    if (B <= 0 || hc <= 0 || wc <= 0) { dvq_set_error("%s: sizes must be positive", fn); return DVQ_EINVAL; }
    if ((long)hc * wc > dvq_permute_max_cells()) { dvq_set_error("%s: hc*wc=%ld exceeds %d coarse cells", fn, (long)hc * wc, dvq_permute_max_cells()); return DVQ_EUNSUPPORTED; }
    return DVQ_OK;
}

int dvq_permute_dual_count_i64(const int64_t *grain, int B, int hc, int wc, int32_t *counts, int32_t *maxes,
                               void *stream)
{
    if (!grain || !counts || !maxes) { dvq_set_error("dvq_permute_dual_count_i64: null pointer"); return DVQ_EINVAL; }
    int rc = perm_dims_ok("dvq_permute_dual_count_i64", B, hc, wc);
    if (rc) return rc;
    return hip_rc(dvq_launch_permute_count((const long long *)grain, B, hc * wc, counts, maxes, (hipStream_t)stream), "permute_count");
}

int dvq_permute_dual_forward_i64(const int64_t *codes, const int64_t *grain, int B, int hc, int wc,
                                 int order, int Lc, int Lf, const int64_t *special,
                                 int64_t *coarse_content, int64_t *coarse_position, int64_t *coarse_segment,
                                 int64_t *fine_content, int64_t *fine_position, int64_t *fine_segment,
                                 void *stream)
{
    if (!codes || !grain || !special || !coarse_content || !coarse_position || !coarse_segment || !fine_content ||
        !fine_position || !fine_segment) { dvq_set_error("dvq_permute_dual_forward_i64: null pointer"); return DVQ_EINVAL; }
    int rc = perm_dims_ok("dvq_permute_dual_forward_i64", B, hc, wc);
    if (rc) return rc;
    if (order != 0 && order != 1) { dvq_set_error("dvq_permute_dual_forward_i64: order %d (0 region-first, 1 row-first)", order); return DVQ_EINVAL; }
    if (Lc <= 0 || Lf <= 0) { dvq_set_error("dvq_permute_dual_forward_i64: Lc, Lf must be positive"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_permute_forward((const long long *)codes, (const long long *)grain, B, hc, wc, order, Lc, Lf,
                                             (const long long *)special, (long long *)coarse_content,
                                             (long long *)coarse_position, (long long *)coarse_segment,
                                             (long long *)fine_content, (long long *)fine_position,
                                             (long long *)fine_segment, (hipStream_t)stream), "permute_forward");
}

int dvq_permute_dual_backward_i64(const int64_t *coarse_content, const int64_t *fine_content,
                                  const int64_t *coarse_position, const int64_t *fine_position,
                                  int B, int Lc, int Lf, int hc, int wc,
                                  int64_t coarse_position_eos, int64_t fine_position_eos,
                                  int64_t *target, void *stream)
{
    if (!coarse_content || !fine_content || !coarse_position || !fine_position || !target) { dvq_set_error("dvq_permute_dual_backward_i64: null pointer"); return DVQ_EINVAL; }
    int rc = perm_dims_ok("dvq_permute_dual_backward_i64", B, hc, wc);
    if (rc) return rc;
    if (Lc <= 0 || Lf <= 0) { dvq_set_error("dvq_permute_dual_backward_i64: Lc, Lf must be positive"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_permute_backward((const long long *)coarse_content, (const long long *)fine_content,
                                              (const long long *)coarse_position, (const long long *)fine_position,
                                              B, Lc, Lf, hc, wc, coarse_position_eos, fine_position_eos,
                                              (long long *)target, (hipStream_t)stream), "permute_backward");
}

size_t dvq_qconv_prep_bytes(int D) { return dim_ok(D) ? dvq_qconv_prep_bytes_impl(D) : 0; }

int dvq_qconv_prepare_f32(const float *weight, const float *bias, int D, void *prep, size_t prep_bytes, void *stream)
{
    if (!weight || !prep) { dvq_set_error("dvq_qconv_prepare_f32: null pointer"); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("dvq_qconv_prepare_f32: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", D); return DVQ_EUNSUPPORTED; }
    if (prep_bytes < dvq_qconv_prep_bytes(D)) { dvq_set_error("dvq_qconv_prepare_f32: prep buffer %zu < %zu bytes", prep_bytes, dvq_qconv_prep_bytes(D)); return DVQ_EWORKSPACE; }
    if (((uintptr_t)prep & 255) != 0) { dvq_set_error("dvq_qconv_prepare_f32: prep must be 256-byte aligned"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_qconv_prep(weight, bias, D, prep, (hipStream_t)stream), "qconv_prep");
}

int dvq_qconv_f32(const float *x, const void *prep, int B, int D, int HW, float *h, void *stream)
{
    if (!x || !prep || !h) { dvq_set_error("dvq_qconv_f32: null pointer"); return DVQ_EINVAL; }
    if (B <= 0 || HW <= 0) { dvq_set_error("dvq_qconv_f32: sizes must be positive"); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("dvq_qconv_f32: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", D); return DVQ_EUNSUPPORTED; }
    if ((long)B * HW >= (1L << 31)) { dvq_set_error("dvq_qconv_f32: tensor too large"); return DVQ_EUNSUPPORTED; }
    return hip_rc(dvq_launch_qconv(x, nullptr, prep, D, HW, (long)B * HW, h, (hipStream_t)stream), "qconv");
}

int dvq_qconv_select_f32(int num_branches, const void *gate, int gate_kind, float threshold,
                         const float *h_coarse, const float *h_median, const float *h_fine, const void *prep,
                         int B, int D, int hc, int wc, float *h, int64_t *indices, float *cmask, int64_t *gate_out,
                         void *stream)
{
    const char *fn = "dvq_qconv_select_f32";
    if (num_branches != 2 && num_branches != 3) { dvq_set_error("%s: num_branches=%d (2 or 3)", fn, num_branches); return DVQ_EINVAL; }
    if (!gate || !h_coarse || !h_fine || !prep || !h || !indices || !cmask || (num_branches == 3 && !h_median)) {
        dvq_set_error("%s: null pointer", fn); return DVQ_EINVAL;
    }
    if (B <= 0 || hc <= 0 || wc <= 0) { dvq_set_error("%s: sizes must be positive", fn); return DVQ_EINVAL; }
    if (gate_kind != DVQ_GATE_F32 && gate_kind != DVQ_GATE_I64 && gate_kind != DVQ_GATE_ENTROPY) { dvq_set_error("%s: gate_kind %d", fn, gate_kind); return DVQ_EINVAL; }
    if (gate_kind == DVQ_GATE_ENTROPY && num_branches != 2) { dvq_set_error("%s: the entropy gate is a dual-granularity router", fn); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("%s: D=%d unsupported (kernel widths 64, 128, 256; a multiple of 32 below 256 runs EXACTLY at the next width with zero channels appended to latents and codebook, as the Python drop-in does)", fn, D); return DVQ_EUNSUPPORTED; }
    const int SC = (num_branches == 2) ? 2 : 4;
    const long N = (long)B * SC * hc * SC * wc;
    if (N >= (1L << 31)) { dvq_set_error("%s: tensor too large", fn); return DVQ_EUNSUPPORTED; }
    DvqRouted rv{};
    rv.G = num_branches; rv.B = B; rv.D = D; rv.hc = hc; rv.wc = wc; rv.Wout = SC * wc; rv.HWout = SC * hc * SC * wc;
    rv.gate = gate; rv.gate_mode = (gate_kind == DVQ_GATE_ENTROPY) ? 2 : (gate_kind == DVQ_GATE_I64 ? 1 : 0);
    rv.thr = threshold; rv.indices = (const long long *)indices;
    rv.indices_out = (long long *)indices; rv.cmask_out = cmask; rv.gate_out = (long long *)gate_out;
    if (num_branches == 2) {
        rv.src[0] = h_coarse; rv.src[1] = h_fine; rv.sub[0] = 1; rv.sub[1] = 2; rv.sub[2] = 1; rv.rep[0] = 2; rv.rep[1] = 1; rv.rep[2] = 1;
    } else {
        rv.src[0] = h_coarse; rv.src[1] = h_median; rv.src[2] = h_fine;
        rv.sub[0] = 1; rv.sub[1] = 2; rv.sub[2] = 4; rv.rep[0] = 4; rv.rep[1] = 2; rv.rep[2] = 1;
    }
    return hip_rc(dvq_launch_qconv(nullptr, &rv, prep, D, rv.HWout, N, h, (hipStream_t)stream), fn);
}

int dvq_debug_filter_scores_f32(const float *tokens, int n, const void *prep, int D, int K, float *scores,
                                float *threshold, float *xn, float *scale, void *stream)
{
    if (!tokens || !prep || !scores || !threshold || !xn) { dvq_set_error("dvq_debug_filter_scores_f32: null pointer"); return DVQ_EINVAL; }
    if (n <= 0 || K <= 0) { dvq_set_error("dvq_debug_filter_scores_f32: n, K must be positive"); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("dvq_debug_filter_scores_f32: D=%d unsupported", D); return DVQ_EUNSUPPORTED; }
    return hip_rc(dvq_launch_filter_scores_debug(tokens, n, prep, D, K, scores, threshold, xn, scale, (hipStream_t)stream),
                  "filter_scores_debug");
}

int dvq_debug_fold_scores_f32(const float *tokens, int n, const void *fold_prep, int D, int K, float *scores,
                              float *threshold, float *xn, float *scale, void *stream)
{
    if (!tokens || !fold_prep || !scores || !threshold || !xn) { dvq_set_error("dvq_debug_fold_scores_f32: null pointer"); return DVQ_EINVAL; }
    if (n <= 0 || K <= 0) { dvq_set_error("dvq_debug_fold_scores_f32: n, K must be positive"); return DVQ_EINVAL; }
    if (!dim_ok(D)) { dvq_set_error("dvq_debug_fold_scores_f32: D=%d unsupported", D); return DVQ_EUNSUPPORTED; }
    return hip_rc(dvq_launch_filter_scores_debug(tokens, n, fold_prep, D, K, scores, threshold, xn, scale, (hipStream_t)stream, 1),
                  "fold_scores_debug");
}

size_t dvq_exchange_bytes(int64_t codes_per_image, int64_t grain_per_image, int b_max, int num_codes)
{
    if (codes_per_image <= 0 || grain_per_image < 0 || b_max <= 0 || num_codes <= 0) return 0;
    return dvq_xch_bytes((long)codes_per_image, (long)grain_per_image, b_max, num_codes);
}

int dvq_exchange_pack(const int64_t *codes, const int64_t *grain, const float *loss, double numel, int b_local,
                      int b_max, int64_t codes_per_image, int64_t grain_per_image, int num_codes, void *buf,
                      void *stream)
{
    if (!codes || !buf) { dvq_set_error("dvq_exchange_pack: null pointer"); return DVQ_EINVAL; }
    if (b_local < 0 || b_max <= 0 || b_local > b_max || codes_per_image <= 0 || grain_per_image < 0 || num_codes <= 0) {
        dvq_set_error("dvq_exchange_pack: bad sizes"); return DVQ_EINVAL;
    }
    if (grain_per_image > 0 && !grain) { dvq_set_error("dvq_exchange_pack: null grain"); return DVQ_EINVAL; }
    if (((uintptr_t)buf & 7) != 0) { dvq_set_error("dvq_exchange_pack: buffer must be 8-byte aligned"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_xch_pack((const long long *)codes, grain_per_image > 0 ? (const long long *)grain : nullptr,
                                      loss, numel, b_local, b_max, (long)codes_per_image, (long)grain_per_image,
                                      num_codes, buf, (hipStream_t)stream), "exchange_pack");
}

int dvq_exchange_unpack(const void *gathered, int world, int global_batch, int64_t codes_per_image,
                        int64_t grain_per_image, int num_codes, int64_t *codes, int64_t *grain, float *mean,
                        void *stream)
{
    if (!gathered || !codes) { dvq_set_error("dvq_exchange_unpack: null pointer"); return DVQ_EINVAL; }
    if (world <= 0 || global_batch <= 0 || codes_per_image <= 0 || grain_per_image < 0 || num_codes <= 0) {
        dvq_set_error("dvq_exchange_unpack: bad sizes"); return DVQ_EINVAL;
    }
    if (grain_per_image > 0 && !grain) { dvq_set_error("dvq_exchange_unpack: null grain"); return DVQ_EINVAL; }
    return hip_rc(dvq_launch_xch_unpack(gathered, world, global_batch, (long)codes_per_image, (long)grain_per_image,
                                        num_codes, (long long *)codes, grain_per_image > 0 ? (long long *)grain : nullptr,
                                        mean, (hipStream_t)stream), "exchange_unpack");
}

}  // extern "C"
