/*
 * dvq.h -- C ABI of libdvq.so: the MI355X (gfx950) implementation of the DQ-VAE
 * vector-quantization hot path.
 *
 * The reference (Corleone-Huang/DynamicVectorQuantization) is pure Python and
 * has no FFI of its own; each entry point below replaces the listed PyTorch op
 * sequence of the reference, and is what a ctypes binding in the reference
 * would call (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch
 *     tensor's data_ptr()); nothing is allocated or freed inside;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls
 *     are stream-ordered, never synchronise, and are re-entrant: the library keeps
 *     no mutable process-wide state and reads no environment variable (kernel
 *     choices are compile-time constants; the A/B switches of the tuning build,
 *     libdvq_tuning.so made by `make tuning`, are not part of this ABI);
 *   - return value: DVQ_OK (0) or a negative DVQ_E* code; the message of the
 *     last failure on the calling thread is dvq_last_error_string();
 *   - tensors are dense, C-contiguous, float32 unless stated; code indices and
 *     grain indices are int64 (the reference's dtype);
 *   - channel counts (codebook_dim D): the assign kernels are instantiated for 64, 128 and
 *     256 (every reference config uses 256); other values return DVQ_EUNSUPPORTED.  A
 *     multiple of 32 below 256 (32, 96, 160, 192, 224) is served EXACTLY at the next kernel
 *     width by appending zero channels to latents and codebook -- a zero channel adds
 *     fma(0, 0, acc) = acc to the dot chain and + 0 to the norm's partial sums, whose 32-way
 *     grouping does not depend on D -- which is what the Python drop-in does
 *     (quantize.py: _padded_width); divide the returned loss mean by D / D_padded.
 *
 * Versions (dvq_version() = 100 major + minor; re-query every *_bytes function after an upgrade: buffer sizes are part of a version)
 *   0.6.0  dvq_ema_update_f32 (new); the conv-fused assigns (dvq_vq_assign_qconv_f32, dvq_vq_assign_routed_qconv_*_f32) take h_buf = NULL: no scratch tensor
 *          (0.3 - 0.5 required a full-size one); dvq_entropy_map_f32 refuses more than 2^30 patches per call; DVQ_MODE_WS_CLEAN's
 *          contract spelled out (valid for the same entry point and shape only).  No signature changed.
 *   0.5.0  DVQ_MODE_WS_CLEAN (self-cleaning workspace), dvq_vq_assign_flat_f32, dvq_restart_pick_i64,
 *          dvq_router_gate_prepare_norm_f32; DVQ_COUNTER_BYTES, the filter workspace (+ split buffer) and the gate's weight
 *          prep (+ 256-byte tail) grew: buffers sized by a 0.4.x build are too small for 0.5 and later.
 */
#ifndef DVQ_H_
#define DVQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libdvq.so is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table */
#define DVQ_API __attribute__((visibility("default")))

#define DVQ_OK            0
#define DVQ_EINVAL       (-1)  /* bad argument (null pointer, non-positive size, ...) */
#define DVQ_EUNSUPPORTED (-2)  /* shape outside what the kernels implement            */
#define DVQ_EWORKSPACE   (-3)  /* workspace / prep buffer too small                   */
#define DVQ_EHIP         (-4)  /* HIP runtime error on launch                         */

/* assign modes */
#define DVQ_MODE_EXACT   0  /* every (token, code) distance by the exact fp32 MFMA chain     */
#define DVQ_MODE_FILTER  1  /* fp16-MFMA filter with a rigorous error bound; every token that  */
                            /* is not provably decided is re-evaluated by the exact chain.     */
                            /* Output is identical to DVQ_MODE_EXACT.                          */
#define DVQ_MODE_FILTER_PASS1 2  /* profiling aid: ONLY the pass-1 (fp16 filter) kernel of DVQ_MODE_FILTER, */
                                 /* the dominant kernel; queued tokens keep their provisional code, the     */
                                 /* loss is not finalised.  Not a production mode.                          */
#define DVQ_MODE_FILTER_WIDE 3   /* testing aid: DVQ_MODE_FILTER with the two-blocks-per-wave pass-1 kernel */
                                 /* forced (D = 256); normally chosen automatically for K >= 2048 and       */
                                 /* >= 131072 tokens.  Same output.                                         */

/* Flag, OR-ed into `mode` of the filter-path ops (every dvq_vq_assign_* entry point; ignored in DVQ_MODE_EXACT): the caller
 * guarantees that this workspace is CLEAN FOR THIS CALL'S LAYOUT and that no other stream is using it.  Clean means one of:
 *   (a) the whole workspace was zero-filled (hipMemsetAsync / torch.zeros over all of its bytes) and no op has used it since, or
 *   (b) the last call that used it was a filter-path op WITHOUT DVQ_MODE_FILTER_PASS1 that returned DVQ_OK, through the SAME
 *       entry point and with the SAME shape arguments (B, D, HW or hc / wc, K) as this call.
 * (b) is shape-bound because the live words -- the 1-KiB counter block and the resolver's chunk tickets -- sit BEHIND the loss
 * partials, whose size is a function of B * HW, and the number of tickets is a function of the queue capacity (also B * HW): an
 * op leaves exactly ITS live words zero (its last consumer workgroup puts every counter back), not those of another shape's
 * layout.  A workspace that is merely large enough but was last used with other shape arguments is NOT clean: passing the flag
 * then lets pass 1 index its lists with stale counters (out-of-bounds device writes).  Keep one workspace per (stream, entry
 * point, shape) -- what the Python classes do, quantize._CodebookPrep.workspace -- or drop the flag when the shape changes.
 * With the flag no zeroing kernel is launched: one kernel boundary (~5 us) less per op.  Without it the op zeroes what it needs
 * first and any bytes are fine (the behaviour of versions before 0.5.0). */
#define DVQ_MODE_WS_CLEAN 0x100

/* gate kinds for the router select / routed assign */
#define DVQ_GATE_F32 0      /* float32 gate logits [.., G]                                      */
#define DVQ_GATE_I64 1      /* int64 gate [.., G] (what DualGrainFixedEntropyRouter returns)    */
#define DVQ_GATE_ENTROPY 2  /* routed assign only: float32 entropy map [B, hc, wc] + threshold  */
                            /* (the fixed-entropy router fused in)                              */

DVQ_API int dvq_version(void);
DVQ_API const char *dvq_last_error_string(void);

/*
 * Codebook preparation -- run once per codebook (weights change only in training).
 * Replaces: codebook_t.pow(2).sum(0)  (quantize2_mask.py:31,40) /
 *           torch.sum(embedding.weight**2, dim=1) (quantize_vqgan.py:281)
 * and lays the codebook out as the LDS tile images the assign kernels stream.
 *   codebook [K, D] (for VQEmbedding pass weight[:-1]); prep: >= dvq_codebook_prep_bytes(K, D)
 */
DVQ_API size_t dvq_codebook_prep_bytes(int K, int D);
DVQ_API int dvq_codebook_prepare_f32(const float *codebook, int K, int D,
                             void *prep, size_t prep_bytes, void *stream);

/*
 * Nearest-codebook assignment + quantised latents + commitment-loss sum.
 * Replaces: VectorQuantize2.forward (quantize2_mask.py:157-191: NCHW->NHWC copy,
 *           VQEmbedding.compute_distances :29-48, find_nearest_embedding :50-55,
 *           embed :130-132, masked loss :172-179, straight-through :182, NHWC->NCHW :187-189)
 *           and VectorQuantizer2.forward (quantize_vqgan.py:271-312).
 *   z        [B, D, HW]        (NCHW feature map, HW = H*W; HW == 1 is the flat [N, D] case)
 *   codebook [K, D], prep from dvq_codebook_prepare_f32 of the SAME codebook
 *   mask     nullable [B, HW]  (codebook_mask [B,1,H,W])
 *   zq       nullable [B, D, HW]   z + (e - z), two fp32 roundings like the reference
 *   codes    [B, HW] int64         first-index argmin, NaN = minimum (torch CPU semantics)
 *   loss     nullable [2] f32: loss[0] = mean((e-z)^2*mask), loss[1] = fl(fl(beta*mean)+mean)
 *   ws       >= dvq_vq_assign_workspace_bytes(...)
 * Distances follow the reference's fp32 arithmetic bit for bit: sequential-k FMA
 * chain, ATen-order norms, d = fl(fl(xn+en) - 2 dot).
 */
DVQ_API size_t dvq_vq_assign_workspace_bytes(int B, int D, int HW, int K, int mode);
DVQ_API int dvq_vq_assign_nchw_f32(const float *z, const float *codebook, const void *prep,
                           const float *mask, int B, int D, int HW, int K, float beta,
                           float *zq, int64_t *codes, float *loss,
                           void *ws, size_t ws_bytes, int mode, void *stream);

/*
 * The same op on ROW-MAJOR latents z [N, D] (a token's channels contiguous) -- what the reference builds before it calls the
 * codebook: `rearrange(x, 'b c h w -> b (h w) c')` (quantize2_mask.py:160-167; channel_last=True inputs arrive that way),
 * the concatenated item rows of VectorQuantize2List (quantize2_list.py:153-170) and VQEmbedding.forward's inputs [..., D]
 * (quantize2_mask.py:117-128).  mask nullable [N]; zq nullable [N, D]; codes [N] int64.  Equivalent to
 * dvq_vq_assign_nchw_f32 with B = N, HW = 1 (which takes the same row-major kernel form for HW == 1): in DVQ_MODE_FILTER
 * pass 1 reads and writes each row with 16-byte accesses (z and zq 16-byte aligned; otherwise, and in DVQ_MODE_EXACT,
 * lane-per-token 4-byte accesses).  Workspace: dvq_vq_assign_workspace_bytes(N, D, 1, K, mode).
 */
DVQ_API int dvq_vq_assign_flat_f32(const float *z, const float *codebook, const void *prep, const float *mask,
                           int64_t N, int D, int K, float beta, float *zq, int64_t *codes, float *loss,
                           void *ws, size_t ws_bytes, int mode, void *stream);

/* Diagnostic: byte offset inside the workspace of two int32 counters of the last DVQ_MODE_FILTER
 * call on that workspace: [0] tokens queued for the resolver (best and runner-up closer than the
 * error bound), [1] tokens handed to the full exact pass (non-finite / unscalable tokens,
 * overflow).  Read them after the stream has drained. */
DVQ_API size_t dvq_vq_assign_fallback_count_offset(int B, int D, int HW, int K);

/* nn.Embedding gather (quantize2_mask.py:130-132, get_codebook_entry :207-210):
 * out[n, :] = codebook[idx[n], :];  an index outside [0, K) writes NaNs to that row. */
DVQ_API int dvq_embed_gather_f32(const float *codebook, int K, int D, const int64_t *idx,
                         int64_t n, float *out, void *stream);

/* DualGrainFixedEntropyRouter.forward (RouterDual.py:53-57):
 * gate[i, 0] = entropy[i] <= thr, gate[i, 1] = entropy[i] > thr, int64. */
DVQ_API int dvq_entropy_gate_f32(const float *entropy, int64_t n, float thr, int64_t *gate,
                         void *stream);

/*
 * Routing tail of DualGrainEncoder.forward in eval mode (EncoderDual.py:134-149):
 * argmax over the 2 gate values, nearest x2 upsample of h_coarse, select, codebook_mask.
 *   gate [B, hc, wc, 2] (DVQ_GATE_F32 logits or DVQ_GATE_I64)
 *   h_coarse [B, C, hc, wc], h_fine [B, C, 2hc, 2wc]
 *   h_out [B, C, 2hc, 2wc], indices [B, hc, wc] int64, cmask [B, 1, 2hc, 2wc] (0.25 / 1.0)
 */
DVQ_API int dvq_route_select_dual_f32(const void *gate, int gate_dtype,
                              const float *h_coarse, const float *h_fine,
                              int B, int C, int hc, int wc,
                              float *h_out, int64_t *indices, float *cmask, void *stream);

/* The fixed-entropy router (RouterDual.py:46-57) fused into the dual select: entropy [B, hc, wc] f32,
 * gate = [(entropy <= threshold), (entropy > threshold)]; outputs as dvq_route_select_dual_f32 plus,
 * if gate_out != NULL, the router's int64 gate [B, hc, wc, 2] (DualGrainEncoder returns it). */
DVQ_API int dvq_route_select_dual_entropy_f32(const float *entropy, float threshold, const float *h_coarse,
                                      const float *h_fine, int B, int C, int hc, int wc,
                                      float *h_out, int64_t *indices, float *cmask, int64_t *gate_out,
                                      void *stream);

/*
 * Routing tail of TripleGrainEncoder.forward in eval mode (EncoderTriple.py:148-176).
 *   gate [B, hc, wc, 3]; h_coarse [B,C,hc,wc], h_median [B,C,2hc,2wc], h_fine [B,C,4hc,4wc]
 *   cmask values 0.0625 / 0.25 / 1.0
 */
DVQ_API int dvq_route_select_triple_f32(const void *gate, int gate_dtype,
                                const float *h_coarse, const float *h_median,
                                const float *h_fine, int B, int C, int hc, int wc,
                                float *h_out, int64_t *indices, float *cmask, void *stream);

/*
 * Routed assignment: routing tail + VectorQuantize2.forward as ONE op straight from the encoder
 * branches -- replaces dvq_route_select_{dual,triple}_f32 followed by dvq_vq_assign_nchw_f32
 * (EncoderDual.py:134-149 / EncoderTriple.py:148-176 + quantize2_mask.py:157-191) when nothing sits
 * between select and quantizer.  The select is fused into the assign's first kernel: every OUTPUT POSITION is a
 * token, read from the branch that won its cell (the grain is derived from the gate inside the kernel), scored,
 * and written; h_dual / h_triple is never materialised and indices / cmask / gate_out come out as by-products.
 * (The 2x2 / 4x4 positions of a coarse cell are copies of one vector and so get the same code; scoring each unique
 * vector once was built and measured slower -- DESIGN.md section 8.)  Same per-token arithmetic as the dense op:
 * codes, z_q, indices, cmask identical bit for bit to select + assign, loss within 1e-5.
 *   gate      DVQ_GATE_F32 / DVQ_GATE_I64: [B, hc, wc, G]; DVQ_GATE_ENTROPY (dual only): entropy [B, hc, wc]
 *             with `threshold` (gate = [(e <= thr), (e > thr)], written to gate_out [B, hc, wc, 2] if non-NULL)
 *   h_coarse  [B, D, hc, wc]; h_median [B, D, 2hc, 2wc] (triple); h_fine [B, D, S hc, S wc], S = 2 (dual) / 4 (triple)
 *   zq        nullable [B, D, S hc, S wc]; codes [B, S hc, S wc] int64; loss nullable [2] as dvq_vq_assign_nchw_f32
 *   indices   [B, hc, wc] int64 grain index per coarse cell; cmask [B, 1, S hc, S wc] (0.0625 / 0.25 / 1.0)
 *   ws        >= dvq_vq_assign_routed_workspace_bytes(num_branches, ...), 256-byte aligned
 *   mode      DVQ_MODE_EXACT / DVQ_MODE_FILTER (/ DVQ_MODE_FILTER_PASS1)
 * hc * wc <= 1024 coarse cells per image, B <= 32768; any hc, wc (32-wide output grids, i.e. every reference
 * config, take a form that stages the coarser branches through LDS).
 */
DVQ_API size_t dvq_vq_assign_routed_workspace_bytes(int num_branches, int B, int D, int hc, int wc, int K, int mode);
DVQ_API int dvq_vq_assign_routed_dual_f32(const void *gate, int gate_kind, float threshold,
                                  const float *h_coarse, const float *h_fine,
                                  const float *codebook, const void *prep,
                                  int B, int D, int hc, int wc, int K, float beta,
                                  float *zq, int64_t *codes, float *loss,
                                  int64_t *indices, float *cmask, int64_t *gate_out,
                                  void *ws, size_t ws_bytes, int mode, void *stream);
DVQ_API int dvq_vq_assign_routed_triple_f32(const void *gate, int gate_kind,
                                    const float *h_coarse, const float *h_median, const float *h_fine,
                                    const float *codebook, const void *prep,
                                    int B, int D, int hc, int wc, int K, float beta,
                                    float *zq, int64_t *codes, float *loss,
                                    int64_t *indices, float *cmask,
                                    void *ws, size_t ws_bytes, int mode, void *stream);
/*
 * The 1x1 quant_conv of the stage-1 models (nn.Conv2d(D, D, 1): dqvae_dual_feat.py:34,66, dqvae_triple_feat.py:39,75,
 * vqgan.py:42,70) as a GEMM on the fp16 matrix cores at fp32 grade (both operands split hi + lo, three MFMAs,
 * fp32 accumulation; 2^-22 products): equal to the reference's conv within 1e-5 relative to |x||w|, not bit for bit.
 *   prepare       weight [D, D] (= conv.weight[:, :, 0, 0], row = output channel), bias nullable [D]; run once per weight
 *   dvq_qconv_f32          x [B, D, HW] -> h [B, D, HW]
 *   dvq_qconv_select_f32   the router select fused in (replaces dvq_route_select_* followed by the conv: h_dual / h_triple
 *                          is never written): gate as for the routed assign, h_coarse / h_median / h_fine the encoder
 *                          branches; outputs h [B, D, S hc, S wc] plus the select's by-products indices, cmask, gate_out
 */
DVQ_API size_t dvq_qconv_prep_bytes(int D);
DVQ_API int dvq_qconv_prepare_f32(const float *weight, const float *bias, int D, void *prep, size_t prep_bytes, void *stream);
DVQ_API int dvq_qconv_f32(const float *x, const void *prep, int B, int D, int HW, float *h, void *stream);
DVQ_API int dvq_qconv_select_f32(int num_branches, const void *gate, int gate_kind, float threshold,
                         const float *h_coarse, const float *h_median, const float *h_fine, const void *prep,
                         int B, int D, int hc, int wc, float *h, int64_t *indices, float *cmask, int64_t *gate_out,
                         void *stream);

/*
 * The model order as ONE op: [select ->] 1x1 quant_conv -> assign, with the conv as the PROLOGUE of the assign's pass 1
 * (reference: `h = self.quant_conv(h_dual)` between the routing tail and `self.quantize`, dqvae_dual_entropy.py:124-134,
 * dqvae_dual_feat.py:59-68, dqvae_triple_feat.py:68-77, vqgan.py:68-72).  Each wave computes its tokens' conv output
 * straight into the registers pass 1 keeps its latents in (same split-fp16 arithmetic as dvq_qconv_f32; the per-token
 * scale follows the running maximum of the input channels): neither h_dual / h_triple nor the conv's output is written.
 * Contract: the conv output h the op scores is within 1e-5 * sum |w||x| of the fp64 conv; codes, z_q and loss are
 * bit-exact GIVEN that h.  D = 256 and DVQ_MODE_FILTER (or DVQ_MODE_FILTER_PASS1) only -- other sizes: dvq_qconv_* followed
 * by the assign (DVQ_EUNSUPPORTED otherwise).
 *   qconv_prep   the buffer dvq_qconv_prepare_f32 filled
 *   h_buf        NULL, or with h_all != 0 [B, D, HW] floats that receive the conv output the op scored, for EVERY token (how the
 *                tests and bench.py check the contract above).  Since 0.6.0 the op needs no scratch: the tokens pass 1 or the
 *                resolver hand to the exact-list kernel (non-finite or out-of-range latents, queue or candidate overflow) get
 *                their conv output computed by that kernel, from the conv's input, with dvq_qconv_f32's arithmetic (0.3.0 -
 *                0.5.0 REQUIRED a full-size h_buf for their rows: 256 MiB per stream at B = 256).  h_all == 0: h_buf is ignored.
 *   everything else as dvq_vq_assign_nchw_f32 / dvq_vq_assign_routed_{dual,triple}_f32 (x / h_coarse.. = the conv's INPUT)
 */
DVQ_API int dvq_vq_assign_qconv_f32(const float *x, const void *qconv_prep, const float *codebook, const void *prep,
                            const float *mask, int B, int D, int HW, int K, float beta,
                            float *zq, int64_t *codes, float *loss, float *h_buf, int h_all,
                            void *ws, size_t ws_bytes, int mode, void *stream);
DVQ_API int dvq_vq_assign_routed_qconv_dual_f32(const void *gate, int gate_kind, float threshold,
                                        const float *h_coarse, const float *h_fine, const void *qconv_prep,
                                        const float *codebook, const void *prep,
                                        int B, int D, int hc, int wc, int K, float beta,
                                        float *zq, int64_t *codes, float *loss,
                                        int64_t *indices, float *cmask, int64_t *gate_out, float *h_buf, int h_all,
                                        void *ws, size_t ws_bytes, int mode, void *stream);
DVQ_API int dvq_vq_assign_routed_qconv_triple_f32(const void *gate, int gate_kind,
                                          const float *h_coarse, const float *h_median, const float *h_fine,
                                          const void *qconv_prep, const float *codebook, const void *prep,
                                          int B, int D, int hc, int wc, int K, float beta,
                                          float *zq, int64_t *codes, float *loss,
                                          int64_t *indices, float *cmask, float *h_buf, int h_all,
                                          void *ws, size_t ws_bytes, int mode, void *stream);

/*
 * The quant_conv FOLDED into the codebook -- opt-in form of the model order for callers that need no loss: loss-free
 * inference and stage 2's tokenisation, which keeps the code indices only
 * (models/stage2_dynamic/dqtransformer_uncond_entropy.py:166-171,182 around models/stage1_dynamic/dqvae_dual_entropy.py:124-134).
 * With h = W x + bias the nearest code maximises  h.e_j - en_j/2 = x.(W^T e_j) + (bias.e_j - en_j/2):  pass 1 scores the conv's
 * INPUT against the image of E W (built in float64 by dvq_fold_prepare_f32, once per codebook / conv pair) and computes NO conv;
 * only tokens it cannot prove decided get their conv output (dvq_qconv_f32's arithmetic, bit for bit) and the reference's fp32
 * distance chain on it (resolver / exact-list kernel).  The decision bound covers every h within the conv contract's tolerance
 * (1e-5 * sum |w||x| of the real-number conv), so:
 *   codes = the reference argmin (quantize2_mask.py:29-55) evaluated on h = dvq_qconv_f32(x) -- identical to
 *           dvq_qconv_f32 followed by dvq_vq_assign_nchw_f32; versus another conv implementation within that tolerance, codes can
 *           differ only at near-ties of that tolerance (as for dvq_vq_assign_qconv_f32);
 *   zq    = nullable; codebook[code] for the tokens pass 1 / the resolver decide, fl(h + fl(e - h)) for the few the exact-list
 *           kernel handles: within 1e-6 relative of the reference's z_q given h (north_star tolerance 1e-5);
 *   no loss (the conv output of a decided token is never formed; use dvq_vq_assign_*qconv* when the loss is needed).
 * dvq_fold_prepare_f32: codebook [K, D] and ITS prep (dvq_codebook_prepare_f32), conv_weight [D, D] (row = output channel),
 * conv_bias nullable [D] -> fold_prep (>= dvq_fold_prep_bytes(K, D), 256-byte aligned).  qconv_prep: dvq_qconv_prepare_f32 of the
 * same weight / bias.  D in {64, 128, 256}.  Workspaces: dvq_vq_assign_workspace_bytes(.., DVQ_MODE_FILTER) /
 * dvq_vq_assign_routed_workspace_bytes; dvq_vq_assign_*fallback_count_offset apply.  mode: DVQ_MODE_FILTER (or
 * DVQ_MODE_FILTER_PASS1, the profiling aid).  Other arguments as the qconv forms.
 */
DVQ_API size_t dvq_fold_prep_bytes(int K, int D);
DVQ_API int dvq_fold_prepare_f32(const float *codebook, int K, int D, const void *codebook_prep, const float *conv_weight,
                         const float *conv_bias, void *fold_prep, size_t fold_prep_bytes, void *stream);
DVQ_API int dvq_vq_assign_fold_f32(const float *x, const void *qconv_prep, const void *fold_prep, const float *codebook,
                           const void *prep, int B, int D, int HW, int K, float *zq, int64_t *codes,
                           void *ws, size_t ws_bytes, int mode, void *stream);
DVQ_API int dvq_vq_assign_routed_fold_dual_f32(const void *gate, int gate_kind, float threshold,
                                       const float *h_coarse, const float *h_fine, const void *qconv_prep,
                                       const void *fold_prep, const float *codebook, const void *prep,
                                       int B, int D, int hc, int wc, int K, float *zq, int64_t *codes,
                                       int64_t *indices, float *cmask, int64_t *gate_out,
                                       void *ws, size_t ws_bytes, int mode, void *stream);
DVQ_API int dvq_vq_assign_routed_fold_triple_f32(const void *gate, int gate_kind,
                                         const float *h_coarse, const float *h_median, const float *h_fine,
                                         const void *qconv_prep, const void *fold_prep, const float *codebook, const void *prep,
                                         int B, int D, int hc, int wc, int K, float *zq, int64_t *codes,
                                         int64_t *indices, float *cmask, void *ws, size_t ws_bytes, int mode, void *stream);
/* audit aid, as dvq_debug_filter_scores_f32 for the folded image: tokens = the conv's inputs [n, D]; scores G'_j, the per-token
 * threshold 2W' (which also covers the conv tolerance), ||x||^2 and the scale 2^b' */
DVQ_API int dvq_debug_fold_scores_f32(const float *tokens, int n, const void *fold_prep, int D, int K, float *scores,
                              float *threshold, float *xn, float *scale, void *stream);

/*
 * Backward of the quantizer's forward with respect to its input -- what autograd derives from the reference graph
 * (quantize2_mask.py:172-182, quantize_vqgan.py:290-298): identity through z + (z_q - z).detach() plus the commitment term,
 *   g_z = g_zq + (g_loss * coef_scale) * ((z - e) * mask),   e = codebook[codes]  (the codebook AS IT WAS at forward time:
 *   callers keep a snapshot, the EMA update rewrites the weight between forward and backward), coef_scale = 2 c / numel.
 * One streaming pass (the reference: five or six element-wise passes and a transposed copy of e); same fp32 operation order
 * as that expression.  g_zq nullable (only the loss was used), g_loss nullable (only z_q was used; a device scalar otherwise),
 * mask nullable; D % 16 == 0.
 */
DVQ_API int dvq_vq_backward_nchw_f32(const float *z, const float *codebook, const int64_t *codes, const float *mask,
                             const float *g_zq, const float *g_loss, float coef_scale,
                             int B, int D, int HW, int K, float *g_z, void *stream);
/* ... and with respect to a codebook trained by back-propagation (VectorQuantizer2, quantize_vqgan.py:290-298; no EMA):
 *   g_weight[j, :] += -(g_loss * coef_scale) * sum over tokens with code j of (z - codebook[j]) * mask
 * ACCUMULATES into g_weight [>= K rows, D] (zero it first); float atomics, one row per distinct code of a 64-token tile:
 * equal to the reference's index_add_ of the [N, D] differences up to summation order.  K <= 8192. */
DVQ_API int dvq_vq_backward_codebook_nchw_f32(const float *z, const float *codebook, const int64_t *codes, const float *mask,
                                      const float *g_loss, float coef_scale, int B, int D, int HW, int K,
                                      float *g_weight, void *stream);

/* Audit aid (tools/bound_audit.py, tests/test_bound_audit.py): the pass-1 score arithmetic of DVQ_MODE_FILTER on
 * n tokens given as rows [n, D] -- every fp16-MFMA score G_j ~ -2^(b-1) (d_j - xn) as pass 1 sees it (index bits
 * packed into the low mantissa bits), the per-token decision threshold 2W, the exact norm xn and the codebook scale
 * 2^b -- so that the bound |G_j - truth_j| <= W can be checked against the reference arithmetic in float64 on the host.
 *   scores [n, 32*ceil(K/32)] (padding codes hold -3e38), threshold [n], xn [n], scale [1] (nullable) */
DVQ_API int dvq_debug_filter_scores_f32(const float *tokens, int n, const void *prep, int D, int K, float *scores,
                                float *threshold, float *xn, float *scale, void *stream);

/* as dvq_vq_assign_fallback_count_offset, for a routed workspace */
DVQ_API size_t dvq_vq_assign_routed_fallback_count_offset(int num_branches, int B, int D, int hc, int wc, int K);

/*
 * Training-mode codebook statistics, the dense part of VQEmbedding._update_buffers
 * (quantize2_mask.py:66-84: one-hot [K, N] matrix, row sum, matmul):
 *   cluster_size[j] = #tokens with code j, vectors_sum[j, :] = sum of those tokens' vectors.
 * z [B, D, HW] f32 (NCHW), codes [B, HW] int64; both outputs are overwritten.  Codes outside
 * [0, K) are ignored.  Float atomics: equal to the reference within rounding (1e-5).
 */
DVQ_API int dvq_ema_accumulate_nchw_f32(const float *z, const int64_t *codes, int B, int D, int HW, int K,
                                float *cluster_size, float *vectors_sum, void *stream);

/*
 * Dead-code restart of the training-mode codebook update: the reference takes the first K entries of torch.randperm(n_vectors)
 * as the input vectors that replace dead codes (quantize2_mask.py:93-96) -- on a GPU a full sort of n keys to keep K of them.
 * out [k] int64 = k DISTINCT indices of [0, n), every index equally likely, in random order (the distribution of a permutation's
 * prefix: independent draws, first occurrences kept), a pure function of (seed, n, k); one small workgroup.
 * 1 <= k <= 2048, 16 k <= n < 2^32.  RNG parity with torch is not possible either way.
 * (Distinctness is a practical, not a formal property: the kernel makes 2 k draws; should fewer than k different values occur among
 * them -- probability below 1e-100 for n >= 16 k -- a slot that stayed empty keeps its own index i, which may repeat a chosen one.)
 */
DVQ_API int dvq_restart_pick_i64(uint64_t seed, int64_t n, int k, int64_t *out, void *stream);

/*
 * The rest of the training-mode codebook update as ONE kernel (quantize2_mask.py:89-115; ~20 small torch kernels in the reference):
 *   cluster_size' = cluster_size_ema * decay + (1 - decay) * stats_count          (:89)
 *   embed_ema     = embed_ema * decay + (1 - decay) * stats_sum                   (:90)
 *   restart != 0: codes with cluster_size' < 1 take their restart row and count 1   (:102-105)
 *   n = sum(cluster_size'); weight[j, :] = embed_ema[j, :] / (n * (cluster_size'[j] + eps) / (n + K * eps))   (:107-115)
 * stats_sum [K, D] / stats_count [K]: this step's statistics (dvq_ema_accumulate_nchw_f32, all-reduced by the caller).
 * cluster_size_ema [K] is READ, the new counts go to cluster_size_out [K] (must not alias: every workgroup sums the old counts;
 * copy it over the buffer afterwards); embed_ema [K, D] is updated in place; weight: rows 0 .. K-1 of a [>= K, D] tensor are written.
 * restart: 0 none; 1 rows from restart_rows [K, D] (data-parallel: rank 0's, broadcast by the caller); 2 rows gathered from the NCHW
 * latents z [B, D, HW] at token index pick[j] in [0, B * HW) (dvq_restart_pick_i64).  fp32, the reference's operation order; the sum n is
 * accumulated in double: equal to the reference within rounding (1e-5).
 */
DVQ_API int dvq_ema_update_f32(const float *stats_sum, const float *stats_count, float decay, float eps, int K, int D,
                               const float *cluster_size_ema, float *cluster_size_out, float *embed_ema, float *weight,
                               int restart, const float *restart_rows, const float *z, int B, int HW, const int64_t *pick, void *stream);

/*
 * Fused feature-router gate (inference), the forward of DualGrainFeatureRouter
 * (modules/dynamic_modules/RouterDual.py:35-43) and TripleGrainFeatureRouter
 * (modules/dynamic_modules/RouterTriple.py:46-56): GroupNorm per branch, average pooling of the
 * finer branches onto the coarse grid, channel concat (coarse, [median,] fine), then
 *   activation == DVQ_ACT_NONE : gate = w2 x + b2                (gate_type "1layer-fc"; w1/b1 ignored,
 *                                                                 w2 is [nb, nb*C], hidden ignored)
 *   DVQ_ACT_SILU / DVQ_ACT_RELU: gate = w2 act(w1 x + b1) + b2   ("2layer-fc-SiLu" / "2layer-fc-ReLu";
 *                                                                 w1 [hidden, nb*C], w2 [nb, hidden])
 * num_branches nb = 2 (h_median must be NULL; h_fine is 2x the coarse grid) or 3 (median 2x, fine 4x).
 * h_* are [B, C, rows, cols] f32 NCHW; num_groups == 0 means normalization_type "none" (gn_* ignored),
 * otherwise gn_w_* / gn_b_* are the [C] affine parameters of each branch's GroupNorm(num_groups, C, eps).
 * gate [B, hc, wc, nb] f32 logits.  C % 8 == 0, nb*C <= 1280.  The hidden layer runs on the fp16 matrix
 * cores with both operands split hi + lo (hi*hi + hi*lo + lo*hi, fp32 accumulation: 2^-22 products);
 * summation order differs from ATen/MKL: logits equal the reference within 1e-4, not bit for bit.
 */
#define DVQ_ACT_NONE 0
#define DVQ_ACT_SILU 1
#define DVQ_ACT_RELU 2
/* The workspace holds the GroupNorm statistics and the per-cell averages of every branch (B * nb*C * hc*wc floats): the
 * features are read from HBM once.  w1_prep (nullable): the split fp16 tile images of w1 made by
 * dvq_router_gate_prepare_f32 into a caller-kept buffer of dvq_router_gate_prep_bytes -- valid while w1 is unchanged;
 * NULL: they are rebuilt inside the call. */
DVQ_API size_t dvq_router_gate_workspace_bytes(int num_branches, int B, int C, int hc, int wc, int num_groups, int hidden);
DVQ_API size_t dvq_router_gate_prep_bytes(int num_branches, int C, int hidden);
DVQ_API int dvq_router_gate_prepare_f32(const float *w1, int num_branches, int C, int hidden, void *w1_prep,
                                size_t w1_prep_bytes, void *stream);
/* Optional, after dvq_router_gate_prepare_f32 into the same buffer and again whenever a GroupNorm parameter changes: the
 * per-branch maxima of |weight| and |bias| of the branches' GroupNorms go into the buffer's tail; the gate's pooling pass then
 * derives the fp16 range of its operand images from six scalars instead of scanning all num_branches * C parameters in every
 * workgroup (-5 us of 83 at B = 128, -39 of 534 at B = 1024, triple).  Same results (the scale is an exact power of two). */
DVQ_API int dvq_router_gate_prepare_norm_f32(int num_branches, int C, int hidden,
                                     const float *gn_w_coarse, const float *gn_b_coarse,
                                     const float *gn_w_median, const float *gn_b_median,
                                     const float *gn_w_fine, const float *gn_b_fine,
                                     void *w1_prep, size_t w1_prep_bytes, void *stream);
DVQ_API int dvq_router_gate_f32(int num_branches, const float *h_coarse, const float *h_median, const float *h_fine,
                        int B, int C, int hc, int wc, int num_groups, float eps,
                        const float *gn_w_coarse, const float *gn_b_coarse,
                        const float *gn_w_median, const float *gn_b_median,
                        const float *gn_w_fine, const float *gn_b_fine,
                        const float *w1, const float *b1, const float *w2, const float *b2,
                        int hidden, int activation, const void *w1_prep, float *gate, void *ws, size_t ws_bytes,
                        void *stream);

/*
 * Patch-entropy map, Entropy.forward (models/stage1_dynamic/dqvae_dual_entropy.py:13-63) with
 * patch_size 16: images [B, 3, H, W] f32 (H, W multiples of 16) -> out [B, H/16, W/16] f32.
 * Transcendental fp32 math: equal to the reference within 1e-5, not bit for bit.
 */
DVQ_API int dvq_entropy_map_f32(const float *images, int B, int H, int W, int patch, float *out, void *stream);

/*
 * DualGrainSeperatePermuter (modules/dynamic_modules/permuter.py): dense codes + grain map <->
 * variable-length coarse / fine streams.  All tensors int64 like the reference.
 *   count    counts[B,2] = (#coarse, #fine cells) per image, maxes[2] = their batch maxima; the
 *            caller reads maxes (one device->host copy, as pad_sequence implies) and allocates
 *            Lc = maxes[0] + 1, Lf = 4*maxes[1] + 1.
 *   forward  permuter.py:50-109.  codes [B, 2hc, 2wc], grain [B, hc, wc] (0 coarse / 1 fine),
 *            order 0 = "region-first", 1 = "row-first"; special[6] (HOST array) = content_pad,
 *            content_eos, coarse_position_pad, coarse_position_eos, fine_position_pad,
 *            fine_position_eos; outputs [B, Lc] x3 (content, position, segment = 0) and [B, Lf] x3
 *            (content, position, segment = 1).  Entries that do not fit Lc / Lf are dropped.
 *   backward permuter.py:111-135 (sequential semantics: later entries win, entries after EOS
 *            ignored, no coarse upsample if the coarse stream has no EOS) -> target [B, 2hc, 2wc].
 * hc*wc <= 1024.
 */
DVQ_API int dvq_permute_dual_count_i64(const int64_t *grain, int B, int hc, int wc,
                               int32_t *counts, int32_t *maxes, void *stream);
DVQ_API int dvq_permute_dual_forward_i64(const int64_t *codes, const int64_t *grain, int B, int hc, int wc,
                                 int order, int Lc, int Lf, const int64_t *special,
                                 int64_t *coarse_content, int64_t *coarse_position, int64_t *coarse_segment,
                                 int64_t *fine_content, int64_t *fine_position, int64_t *fine_segment,
                                 void *stream);
DVQ_API int dvq_permute_dual_backward_i64(const int64_t *coarse_content, const int64_t *fine_content,
                                  const int64_t *coarse_position, const int64_t *fine_position,
                                  int B, int Lc, int Lf, int hc, int wc,
                                  int64_t coarse_position_eos, int64_t fine_position_eos,
                                  int64_t *target, void *stream);

/*
 * Wire format of the image-parallel exchange (one all-gather per batch; the reference gathers nothing --
 * it runs one process per GPU under Lightning DDP, train.py:230 -- this is what lets downstream consumers
 * (permuter, stage-2 transformer) see the global code tensor).  Per rank one byte buffer of
 * dvq_exchange_bytes(): [codes as int16 (num_codes <= 32768) / int32, b_max images][grain indices as int8]
 * [(loss numerator, element count) as 2 x float64].
 *   pack    codes [b_local, codes_per_image] int64, grain [b_local, grain_per_image] int64 (nullable when
 *           grain_per_image == 0), loss nullable ([0] = local mean; numel = local element count);
 *           rows b_local .. b_max-1 are zero padding (ragged shards)
 *   unpack  gathered [world, bytes] -> codes [global_batch, codes_per_image] int64, grain, mean[0] = global
 *           mean (pairs added in rank order: same bits on every rank); shard r holds the images
 *           [r*base + min(r, extra), ...) of global_batch = world*base + extra, b_max = ceil(global_batch / world)
 */
DVQ_API size_t dvq_exchange_bytes(int64_t codes_per_image, int64_t grain_per_image, int b_max, int num_codes);
DVQ_API int dvq_exchange_pack(const int64_t *codes, const int64_t *grain, const float *loss, double numel, int b_local,
                      int b_max, int64_t codes_per_image, int64_t grain_per_image, int num_codes, void *buf,
                      void *stream);
DVQ_API int dvq_exchange_unpack(const void *gathered, int world, int global_batch, int64_t codes_per_image,
                        int64_t grain_per_image, int num_codes, int64_t *codes, int64_t *grain, float *mean,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVQ_H_ */
