"""Encode glue of the hot path and its image-parallel (one process per GPU) form.

Mirrors the part of the reference's `encode` methods that follows the CNN trunk:
  DualGrainVQModel.encode    models/stage1_dynamic/dqvae_dual_feat.py:59-68, dqvae_dual_entropy.py:124-134
  TripleGrainVQModel.encode  models/stage1_dynamic/dqvae_triple_feat.py:68-77
  VQModel.encode             models/stage1/vqgan.py:68-72
i.e. router -> routing tail (select + codebook_mask) -> quant_conv -> quantizer, returning
(quant, emb_loss, info, grain_indices, gate) with the reference's shapes and dtypes.  The trunk
(dense convs / attention) is not part of this package; feed its branch outputs.

Sharding: images are independent, so rank r of G encodes the contiguous
slice [r*B/G, (r+1)*B/G) with the codebook replicated; the only exchange is ONE all-gather per
batch of a packed byte buffer (codes as int16/int32, grain indices as int8, the (loss-sum,
element-count) pair as float64) -- RCCL over xGMI when the process group is "nccl" -- which can run
asynchronously under the next batch's kernels.
"""
import torch
import torch.distributed as dist

from . import _lib, qconv
from .quantize import VectorQuantize2, vq_assign_routed_dual, vq_assign_routed_triple
from .router import DualGrainFixedEntropyRouter, route_select_dual, route_select_dual_entropy, route_select_triple


def _can_route(quantize, quant_conv, *feats):
    """the fused routed assign applies when nothing sits between select and quantizer and nothing needs a
    gradient (inference / frozen stage 1): an eval-mode VectorQuantize2 on NCHW feature maps, within the kernel's own
    preconditions (dvq_abi.hip: routed_common) -- D in (64, 128, 256), at most 1024 coarse cells per image, K < 2^20,
    assign mode EXACT or FILTER; any grid width (odd ones included).  Anything else takes select + dense assign."""
    if quant_conv is not None or not isinstance(quantize, VectorQuantize2):
        return False
    if quantize.training or not quantize.accept_image_fmap:
        return False
    if torch.is_grad_enabled() and any(t.requires_grad for t in feats):
        return False
    coarse = feats[0]
    if coarse.dim() != 4 or coarse.shape[1] not in (64, 128, 256) or coarse.shape[2] * coarse.shape[3] > 1024:
        return False
    if quantize.assign_mode not in (_lib.MODE_EXACT, _lib.MODE_FILTER) or quantize.codebook.n_embed >= (1 << 20):
        return False
    return all(t.is_cuda and t.dtype == torch.float32 for t in feats)


def _can_fuse_conv(quantize, quant_conv, *feats):
    """select + 1x1 quant_conv as one kernel (qconv.quant_conv_select): inference on fp32 GPU feature maps"""
    if quant_conv is None or not qconv.usable(quant_conv) or quant_conv.in_channels != feats[0].shape[1]:
        return False
    if torch.is_grad_enabled() and (any(t.requires_grad for t in feats) or
                                    any(p.requires_grad for p in quant_conv.parameters())):
        return False
    return all(t.is_cuda and t.dtype == torch.float32 for t in feats)


def _can_fuse_vqgan(quantize, quant_conv, h):
    """VectorQuantizer2 (quantize_vqgan.py:213-341) in eval mode behind a fusable 1x1 conv on 256 channels, no remap, filter path:
    quant_conv -> quantizer as one op (dvq_vq_assign_qconv_f32), as _can_route_conv allows it for VectorQuantize2"""
    from .quantize import VectorQuantizer2
    return (isinstance(quantize, VectorQuantizer2) and not quantize.training and quantize.remap is None
            and quantize.assign_mode == _lib.MODE_FILTER and quantize.e_dim == 256 and h.shape[1] == 256
            and quantize.embedding.weight.is_cuda and quantize.embedding.weight.dtype == torch.float32
            and quant_conv.out_channels == 256 and quantize.n_e < (1 << 20) and _can_fuse_conv(quantize, quant_conv, h)
            and not (torch.is_grad_enabled() and quantize.embedding.weight.requires_grad))


def _can_fold(quantize, quant_conv, *feats):
    """the conv folded into the codebook (quantize.vq_assign*(fold=True)): the routed op's preconditions, a fusable conv of the
    codebook's width (64 / 128 / 256 channels), the filter path.  Loss-free: callers ask for it."""
    return (_can_fuse_conv(quantize, quant_conv, *feats) and _can_route(quantize, None, *feats)
            and quantize.assign_mode == _lib.MODE_FILTER and quant_conv.out_channels == quantize.codebook.weight.shape[1])


def _can_route_conv(quantize, quant_conv, *feats):
    """the whole chain select -> 1x1 quant_conv -> quantizer as ONE routed op with the conv as pass 1's prologue
    (dvq_vq_assign_routed_qconv_*): the routed op's preconditions, a fusable conv, 256 channels, the filter path"""
    return (_can_fuse_conv(quantize, quant_conv, *feats) and feats[0].shape[1] == 256
            and _can_route(quantize, None, *feats) and quantize.assign_mode == _lib.MODE_FILTER)


def encode_dual(router, quantize, h_fine, h_coarse, entropy=None, quant_conv=None, temp=0.0, fold=False):
    """-> (quant, emb_loss, info, grain_indices, gate) as DualGrainVQModel.encode
    (dqvae_dual_feat.py:59-68, dqvae_dual_entropy.py:124-134).
    fold=True (opt-in, inference that needs no loss: reconstruction, tokenisation): the quant_conv folded into the codebook --
    no conv is computed for tokens the filter decides; emb_loss comes back as None, quant = codebook[codes] (within 1e-6 of the
    reference's z + (z_q - z)).  Falls back to the default paths when its preconditions do not hold.
    Inference paths (no autograd):
      * no quant_conv: gate + routing tail + quantizer as ONE routed op (the select fused into the assign);
      * a 1x1 quant_conv on 256 channels: the same ONE op with the conv as its prologue (h_dual and the conv's output are
        never written); other channel counts: select + conv as one kernel, then the dense assign;
    otherwise route select -> quant_conv -> dense assign as differentiable pieces."""
    fixed = isinstance(router, DualGrainFixedEntropyRouter) and entropy is not None and entropy.is_cuda
    if fold and _can_fold(quantize, quant_conv, h_coarse, h_fine):
        cb = quantize.codebook
        kw = dict(mode=quantize.assign_mode, conv=quant_conv, fold=True, want_loss=False)
        if fixed:
            r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, entropy=entropy,
                                      threshold=router.fine_grain_threshold, **kw)
        else:
            gate = router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy)
            r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, gate=gate, **kw)
        return r["zq"], None, (None, None, r["codes"]), r["indices"], r["gate"].permute(0, 3, 1, 2)
    if _can_route_conv(quantize, quant_conv, h_coarse, h_fine):
        cb = quantize.codebook
        kw = dict(beta=quantize.beta, mode=quantize.assign_mode, conv=quant_conv)
        if fixed:
            r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, entropy=entropy,
                                      threshold=router.fine_grain_threshold, **kw)
        else:
            gate = router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy)
            r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, gate=gate, **kw)
        return r["zq"], r["loss"][1], (None, None, r["codes"]), r["indices"], r["gate"].permute(0, 3, 1, 2)
    if _can_fuse_conv(quantize, quant_conv, h_coarse, h_fine):
        if fixed:
            sel = qconv.quant_conv_select(quant_conv, h_coarse, h_fine, entropy=entropy,
                                          threshold=router.fine_grain_threshold)
        else:
            sel = qconv.quant_conv_select(quant_conv, h_coarse, h_fine,
                                          gate=router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy))
        quant, emb_loss, info = quantize(x=sel["h"], temp=temp, codebook_mask=sel["codebook_mask"])
        return quant, emb_loss, info, sel["indices"], sel["gate"].permute(0, 3, 1, 2)
    if _can_route(quantize, quant_conv, h_coarse, h_fine):
        cb = quantize.codebook
        kw = dict(beta=quantize.beta, mode=quantize.assign_mode)
        if fixed:
            r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, entropy=entropy,
                                      threshold=router.fine_grain_threshold, **kw)
        else:
            gate = router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy)
            r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, gate=gate, **kw)
        return r["zq"], r["loss"][1], (None, None, r["codes"]), r["indices"], r["gate"].permute(0, 3, 1, 2)
    if fixed:
        sel = route_select_dual_entropy(entropy, router.fine_grain_threshold, h_coarse, h_fine)   # gate fused in
    else:
        gate = router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy)
        sel = route_select_dual(gate, h_coarse, h_fine)
    h = sel["h_dual"]
    if quant_conv is not None:
        h = quant_conv(h)
    quant, emb_loss, info = quantize(x=h, temp=temp, codebook_mask=sel["codebook_mask"])
    return quant, emb_loss, info, sel["indices"], sel["gate"]


def encode_triple(router, quantize, h_fine, h_median, h_coarse, quant_conv=None, temp=0.0, fold=False):
    """-> (quant, emb_loss, info, grain_indices, gate) as TripleGrainVQModel.encode (dqvae_triple_feat.py:68-77).
    fold=True: see encode_dual (emb_loss = None)."""
    gate = router(h_fine=h_fine, h_median=h_median, h_coarse=h_coarse, entropy=None)
    if fold and _can_fold(quantize, quant_conv, h_coarse, h_median, h_fine):
        cb = quantize.codebook
        r = vq_assign_routed_triple(h_coarse, h_median, h_fine, cb.codes, cb._prep, gate, mode=quantize.assign_mode,
                                    conv=quant_conv, fold=True, want_loss=False)
        return r["zq"], None, (None, None, r["codes"]), r["indices"], gate.permute(0, 3, 1, 2)
    if _can_route_conv(quantize, quant_conv, h_coarse, h_median, h_fine):
        cb = quantize.codebook
        r = vq_assign_routed_triple(h_coarse, h_median, h_fine, cb.codes, cb._prep, gate, beta=quantize.beta,
                                    mode=quantize.assign_mode, conv=quant_conv)
        return r["zq"], r["loss"][1], (None, None, r["codes"]), r["indices"], gate.permute(0, 3, 1, 2)
    if _can_fuse_conv(quantize, quant_conv, h_coarse, h_median, h_fine):
        sel = qconv.quant_conv_select(quant_conv, h_coarse, h_fine, h_median=h_median, gate=gate)
        quant, emb_loss, info = quantize(x=sel["h"], temp=temp, codebook_mask=sel["codebook_mask"])
        return quant, emb_loss, info, sel["indices"], gate.permute(0, 3, 1, 2)
    if _can_route(quantize, quant_conv, h_coarse, h_median, h_fine):
        cb = quantize.codebook
        r = vq_assign_routed_triple(h_coarse, h_median, h_fine, cb.codes, cb._prep, gate, beta=quantize.beta,
                                    mode=quantize.assign_mode)
        return r["zq"], r["loss"][1], (None, None, r["codes"]), r["indices"], gate.permute(0, 3, 1, 2)
    sel = route_select_triple(gate, h_coarse, h_median, h_fine)
    h = sel["h_triple"]
    if quant_conv is not None:
        h = quant_conv(h)
    quant, emb_loss, info = quantize(x=h, temp=temp, codebook_mask=sel["codebook_mask"])
    return quant, emb_loss, info, sel["indices"], sel["gate"]


def encode_fixed(quantize, h, quant_conv=None, fold=False):
    """-> (quant, emb_loss, info) as VQModel.encode (fixed granularity, models/stage1/vqgan.py:68-72).  An eval-mode
    VectorQuantize2 behind a 1x1 conv on 256 channels runs as one op (the conv is the assign's prologue).
    fold=True: the conv folded into the codebook (emb_loss = None), see encode_dual."""
    if fold and quant_conv is not None and h.dim() == 4 and _can_fold(quantize, quant_conv, h):
        from .quantize import vq_assign
        cb = quantize.codebook
        zq, codes, _ = vq_assign(h, cb.codes, cb._prep, None, mode=quantize.assign_mode, conv=quant_conv, fold=True, want_loss=False)
        return zq, None, (None, None, codes)
    if quant_conv is not None and h.dim() == 4 and _can_route_conv(quantize, quant_conv, h):
        from .quantize import vq_assign
        cb = quantize.codebook
        zq, codes, loss = vq_assign(h, cb.codes, cb._prep, None, beta=quantize.beta, mode=quantize.assign_mode, conv=quant_conv)
        return zq, loss[1], (None, None, codes)
    if quant_conv is not None and h.dim() == 4 and _can_fuse_vqgan(quantize, quant_conv, h):
        # the taming-style quantizer of the fixed-granularity VQModel (BASELINE configs[0]): the same ONE op; the loss
        # beta * m + m is the same float whichever of the two means carries beta (legacy), IEEE addition being commutative
        from .quantize import vq_assign
        zq, codes, loss = vq_assign(h, quantize.embedding.weight, quantize._prep, None, beta=float(quantize.beta),
                                    mode=quantize.assign_mode, conv=quant_conv)
        idx = codes if quantize.sane_index_shape else codes.reshape(-1)
        return zq, loss[1], (None, None, idx)
    if quant_conv is not None:
        h = qconv.quant_conv(quant_conv, h) if _can_fuse_conv(quantize, quant_conv, h) else quant_conv(h)
    return quantize(h)


def encode_to_tokens(router, quantize, permuter, h_fine, h_coarse, entropy=None, max_len=None, out=None, quant_conv=None,
                     fold=False):
    """Codes-only tokenisation for stage 2 (reference models/stage2_dynamic/dqtransformer_uncond_entropy.py:166-171,182:
    `_, z_out = self.encode_to_z(x)` keeps only `permuter(indices, grain_indices)` and discards quant): the routed assign
    with want_zq = False / want_loss = False (pass 1 writes no z_q: 1032 B per token instead of 2060) followed by the
    permuter on the same stream.  With `max_len` (see DualGrainSeperatePermuter.forward; `permuter.max_lengths()`) nothing
    is read back to the host: three kernels (counter zero, pass 1, resolver + list) and one permuter kernel, all queued.
    -> (permuter dict, grain_indices [B, hc, wc] int64, codes [B, 2hc, 2wc] int64).
    quant_conv: the first stage's 1x1 conv between select and quantizer (what `encode_to_z` runs through the stage-1 `encode`):
    fused into the same op (256 channels).  fold=True: that conv FOLDED into the codebook instead (64 / 128 / 256 channels):
    pass 1 scores the branches against E W and computes no conv at all -- the same codes as the fused op (both evaluate the
    reference chain on dvq_qconv_f32's h), at the speed of the conv-free tokenisation.
    Needs the fused routed op's preconditions (eval-mode VectorQuantize2, no autograd)."""
    if not _can_route(quantize, None, h_coarse, h_fine):
        raise _lib.DvqError("encode_to_tokens: needs an eval-mode VectorQuantize2 on fp32 GPU feature maps without autograd")
    if fold and (quant_conv is None or not _can_fold(quantize, quant_conv, h_coarse, h_fine)):
        raise _lib.DvqError("encode_to_tokens(fold=True): needs a 1x1 nn.Conv2d(D, D) quant_conv on the GPU and the filter mode")
    if quant_conv is not None and not fold and not _can_route_conv(quantize, quant_conv, h_coarse, h_fine):
        raise _lib.DvqError("encode_to_tokens: the quant_conv must be a 1x1 nn.Conv2d(256, 256) on the GPU (filter mode)")
    cb = quantize.codebook
    kw = dict(beta=quantize.beta, mode=quantize.assign_mode, want_zq=False, want_loss=False, conv=quant_conv, fold=bool(fold))
    if isinstance(router, DualGrainFixedEntropyRouter) and entropy is not None and entropy.is_cuda:
        r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep, entropy=entropy,
                                  threshold=router.fine_grain_threshold, **kw)
    else:
        r = vq_assign_routed_dual(h_coarse, h_fine, cb.codes, cb._prep,
                                  gate=router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy), **kw)
    seqs = permuter(r["codes"], r["indices"], max_len=max_len, out=out)
    return seqs, r["indices"], r["codes"]


def shard_slice(global_batch, rank, world_size):
    """contiguous image slice of `rank`; the first (global_batch % world_size) ranks get one extra"""
    base, extra = divmod(global_batch, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _wire_dtype(num_codes):
    return torch.int16 if num_codes <= 32768 else torch.int32


class _PendingGather:
    """Handle of an exchange in flight (all_gather_codes(..., async_op=True)): `.wait()` makes the
    current stream wait for the collective and returns (codes, grain_indices, mean)."""

    def __init__(self, work, unpack):
        self._work, self._unpack, self._result = work, unpack, None

    def wait(self):
        if self._result is None:
            if self._work is not None:
                self._work.wait()
            self._result = self._unpack()
        return self._result


def all_gather_codes(codes, grain_indices, loss_sum, numel, num_codes, global_batch, group=None,
                     async_op=False):
    """Exchange step of the image-parallel encode: ONE all-gather per batch.

    codes [b_local, H, W] int64, grain_indices [b_local, h, w] int64 or None, loss_sum 0-dim tensor
    (local sum of the loss numerator), numel = local element count.  Every rank packs
    [codes as int16/int32 | grain indices as int8 | (loss_sum, numel) as 2 x float64] into one byte
    buffer (neither RCCL nor gloo has an int16 type, so everything travels as uint8), the buffers
    are all-gathered, and each rank adds the per-rank loss pairs in rank order (same bits on every
    rank).  Returns (codes [B, H, W] int64, grain_indices [B, h, w] int64 or None, global mean) on
    every rank; with async_op=True a handle whose .wait() returns that tuple, so the exchange of
    batch i overlaps the kernels of batch i+1.  Ragged shards (B % world != 0) are padded to the
    largest shard on the wire."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_slice(global_batch, r, world) for r in range(world)]
    assert sizes[rank][1] - sizes[rank][0] == codes.shape[0]
    bmax = max(e - s for s, e in sizes)
    wd = _wire_dtype(num_codes)
    dev = codes.device
    esize = torch.empty((), dtype=wd).element_size()
    cshape, gshape = tuple(codes.shape[1:]), (tuple(grain_indices.shape[1:]) if grain_indices is not None else None)
    n_c = bmax * int(torch.Size(cshape).numel()) * esize
    n_g = bmax * int(torch.Size(gshape).numel()) if gshape is not None else 0
    n_c_pad = (n_c + 7) // 8 * 8
    n_g_pad = (n_g + 7) // 8 * 8
    nbytes = n_c_pad + n_g_pad + 16
    b_local = codes.shape[0]

    local = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    local[:b_local * (n_c // bmax)] = codes.to(wd).contiguous().view(torch.uint8).reshape(-1)
    if grain_indices is not None:
        local[n_c_pad:n_c_pad + b_local * (n_g // bmax)] = grain_indices.to(torch.int8).contiguous().view(torch.uint8).reshape(-1)
    pair = torch.stack([loss_sum.detach().to(torch.float64).reshape(()),
                        torch.full((), float(numel), dtype=torch.float64, device=dev)])
    local[n_c_pad + n_g_pad:] = pair.view(torch.uint8)
    out = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)

    def unpack():
        buf = out.view(world, nbytes)
        c_parts, g_parts = [], []
        for r, (s, e) in enumerate(sizes):
            c = buf[r, :n_c].view(wd).reshape((bmax,) + cshape)[: e - s]
            c_parts.append(c)
            if gshape is not None:
                g_parts.append(buf[r, n_c_pad:n_c_pad + n_g].view(torch.int8).reshape((bmax,) + gshape)[: e - s])
        pairs = buf[:, n_c_pad + n_g_pad:].contiguous().view(torch.float64).reshape(world, 2)
        tot = pairs.sum(0)                                   # rank order: identical on every rank
        g_codes = torch.cat(c_parts, 0).to(torch.int64)
        g_grain = torch.cat(g_parts, 0).to(torch.int64) if gshape is not None else None
        return g_codes, g_grain, (tot[0] / tot[1]).to(torch.float32)

    if async_op:
        return _PendingGather(work, unpack)
    return unpack()


class CodeExchange:
    """The exchange step with everything preallocated: per step ONE pack kernel (`dvq_exchange_pack`), ONE
    all-gather (RCCL when the group is "nccl"), ONE unpack kernel (`dvq_exchange_unpack`) -- no allocation,
    no small torch ops, no host round trip.  Same wire format as all_gather_codes.

        xch = CodeExchange(codes, grain, num_codes, global_batch, numel_per_image)   # shapes / device from templates
        xch.start(codes, grain, loss)     # loss[0] = local mean; queues pack + async all-gather
        ... next batch's kernels ...
        g_codes, g_grain, g_mean = xch.finish()   # stream-waits for the collective, queues the unpack

    CPU tensors (the gloo tests of the host logic) take the torch-op pack of all_gather_codes."""

    def __init__(self, codes, grain, num_codes, global_batch, numel_per_image, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_codes, self.global_batch, self.numel_per_image = num_codes, global_batch, numel_per_image
        sizes = [shard_slice(global_batch, r, self.world) for r in range(self.world)]
        self.shard_sizes = [e - s for s, e in sizes]
        self.b_local = self.shard_sizes[self.rank]
        self.b_max = max(self.shard_sizes)
        self.cshape = tuple(codes.shape[1:])
        self.gshape = tuple(grain.shape[1:]) if grain is not None else None
        self.cpi = int(torch.Size(self.cshape).numel())
        self.gpi = int(torch.Size(self.gshape).numel()) if self.gshape is not None else 0
        self.dev = codes.device
        self.on_gpu = codes.is_cuda
        self._pending = None
        self._result = None
        if self.on_gpu:
            self.nbytes = _lib.lib.dvq_exchange_bytes(self.cpi, self.gpi, self.b_max, num_codes)
            self.local = torch.empty(self.nbytes, dtype=torch.uint8, device=self.dev)
            self.gathered = torch.empty(self.world * self.nbytes, dtype=torch.uint8, device=self.dev)
            self.g_codes = torch.empty((global_batch,) + self.cshape, dtype=torch.int64, device=self.dev)
            self.g_grain = (torch.empty((global_batch,) + self.gshape, dtype=torch.int64, device=self.dev)
                            if self.gshape is not None else None)
            self.g_mean = torch.empty(1, dtype=torch.float32, device=self.dev)

    def start(self, codes, grain, loss):
        """loss: tensor whose element 0 is the local mean of the loss numerator (vq_assign's loss[0]), or None"""
        assert self._pending is None, "finish() the previous exchange first"
        numel = float(self.b_local * self.numel_per_image)
        if not self.on_gpu:
            lsum = (loss.reshape(-1)[0].double() * numel) if loss is not None else torch.zeros((), dtype=torch.float64)
            self._pending = all_gather_codes(codes, grain, lsum, numel if loss is not None else 0.0, self.num_codes,
                                             self.global_batch, group=self.group, async_op=True)
            return
        assert codes.is_contiguous() and codes.dtype == torch.int64 and codes.shape[0] == self.b_local
        with _lib.on_device(self.dev):
            _lib.check(_lib.lib.dvq_exchange_pack(
                codes.data_ptr(), _lib.ptr(grain) if self.gpi else 0, _lib.ptr(loss), numel, self.b_local, self.b_max,
                self.cpi, self.gpi, self.num_codes, self.local.data_ptr(), _lib.stream_ptr(self.dev)),
                "dvq_exchange_pack")
        self._pending = dist.all_gather_into_tensor(self.gathered, self.local, group=self.group, async_op=True)

    def finish(self):
        if self._pending is None:
            return self._result
        if not self.on_gpu:
            self._result = self._pending.wait()
            self._pending = None
            return self._result
        self._pending.wait()                       # the current stream waits for the collective
        self._pending = None
        with _lib.on_device(self.dev):
            _lib.check(_lib.lib.dvq_exchange_unpack(
                self.gathered.data_ptr(), self.world, self.global_batch, self.cpi, self.gpi, self.num_codes,
                self.g_codes.data_ptr(), _lib.ptr(self.g_grain), self.g_mean.data_ptr(), _lib.stream_ptr(self.dev)),
                "dvq_exchange_unpack")
        self._result = (self.g_codes, self.g_grain, self.g_mean[0])
        return self._result

    def result(self):
        return self._result


class StreamSlots:
    """Round-robin HIP streams for INDEPENDENT batches (inference / dataset encoding): batch i runs on slot i % n.

    One encode is a big power-limited kernel (pass 1) followed by a short latency-bound tail (resolver on ~300
    workgroups, list kernel + loss finalize, counter zero, the exchange's pack / unpack): on one stream the tail is
    dead time, on n = 3 streams it runs under the next batch's pass 1 (BASELINE configs[2]: 0.255 -> 0.223 ms per batch).
    The ops keep their workspaces per stream (quantize._CodebookPrep), so the only thing the caller owns per slot is the
    set of output tensors (or lets the ops allocate: torch's caching allocator is stream-aware).

        slots = StreamSlots(3)
        for i, batch in enumerate(loader):
            with slots.next() as slot:           # enters the slot's stream; it first waits for the caller's stream
                out[slot.index] = encode_dual(router, vq, *batch)
        slots.join()                             # the caller's stream waits for every slot

    Everything a slot's stream writes is ordered after the work queued on the caller's stream at `next()` time (inputs
    produced there are safe to read), and `join()` / `slot.wait()` order the caller's stream after the slot.
    The first call through a fresh quantizer builds the codebook image on whichever stream makes it: run one batch and
    synchronize (or call it on the caller's stream) before fanning out."""

    class _Slot:
        def __init__(self, index, stream):
            self.index, self.stream = index, stream
            self.done = torch.cuda.Event()
            self._ctx = None
            self._caller = None

        def __enter__(self):
            self._caller = torch.cuda.current_stream(self.stream.device)
            self.stream.wait_stream(self._caller)
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
            return self

        def __exit__(self, *exc):
            self.done.record(self.stream)
            ctx, self._ctx = self._ctx, None
            return ctx.__exit__(*exc)

        def wait(self):
            """the current stream waits for this slot's last batch"""
            torch.cuda.current_stream(self.stream.device).wait_event(self.done)

    def __init__(self, n=3, device=None):
        if n < 1:
            raise ValueError("StreamSlots needs at least one slot")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.slots = [self._Slot(i, torch.cuda.Stream(dev)) for i in range(n)]
        self._i = 0

    def __len__(self):
        return len(self.slots)

    def next(self):
        s = self.slots[self._i % len(self.slots)]
        self._i += 1
        return s

    def join(self):
        for s in self.slots:
            s.wait()

