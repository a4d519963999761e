"""Drop-in quantizers backed by the HIP kernels of libdvq.so.

Mirrors (constructor kwargs, forward signature, return structure, state_dict keys):
  * VQEmbedding / VectorQuantize2  -- reference modules/vector_quantization/quantize2_mask.py:10-210
    (also covers quantize2.py: identical class without the mask argument)
  * VectorQuantizer2               -- reference modules/vector_quantization/quantize_vqgan.py:213-341
so a reference YAML selects them by changing only the `target:` string, e.g.
  target: dynamicvectorquantization_amd.quantize.VectorQuantize2

The NCHW->NHWC copy, the [N, K] distance matrix, argmin, embedding gather, masked loss,
straight-through add and the copy back of the reference are ONE kernel launch here
(`dvq_vq_assign_nchw_f32`) reading z once from NCHW and writing z_q once.  Code indices equal the
reference CPU path bit for bit, z_q too (two fp32 roundings), the loss to 1e-5 relative.
"""
import math

import torch
import torch.distributed as dist
from torch import nn
from torch.nn import functional as F

from . import _lib

_lib_handle = _lib.lib


class _Workspace:
    """One workspace of the filter-path ops (loss partials, queue counters, record stamps, lists, records) and whether it is
    CLEAN: every filter-path op leaves it clean (its last consumer workgroup puts the live counters back to zero), so from the
    second op on the steady state passes `DVQ_MODE_WS_CLEAN` and no zeroing kernel is launched (include/dvq.h).  One workspace
    per (stream, entry point, shape): the flag is shape-bound.
    `begin` marks it dirty until `end` has seen the call return DVQ_OK; the profiling mode MODE_FILTER_PASS1 leaves it dirty."""
    __slots__ = ("t", "clean")

    def __init__(self, nbytes, device):
        # not zero-filled (a memset of the whole workspace, 4-5 MB and more, per new shape cost more than the ~5-us zero kernel the
        # flag saves -- ADVICE r5): the FIRST filter-path op on it runs without the flag, zeroes what it needs and leaves it clean
        self.t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        self.clean = False

    def begin(self, mode):
        """-> (pointer, bytes, mode [| MODE_WS_CLEAN]) for the ABI call"""
        flag = _lib.MODE_WS_CLEAN if (self.clean and _lib.HAS_WS_CLEAN and mode in _lib.FILTER_MODES) else 0
        self.clean = False
        return self.t.data_ptr(), self.t.numel(), mode | flag

    def end(self, mode):
        self.clean = mode in (_lib.MODE_FILTER, _lib.MODE_FILTER_WIDE)

    def check(self, mode, rc, what):
        """ws.check(mode, lib.fn(..., *ws.begin(mode), ...), "fn"): raises on failure (the workspace then stays dirty)"""
        _lib.check(rc, what)
        self.end(mode)

    def __getitem__(self, sl):                      # (fallback_count reads two counters)
        return self.t[sl]


class _CodebookPrep:
    """Per-codebook device buffers of the assign kernels (tile images, exact norms), rebuilt
    whenever the codebook tensor changes: the key is (storage, autograd version, shape, device),
    and `invalidate()` drops it.  Writes through `.data` do NOT bump the version counter
    (`w.data.copy_(...)`, old-style optimizers): the owning modules call invalidate() from
    `_load_from_state_dict`, `_apply`, the EMA update and at every training-mode forward; any
    other `.data` writer must call `module.invalidate_codebook_cache()` itself.

    Workspaces (queue counters, records, loss partials) are kept PER STREAM, so two streams driving
    the same quantizer never share a queue.  The image itself is ONE buffer: the stream that (re)builds it records an
    event, every other stream waits for that event before its next use, and a rebuild first waits for the uses other
    streams have queued (encode.StreamSlots with a training-mode quantizer therefore serialises on the rebuild --
    correct, not fast; inference builds the image once)."""

    def __init__(self):
        self.key = None
        self.buf = None
        self._built = None       # (stream handle, event) of the last build
        self.track_users = False # set by training-mode forwards: the image may be rebuilt while other streams read it
        self._users = {}         # stream handle -> event after that stream's last use
        self._ws = {}            # (B, D, HW, K, mode, device, stream) -> uint8 tensor
        self._last_ws = None
        self._retired = []       # images replaced outside training: kept alive for readers other streams may still have queued
        self._fold = {}          # id(conv) -> (key, buffer, (stream, event)): the conv folded into this codebook (vq_fold.hip)

    def invalidate(self):
        self.key = None
        self._padded = None

    def get(self, codebook):
        K, D = codebook.shape
        key = (codebook.data_ptr(), codebook._version, K, D, codebook.device)
        if key == self.key:
            # steady state: no Stream object is made (torch.cuda.current_stream() alone is ~5 us of every call)
            b = self._built
            if b is None:
                return self.buf
            if b[0] == _lib.stream_ptr(codebook.device):     # built on this stream: ordered; forget the event once it has completed
                if not torch.cuda.is_current_stream_capturing() and b[1].query():
                    self._built = None
                return self.buf
        cur = torch.cuda.current_stream(codebook.device)
        if key != self.key:
            nbytes = _lib_handle.dvq_codebook_prep_bytes(K, D)
            if nbytes == 0:
                raise _lib.DvqError("unsupported codebook shape K=%d D=%d" % (K, D))
            if self.buf is None or self.buf.numel() < nbytes or self.buf.device != codebook.device or not self.track_users:
                # outside training (load_state_dict, invalidate, .to()) no use events exist: the new image goes to a FRESH
                # buffer and the old one is kept for whatever other streams still have queued against it
                if self.buf is not None:
                    self._retired = (self._retired + [self.buf])[-4:]
                self.buf = torch.empty(nbytes, dtype=torch.uint8, device=codebook.device)
            for h, ev in self._users.items():        # other streams may still be reading the old image
                if h != cur.cuda_stream:
                    cur.wait_event(ev)
            self._users.clear()
            self._retired = (self._retired + [ent[1] for ent in self._fold.values()])[-4:]
            self._fold.clear()
            _lib.check(_lib_handle.dvq_codebook_prepare_f32(
                codebook.data_ptr(), K, D, self.buf.data_ptr(), self.buf.numel(),
                cur.cuda_stream), "dvq_codebook_prepare_f32")
            ev = torch.cuda.Event()
            ev.record(cur)
            self._built = (cur.cuda_stream, ev)
            self.key = key
        elif self._built is not None and self._built[0] != cur.cuda_stream and not torch.cuda.is_current_stream_capturing():
            # (a capture is always preceded by uncaptured warm-up calls on the capturing stream: ordered there)
            if self._built[1].query():
                self._built = None               # long done: nothing to order any more
            else:
                cur.wait_event(self._built[1])
        return self.buf

    def fold(self, codebook, conv):
        """the image of `codebook` with the 1x1 `conv` folded in (dvq_fold_prepare_f32: E W, seeds, bound constants), built once
        per (codebook, conv weight) pair; rebuilt into a fresh buffer when either changes.  Inference only."""
        from . import qconv as _qconv
        pbuf = self.get(codebook)
        w, bias = conv.weight, conv.bias
        K, D = codebook.shape
        key = (self.key, w.data_ptr(), w._version, None if bias is None else (bias.data_ptr(), bias._version))
        cur = torch.cuda.current_stream(codebook.device)
        ent = self._fold.get(id(conv))
        if ent is None or ent[0] != key or conv.training:
            nbytes = _lib_handle.dvq_fold_prep_bytes(K, D)
            if nbytes == 0:
                raise _lib.DvqError("fold: unsupported codebook shape K=%d D=%d" % (K, D))
            if ent is not None:
                self._retired = (self._retired + [ent[1]])[-4:]
            if len(self._fold) >= 4:
                self._fold.clear()
            buf = torch.empty(nbytes, dtype=torch.uint8, device=codebook.device)
            w2 = _lib.require_cuda_f32(w.detach().reshape(D, D), "quant_conv.weight")
            b2 = None if bias is None else _lib.require_cuda_f32(bias.detach(), "quant_conv.bias")
            _lib.check(_lib_handle.dvq_fold_prepare_f32(codebook.data_ptr(), K, D, pbuf.data_ptr(), w2.data_ptr(), _lib.ptr(b2),
                                                        buf.data_ptr(), buf.numel(), cur.cuda_stream), "dvq_fold_prepare_f32")
            ev = torch.cuda.Event()
            ev.record(cur)
            ent = (key, buf, [cur.cuda_stream, ev])
            self._fold[id(conv)] = ent
        elif ent[2] is not None and ent[2][0] != cur.cuda_stream and not torch.cuda.is_current_stream_capturing():
            if ent[2][1].query():
                self._fold[id(conv)] = (ent[0], ent[1], None)
            else:
                cur.wait_event(ent[2][1])
        return pbuf, ent[1]

    def padded_codebook(self, codebook, Dp):
        """[K, Dp] copy of `codebook` with zero channels appended (widths served by padding, _padded_width), rebuilt when the
        codebook tensor changes; `get()` then prepares THAT tensor"""
        key = (codebook.data_ptr(), codebook._version, tuple(codebook.shape), Dp)
        ent = getattr(self, "_padded", None)
        if ent is None or ent[0] != key:
            Ep = codebook.new_zeros((codebook.shape[0], Dp))
            Ep[:, :codebook.shape[1]] = codebook.detach()
            ent = self._padded = (key, Ep)
        return ent[1]

    def used(self, device):
        """called after an op that read the image was queued on the current stream (only needed while the codebook can
        still change: training-mode quantizers)"""
        if torch.cuda.is_current_stream_capturing():
            return
        cur = torch.cuda.current_stream(device)
        ev = self._users.get(cur.cuda_stream)
        if ev is None:
            if len(self._users) >= 16:
                self._users.clear()
            ev = self._users[cur.cuda_stream] = torch.cuda.Event()
        ev.record(cur)

    def workspace(self, B, D, HW, K, mode, device, nbytes=None):
        key = (B, D, HW, K, mode, device, _lib.stream_ptr(device))
        ws = self._ws.get(key)
        if ws is None:
            if nbytes is None and isinstance(HW, tuple):        # routed workspace: ("routed2" | "routed3", hc, wc)
                nbytes = _lib_handle.dvq_vq_assign_routed_workspace_bytes(int(HW[0][-1]), B, D, HW[1], HW[2], K, mode)
                if nbytes == 0:
                    raise _lib.DvqError("routed assign: unsupported shape B=%d D=%d hc=%d wc=%d K=%d" % (B, D, HW[1], HW[2], K))
            if nbytes is None:
                nbytes = _lib_handle.dvq_vq_assign_workspace_bytes(B, D, HW, K, mode)
            if len(self._ws) >= 8:                   # shapes rarely change: keep the table small
                self._ws.clear()
            ws = _Workspace(nbytes, device)
            self._ws[key] = ws
        self._last_ws = (key, ws)
        return ws

    def fallback_count(self):
        """(tokens queued for the resolver, tokens sent to the full exact pass) of the last
        filter-mode call through this object (syncs)"""
        if self._last_ws is None:
            return (0, 0)
        (B, D, HW, K, mode, _dev, _st), ws = self._last_ws
        if mode not in _lib.FILTER_MODES:
            return (0, 0)
        if isinstance(HW, tuple):                   # routed workspace: ("routed2" | "routed3", hc, wc)
            off = _lib_handle.dvq_vq_assign_routed_fallback_count_offset(int(HW[0][-1]), B, D, HW[1], HW[2], K)
        else:
            off = _lib_handle.dvq_vq_assign_fallback_count_offset(B, D, HW, K)
        c = ws[off:off + 8].view(torch.int32).tolist()
        return (int(c[0]), int(c[1]))


def _conv_args(conv, prep, shape, device, h_buf):
    """(prepared conv buffer, h_buf or None, h_all) for the ops with the 1x1 quant_conv fused in.  h_buf [B, D, *spatial] (tests,
    the bench's parity check): the op also writes there the conv output it scored, for EVERY token.  Without one the op needs no
    scratch at all (round 6: the exact-list kernel computes the conv output of its few tokens itself; through round 5 a full-size
    scratch tensor per stream -- 256 MiB at B = 256 -- received their rows)."""
    from . import qconv as _qconv
    if not _qconv.usable(conv) or conv.in_channels != 256:
        raise _lib.DvqError("fused quant_conv: needs a 1x1 nn.Conv2d(256, 256) on the GPU (other sizes: quant_conv + vq_assign)")
    h_all = h_buf is not None
    if h_all and (tuple(h_buf.shape) != tuple(shape) or h_buf.dtype != torch.float32 or not h_buf.is_contiguous() or h_buf.device != device):
        raise ValueError("h_buf must be a contiguous f32 tensor %s on %s" % (tuple(shape), device))
    return _qconv._prep_of(conv).get(conv), h_buf, h_all


def _fold_args(conv, prep, codebook, want_loss, mode):
    """(qconv prep, codebook prep, fold prep) for the ops with the quant_conv FOLDED into the codebook (opt-in `fold=True`)"""
    from . import qconv as _qconv
    if conv is None or not _qconv.usable(conv) or conv.in_channels != codebook.shape[1]:
        raise _lib.DvqError("fold=True needs conv = a 1x1 nn.Conv2d(D, D) on the GPU, D = the codebook dim (64, 128, 256)")
    if want_loss:
        raise ValueError("fold=True computes no loss (the conv's output does not exist for decided tokens): pass want_loss=False")
    if mode not in (_lib.MODE_FILTER, _lib.MODE_FILTER_PASS1):
        raise ValueError("fold=True is a form of the filter path (mode=MODE_FILTER)")
    qbuf = _qconv._prep_of(conv).get(conv)
    pbuf, fbuf = prep.fold(codebook, conv)
    return qbuf, pbuf, fbuf


_TORCH_RANDPERM = torch.randperm        # (tests and goldens pin the dead-code restart by replacing torch.randperm)
_RESTART_GEN = None                     # (torch.initial_seed() it was made for, torch.Generator): _restart_pick's own stream of seeds


def _restart_pick(n, k, device):
    """the first k entries of a uniform random permutation of range(n) -- what the reference's dead-code restart takes from
    `torch.randperm(n_vectors)` (quantize2_mask.py:93-96) -- without permuting all n: drawing indices independently and keeping
    first occurrences IS sequential sampling without replacement, i.e. the same distribution (`dvq_restart_pick_i64`: one small
    workgroup, 2k counter-based draws, an LDS hash table; fewer than k distinct values among them does not happen in practice for n >= 16 k;
    a slot that stayed empty would keep its own index i, which may repeat a chosen one -- `distinct` is a practical, not a formal property).  The full
    permutation is a 262 144-key device sort, ~115 us of the 1.04-ms training step at B = 256 (rocprofv3: profiles/
    r05_train_step.json).  Small batches, CPU tensors and a caller that replaced torch.randperm (the tests pin the permutation
    that way) take torch.randperm itself."""
    if torch.randperm is not _TORCH_RANDPERM or n < 16 * k or k > 2048 or device.type != "cuda":
        return torch.randperm(n, device=device)[:k]
    global _RESTART_GEN
    if _RESTART_GEN is None or _RESTART_GEN[0] != torch.initial_seed():
        # a generator of its own, seeded from torch.manual_seed's value: the pick is reproducible under torch.manual_seed and
        # does not advance the global CPU generator (data-loader shuffles after it draw what they drew without it -- ADVICE r5)
        g = torch.Generator()
        g.manual_seed((torch.initial_seed() ^ 0x5DEECE66D) & 0x7FFFFFFFFFFFFFFF)
        _RESTART_GEN = (torch.initial_seed(), g)
    seed = int(torch.empty((), dtype=torch.int64).random_(generator=_RESTART_GEN[1]).item()) & 0x7FFFFFFFFFFFFFFF
    out = torch.empty(k, dtype=torch.int64, device=device)
    with _lib.on_device(device):
        _lib.check(_lib_handle.dvq_restart_pick_i64(seed, n, k, out.data_ptr(), _lib.stream_ptr(device)), "dvq_restart_pick_i64")
    return out


KERNEL_WIDTHS = (64, 128, 256)          # channel counts the assign kernels are instantiated for


def _padded_width(D):
    """the kernel width a codebook of D channels runs at: D itself, or -- D a multiple of 32 below 256 -- the next kernel width
    with ZERO channels appended to latents and codebook.  That is exact, not approximate: a zero channel adds fma(0, 0, acc) =
    acc to the sequential dot chain and + 0 to a partial sum of the norm whose 32-way grouping does not depend on D, so distances,
    codes and z_q are the reference's bit for bit (pinned for D = 32, 96, 160, 192, 224 by oracle/validate_against_reference.py
    and tests/test_gpu_parity.py).  Other widths (not a multiple of 32: ATen's reduction order for the tail is not pinned; above
    256: no kernel) raise."""
    if D in KERNEL_WIDTHS:
        return D
    if D > 0 and D % 32 == 0 and D < KERNEL_WIDTHS[-1]:
        return min(w for w in KERNEL_WIDTHS if w >= D)
    raise _lib.DvqError("codebook_dim %d: the MI355X kernels serve 64, 128 and 256 channels, and 32, 96, 160, 192, 224 by exact "
                        "zero padding; other widths are not supported (INTEGRATION.md)" % D)


def _vq_assign_padded(z, codebook, prep, mask, beta, want_zq, want_loss, mode, out, Dp):
    """vq_assign for a width the kernels are not instantiated for: run at Dp >= D with zero channels (see _padded_width)"""
    B, D = z.shape[0], z.shape[1]
    zp = z.new_zeros((B, Dp) + tuple(z.shape[2:]))
    zp[:, :D] = z
    Ep = prep.padded_codebook(codebook, Dp)
    zq_p, codes, loss_p = vq_assign(zp, Ep, prep, mask, beta, want_zq, want_loss, mode)
    zq = zq_p[:, :D].contiguous() if want_zq else None
    loss = None
    if loss_p is not None:
        mean = loss_p[0] * (float(Dp) / float(D))          # the kernel divided by N * Dp; the padding's squared error is exactly 0
        loss = torch.stack([mean, beta * mean + mean])
    if out is not None:
        for dst, src in zip(out, (zq, codes, loss)):
            if dst is not None and src is not None:
                dst.copy_(src)
        return out
    return zq, codes, loss


def vq_assign(z, codebook, prep, mask=None, beta=0.25, want_zq=True, want_loss=True,
              mode=_lib.MODE_FILTER, out=None, conv=None, h_buf=None, fold=False):
    """z [B, D, *spatial] f32 cuda, codebook [K, D] -> (zq or None, codes [B, *spatial] i64, loss[2] or None).

    loss[0] = mean((e - z)^2 * mask), loss[1] = beta*mean + mean.  `out` may carry preallocated
    (zq, codes, loss) tensors (used by the benchmark / graph capture).
    conv: a 1x1 nn.Conv2d(256, 256) applied to z first INSIDE the assign kernel (`dvq_vq_assign_qconv_f32`: the model's
    quant_conv; its output never reaches memory).  h_buf: see _conv_args.
    fold=True (needs conv, want_loss=False): the conv FOLDED into the codebook (`dvq_vq_assign_fold_f32`; loss-free inference /
    stage-2 tokenisation): pass 1 scores z against E W and computes no conv; only undecided tokens get their conv output and the
    reference chain.  codes = the reference argmin for a conv output within 1e-5 sum |w||x| of the real-number conv (the same
    contract as conv= alone: near-ties of that tolerance may resolve differently); z_q = codebook[code] (within 1e-6 relative of
    fl(h + fl(e - h)))."""
    z = _lib.require_cuda_f32(z, "z")
    codebook = _lib.require_cuda_f32(codebook, "codebook")
    B, D = z.shape[0], z.shape[1]
    HW = 1
    for s in z.shape[2:]:
        HW *= s
    K = codebook.shape[0]
    if codebook.shape[1] != D:
        raise ValueError("codebook dim %d != feature dim %d" % (codebook.shape[1], D))
    if mask is not None:
        mask = _lib.require_cuda_f32(mask, "codebook_mask")
        if mask.numel() != B * HW:
            raise ValueError("codebook_mask has %d elements, expected B*H*W = %d" % (mask.numel(), B * HW))
    Dp = _padded_width(D)
    if Dp != D and B * HW > 0:
        if conv is not None or fold:
            raise _lib.DvqError("the fused / folded quant_conv needs a kernel width (64, 128, 256 channels), got %d" % D)
        return _vq_assign_padded(z, codebook, prep, mask, beta, want_zq, want_loss, mode, out, Dp)
    if out is not None:
        zq, codes, loss = out
    else:
        zq = torch.empty_like(z) if want_zq else None
        codes = torch.empty((B,) + tuple(z.shape[2:]), dtype=torch.int64, device=z.device)
        loss = torch.empty(2, dtype=torch.float32, device=z.device) if want_loss else None
    if B * HW == 0:
        # empty batch: what the reference's torch ops give (empty codes / z_q, mean of nothing = NaN)
        if loss is not None:
            loss.fill_(float("nan"))
        return zq, codes, loss
    ws = prep.workspace(B, D, HW, K, mode, z.device)
    if fold:
        with _lib.on_device(z.device):
            qbuf, pbuf, fbuf = _fold_args(conv, prep, codebook, loss is not None, mode)
            ws.check(mode, _lib_handle.dvq_vq_assign_fold_f32(
                z.data_ptr(), qbuf.data_ptr(), fbuf.data_ptr(), codebook.data_ptr(), pbuf.data_ptr(), B, D, HW, K,
                _lib.ptr(zq), codes.data_ptr(), *ws.begin(mode), _lib.stream_ptr(z.device)), "dvq_vq_assign_fold_f32")
        return zq, codes, loss
    if conv is not None:
        with _lib.on_device(z.device):
            qbuf, hb, h_all = _conv_args(conv, prep, z.shape, z.device, h_buf)
            pbuf = prep.get(codebook)
            ws.check(mode, _lib_handle.dvq_vq_assign_qconv_f32(
                z.data_ptr(), qbuf.data_ptr(), codebook.data_ptr(), pbuf.data_ptr(), _lib.ptr(mask), B, D, HW, K, float(beta),
                _lib.ptr(zq), codes.data_ptr(), _lib.ptr(loss), _lib.ptr(hb), int(h_all), *ws.begin(mode),
                _lib.stream_ptr(z.device)), "dvq_vq_assign_qconv_f32")
        return zq, codes, loss
    if HW == 1:
        # row-major [N, D] (channel_last inputs, VectorQuantize2List's concatenated rows, VQEmbedding.forward): the entry point
        # whose pass 1 reads / writes a token's row with 16-byte accesses
        with _lib.on_device(z.device):
            pbuf = prep.get(codebook)
            ws.check(mode, _lib_handle.dvq_vq_assign_flat_f32(
                z.data_ptr(), codebook.data_ptr(), pbuf.data_ptr(), _lib.ptr(mask), B, D, K, float(beta),
                _lib.ptr(zq), codes.data_ptr(), _lib.ptr(loss), *ws.begin(mode), _lib.stream_ptr(z.device)), "dvq_vq_assign_flat_f32")
        return zq, codes, loss
    with _lib.on_device(z.device):
        pbuf = prep.get(codebook)
        ws.check(mode, _lib_handle.dvq_vq_assign_nchw_f32(
            z.data_ptr(), codebook.data_ptr(), pbuf.data_ptr(), _lib.ptr(mask), B, D, HW, K, float(beta),
            _lib.ptr(zq), codes.data_ptr(), _lib.ptr(loss), *ws.begin(mode),
            _lib.stream_ptr(z.device)), "dvq_vq_assign_nchw_f32")
    return zq, codes, loss


def _gate_for_routed(gate, G):
    """-> (tensor, gate_kind) for a router output [B, hc, wc, G] (f32 logits or int64)"""
    if not gate.is_cuda:
        raise _lib.DvqError("gate is on %s: the dvq kernels run on the GPU only" % gate.device)
    if gate.dim() != 4 or gate.shape[-1] != G:
        raise ValueError("gate must be [B, h, w, %d], got %s" % (G, tuple(gate.shape)))
    if gate.dtype == torch.int64:
        return gate.contiguous(), _lib.GATE_I64
    return (gate if gate.dtype == torch.float32 else gate.float()).contiguous(), _lib.GATE_F32


def _routed_outputs(h_fine, B, hc, wc, S, want_zq, want_loss, with_gate):
    dev = h_fine.device
    zq = torch.empty_like(h_fine) if want_zq else None
    codes = torch.empty((B, S * hc, S * wc), dtype=torch.int64, device=dev)
    loss = torch.empty(2, dtype=torch.float32, device=dev) if want_loss else None
    indices = torch.empty((B, hc, wc), dtype=torch.int64, device=dev)
    cmask = torch.empty((B, 1, S * hc, S * wc), dtype=torch.float32, device=dev)
    gate_out = torch.empty((B, hc, wc, 2), dtype=torch.int64, device=dev) if with_gate else None
    return zq, codes, loss, indices, cmask, gate_out


def vq_assign_routed_dual(h_coarse, h_fine, codebook, prep, gate=None, entropy=None, threshold=None, beta=0.25,
                          want_zq=True, want_loss=True, mode=_lib.MODE_FILTER, out=None, conv=None, h_buf=None, fold=False):
    """Routing tail of DualGrainEncoder (EncoderDual.py:134-149) + VectorQuantize2.forward
    (quantize2_mask.py:157-191) as ONE op on the unique tokens (`dvq_vq_assign_routed_dual_f32`):
    h_coarse [B, D, hc, wc], h_fine [B, D, 2hc, 2wc]; either `gate` [B, hc, wc, 2] (f32 logits / int64) or
    `entropy` [B, hc, wc] + `threshold` (the fixed-entropy router fused in).
    -> dict(zq, codes [B, 2hc, 2wc] i64, loss[2], indices [B, hc, wc] i64, codebook_mask [B, 1, 2hc, 2wc],
            gate (entropy form: the router's int64 gate [B, hc, wc, 2], else the input gate)).
    `out` = (zq, codes, loss, indices, codebook_mask, gate_out) preallocated (benchmark / graph capture).
    conv: the model's 1x1 quant_conv between select and quantizer, fused in (`dvq_vq_assign_routed_qconv_dual_f32`): the
    order every reference checkpoint runs (dqvae_dual_entropy.py:124-134), one kernel chain, h never written."""
    h_coarse = _lib.require_cuda_f32(h_coarse, "h_coarse")
    h_fine = _lib.require_cuda_f32(h_fine, "h_fine")
    codebook = _lib.require_cuda_f32(codebook, "codebook")
    B, D, hc, wc = h_coarse.shape
    K = codebook.shape[0]
    if tuple(h_fine.shape) != (B, D, 2 * hc, 2 * wc) or codebook.shape[1] != D:
        raise ValueError("shape mismatch: h_coarse %s h_fine %s codebook %s" %
                         (tuple(h_coarse.shape), tuple(h_fine.shape), tuple(codebook.shape)))
    if (gate is None) == (entropy is None):
        raise ValueError("give exactly one of gate / entropy")
    if entropy is not None:
        g = _lib.require_cuda_f32(entropy, "entropy")
        if tuple(g.shape) != (B, hc, wc) or threshold is None:
            raise ValueError("entropy must be [B, hc, wc] and come with a threshold")
        kind, thr = _lib.GATE_ENTROPY, float(threshold)
    else:
        g, kind = _gate_for_routed(gate, 2)
        if tuple(g.shape[:3]) != (B, hc, wc):
            raise ValueError("gate %s does not match h_coarse %s" % (tuple(gate.shape), tuple(h_coarse.shape)))
        thr = 0.0
    if out is not None:
        zq, codes, loss, indices, cmask, gate_out = out
    else:
        zq, codes, loss, indices, cmask, gate_out = _routed_outputs(h_fine, B, hc, wc, 2, want_zq, want_loss,
                                                                    entropy is not None)
    res = {"zq": zq, "codes": codes, "loss": loss, "indices": indices, "codebook_mask": cmask,
           "gate": gate_out if entropy is not None else gate}
    if B * hc * wc == 0:
        if loss is not None:
            loss.fill_(float("nan"))
        return res
    ws = prep.workspace(B, D, ("routed2", hc, wc), K, mode, h_fine.device)
    if fold:                                     # the conv folded into the codebook (see vq_assign): codes [+ z_q], no loss
        with _lib.on_device(h_fine.device):
            qbuf, pbuf, fbuf = _fold_args(conv, prep, codebook, loss is not None, mode)
            ws.check(mode, _lib_handle.dvq_vq_assign_routed_fold_dual_f32(
                g.data_ptr(), kind, thr, h_coarse.data_ptr(), h_fine.data_ptr(), qbuf.data_ptr(), fbuf.data_ptr(),
                codebook.data_ptr(), pbuf.data_ptr(), B, D, hc, wc, K, _lib.ptr(zq), codes.data_ptr(), indices.data_ptr(),
                cmask.data_ptr(), _lib.ptr(gate_out), *ws.begin(mode), _lib.stream_ptr(h_fine.device)),
                "dvq_vq_assign_routed_fold_dual_f32")
        return res
    if conv is not None:
        with _lib.on_device(h_fine.device):
            qbuf, hb, h_all = _conv_args(conv, prep, h_fine.shape, h_fine.device, h_buf)
            pbuf = prep.get(codebook)
            ws.check(mode, _lib_handle.dvq_vq_assign_routed_qconv_dual_f32(
                g.data_ptr(), kind, thr, h_coarse.data_ptr(), h_fine.data_ptr(), qbuf.data_ptr(), codebook.data_ptr(),
                pbuf.data_ptr(), B, D, hc, wc, K, float(beta), _lib.ptr(zq), codes.data_ptr(), _lib.ptr(loss),
                indices.data_ptr(), cmask.data_ptr(), _lib.ptr(gate_out), _lib.ptr(hb), int(h_all), *ws.begin(mode),
                _lib.stream_ptr(h_fine.device)), "dvq_vq_assign_routed_qconv_dual_f32")
        return res
    with _lib.on_device(h_fine.device):
        pbuf = prep.get(codebook)
        ws.check(mode, _lib_handle.dvq_vq_assign_routed_dual_f32(
            g.data_ptr(), kind, thr, h_coarse.data_ptr(), h_fine.data_ptr(), codebook.data_ptr(), pbuf.data_ptr(),
            B, D, hc, wc, K, float(beta), _lib.ptr(zq), codes.data_ptr(), _lib.ptr(loss), indices.data_ptr(),
            cmask.data_ptr(), _lib.ptr(gate_out), *ws.begin(mode), _lib.stream_ptr(h_fine.device)),
            "dvq_vq_assign_routed_dual_f32")
    return res


def vq_assign_routed_triple(h_coarse, h_median, h_fine, codebook, prep, gate, beta=0.25, want_zq=True,
                            want_loss=True, mode=_lib.MODE_FILTER, out=None, conv=None, h_buf=None, fold=False):
    """Routing tail of TripleGrainEncoder (EncoderTriple.py:148-176) + VectorQuantize2.forward as ONE op on the
    unique tokens (`dvq_vq_assign_routed_triple_f32`): h_coarse [B, D, hc, wc], h_median [B, D, 2hc, 2wc],
    h_fine [B, D, 4hc, 4wc], gate [B, hc, wc, 3].  -> dict as vq_assign_routed_dual.
    `out` = (zq, codes, loss, indices, codebook_mask)."""
    h_coarse = _lib.require_cuda_f32(h_coarse, "h_coarse")
    h_median = _lib.require_cuda_f32(h_median, "h_median")
    h_fine = _lib.require_cuda_f32(h_fine, "h_fine")
    codebook = _lib.require_cuda_f32(codebook, "codebook")
    B, D, hc, wc = h_coarse.shape
    K = codebook.shape[0]
    if (tuple(h_median.shape) != (B, D, 2 * hc, 2 * wc) or tuple(h_fine.shape) != (B, D, 4 * hc, 4 * wc)
            or codebook.shape[1] != D):
        raise ValueError("shape mismatch: h_coarse %s h_median %s h_fine %s codebook %s" %
                         (tuple(h_coarse.shape), tuple(h_median.shape), tuple(h_fine.shape), tuple(codebook.shape)))
    g, kind = _gate_for_routed(gate, 3)
    if tuple(g.shape[:3]) != (B, hc, wc):
        raise ValueError("gate %s does not match h_coarse %s" % (tuple(gate.shape), tuple(h_coarse.shape)))
    if out is not None:
        zq, codes, loss, indices, cmask = out
    else:
        zq, codes, loss, indices, cmask, _ = _routed_outputs(h_fine, B, hc, wc, 4, want_zq, want_loss, False)
    res = {"zq": zq, "codes": codes, "loss": loss, "indices": indices, "codebook_mask": cmask, "gate": gate}
    if B * hc * wc == 0:
        if loss is not None:
            loss.fill_(float("nan"))
        return res
    ws = prep.workspace(B, D, ("routed3", hc, wc), K, mode, h_fine.device)
    if fold:
        with _lib.on_device(h_fine.device):
            qbuf, pbuf, fbuf = _fold_args(conv, prep, codebook, loss is not None, mode)
            ws.check(mode, _lib_handle.dvq_vq_assign_routed_fold_triple_f32(
                g.data_ptr(), kind, h_coarse.data_ptr(), h_median.data_ptr(), h_fine.data_ptr(), qbuf.data_ptr(),
                fbuf.data_ptr(), codebook.data_ptr(), pbuf.data_ptr(), B, D, hc, wc, K, _lib.ptr(zq), codes.data_ptr(),
                indices.data_ptr(), cmask.data_ptr(), *ws.begin(mode), _lib.stream_ptr(h_fine.device)),
                "dvq_vq_assign_routed_fold_triple_f32")
        return res
    if conv is not None:
        with _lib.on_device(h_fine.device):
            qbuf, hb, h_all = _conv_args(conv, prep, h_fine.shape, h_fine.device, h_buf)
            pbuf = prep.get(codebook)
            ws.check(mode, _lib_handle.dvq_vq_assign_routed_qconv_triple_f32(
                g.data_ptr(), kind, h_coarse.data_ptr(), h_median.data_ptr(), h_fine.data_ptr(), qbuf.data_ptr(),
                codebook.data_ptr(), pbuf.data_ptr(), B, D, hc, wc, K, float(beta), _lib.ptr(zq), codes.data_ptr(),
                _lib.ptr(loss), indices.data_ptr(), cmask.data_ptr(), _lib.ptr(hb), int(h_all), *ws.begin(mode),
                _lib.stream_ptr(h_fine.device)), "dvq_vq_assign_routed_qconv_triple_f32")
        return res
    with _lib.on_device(h_fine.device):
        pbuf = prep.get(codebook)
        ws.check(mode, _lib_handle.dvq_vq_assign_routed_triple_f32(
            g.data_ptr(), kind, h_coarse.data_ptr(), h_median.data_ptr(), h_fine.data_ptr(), codebook.data_ptr(),
            pbuf.data_ptr(), B, D, hc, wc, K, float(beta), _lib.ptr(zq), codes.data_ptr(), _lib.ptr(loss),
            indices.data_ptr(), cmask.data_ptr(), *ws.begin(mode), _lib.stream_ptr(h_fine.device)),
            "dvq_vq_assign_routed_triple_f32")
    return res


def embed_gather(codebook, idx):
    """codebook[idx] (nn.Embedding forward) through dvq_embed_gather_f32."""
    codebook = _lib.require_cuda_f32(codebook, "codebook")
    if idx.dtype != torch.int64:
        idx = idx.long()
    idx = idx.contiguous()
    K, D = codebook.shape
    out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=codebook.device)
    with _lib.on_device(codebook.device):
        _lib.check(_lib_handle.dvq_embed_gather_f32(codebook.data_ptr(), K, D, idx.data_ptr(), idx.numel(),
                                                    out.data_ptr(), _lib.stream_ptr(codebook.device)),
                   "dvq_embed_gather_f32")
    return out


class _VQStraightThrough(torch.autograd.Function):
    """Forward = the fused kernel.  Backward = what autograd derives from the reference graph
    (quantize2_mask.py:172-182 / quantize_vqgan.py:290-298): identity through z + (z_q - z).detach(),
    plus the commitment-loss gradients 2 c (z - e) m / numel on z (c = coefficient of the
    (z_q.detach() - z)^2 term) and 2 c' (e - z) m / numel scattered onto the codebook rows.

    The live codebook is NOT what backward reads: like the reference (F.embedding keeps only the indices and the rows it read)
    the rows chosen at forward time are what the gradient uses, so the EMA update may overwrite the weight in place between
    forward and backward -- a snapshot of the [K, D] codebook (1 MiB) is saved, not the [B, HW, D] gathered rows.
    d/dz runs as ONE kernel (`dvq_vq_backward_nchw_f32`: read z and g_zq, gather e from the snapshot, write g_z) in the same fp32
    operation order as the torch expression below, which remains the path for CPU tensors / odd channel counts."""

    @staticmethod
    def forward(ctx, z, weight, mask, prep, K, coef_z, coef_e, mode):
        codebook = weight[:K]
        # beta*mean + mean is the same fp32 number whichever addend carries beta (legacy or not)
        zq, codes, loss = vq_assign(z, codebook, prep, mask, beta=(coef_z if coef_e == 1.0 else coef_e), mode=mode)
        if prep.track_users:
            prep.used(z.device)
        need_z, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # forward-time codebook: a [K, D] snapshot -- or, when the call has fewer tokens than the codebook has codes and no
        # codebook gradient is wanted (the items of VectorQuantize2List: one snapshot per item would be 16 MiB each at K = 16384),
        # just the rows the tokens chose, addressed in backward by position
        ctx.compact = bool((need_z and not need_w) and codes.numel() < K and z.is_cuda)
        if ctx.compact:
            ctx.save_for_backward(z, embed_gather(codebook.detach(), codes.reshape(-1)), mask, codes)
        elif need_z or need_w:
            ctx.save_for_backward(z, codebook.detach().clone(), mask, codes)
        ctx.wshape, ctx.coef_z, ctx.coef_e = tuple(weight.shape), coef_z, coef_e
        ctx.mark_non_differentiable(codes)
        return zq, loss[1], codes

    @staticmethod
    def backward(ctx, g_zq, g_loss, _g_codes):
        z, snap, mask, codes = ctx.saved_tensors
        if ctx.compact:                                   # snap = the chosen rows [N, D]: token n's row is row n
            codes = torch.arange(codes.numel(), dtype=torch.int64, device=codes.device).reshape(codes.shape)
        B, D = z.shape[0], z.shape[1]
        need_z, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        scale = 2.0 / z.numel()
        gz, gw = None, None
        fused = (need_z and z.is_cuda and z.dtype == torch.float32 and D % 16 == 0 and z.is_contiguous()
                 and (g_zq is None or (g_zq.dtype == torch.float32 and g_zq.is_cuda)))
        if need_z and g_loss is None:
            gz = g_zq
        elif fused:
            HW = z[0, 0].numel()
            gq = None if g_zq is None else g_zq.contiguous()
            gl = g_loss.reshape(1).to(torch.float32).contiguous()
            m = None if mask is None else _lib.require_cuda_f32(mask, "codebook_mask")
            gz = torch.empty_like(z)
            with _lib.on_device(z.device):
                _lib.check(_lib_handle.dvq_vq_backward_nchw_f32(
                    z.data_ptr(), snap.data_ptr(), codes.data_ptr(), _lib.ptr(m), _lib.ptr(gq), gl.data_ptr(),
                    float(ctx.coef_z * scale), B, D, HW, snap.shape[0], gz.data_ptr(), _lib.stream_ptr(z.device)),
                    "dvq_vq_backward_nchw_f32")
        if need_w and g_loss is not None and z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and snap.shape[0] <= 8192:
            HW = z[0, 0].numel()
            gw = torch.zeros(ctx.wshape, dtype=z.dtype, device=z.device)
            gl = g_loss.reshape(1).to(torch.float32).contiguous()
            m = None if mask is None else _lib.require_cuda_f32(mask, "codebook_mask")
            with _lib.on_device(z.device):
                _lib.check(_lib_handle.dvq_vq_backward_codebook_nchw_f32(
                    z.data_ptr(), snap.data_ptr(), codes.data_ptr(), _lib.ptr(m), gl.data_ptr(), float(ctx.coef_e * scale),
                    B, D, HW, snap.shape[0], gw.data_ptr(), _lib.stream_ptr(z.device)), "dvq_vq_backward_codebook_nchw_f32")
            need_w = False                                # done
        diff = None
        if (need_z and gz is None) or (need_w and g_loss is not None):
            e = embed_gather(snap, codes.reshape(B, -1)) if z.is_cuda else snap[codes.reshape(B, -1)]   # [B, HW, D]
            diff = z - e.permute(0, 2, 1).reshape(z.shape)
            if mask is not None:
                diff = diff * mask.reshape(B, 1, *z.shape[2:])
        if need_z and gz is None:
            gz = (g_zq if g_zq is not None else 0) + g_loss * (ctx.coef_z * scale) * diff
        if need_w or (gw is None and ctx.needs_input_grad[1]):
            gw = torch.zeros(ctx.wshape, dtype=z.dtype, device=z.device)
            if g_loss is not None and need_w:
                ge = (-(g_loss * (ctx.coef_e * scale)) * diff).reshape(B, D, -1).permute(0, 2, 1).reshape(-1, D)
                gw.index_add_(0, codes.reshape(-1), ge)
        return gz, gw, None, None, None, None, None, None


def _vq_straight_through(z, weight, mask, prep, K, coef_z, coef_e, mode):
    """_VQStraightThrough.apply when a graph is being recorded; the bare op otherwise.  (needs_input_grad inside a Function mirrors
    requires_grad of the inputs whether or not autograd is recording: under no_grad -- every inference call of the drop-in modules,
    whose weight is a Parameter -- the Function would still snapshot the codebook: a 1-MiB copy kernel and ~10 us of host time per
    call for a backward nobody can ask for.)"""
    if torch.is_grad_enabled() and (z.requires_grad or weight.requires_grad):
        return _VQStraightThrough.apply(z, weight, mask, prep, K, coef_z, coef_e, mode)
    zq, codes, loss = vq_assign(z, weight[:K], prep, mask, beta=(coef_z if coef_e == 1.0 else coef_e), mode=mode)
    if prep.track_users:
        prep.used(z.device)
    return zq, loss[1], codes


class VQEmbedding(nn.Embedding):
    """EMA codebook; same parameters/buffers as the reference class (quantize2_mask.py:10-132):
    weight [K+1, D] (row K is the padding row), cluster_size_ema [K], embed_ema [K, D]."""

    def __init__(self, n_embed, embed_dim, ema=True, decay=0.99, restart_unused_codes=True, eps=1e-5):
        super().__init__(n_embed + 1, embed_dim, padding_idx=n_embed)
        self.ema = ema
        self.decay = decay
        self.eps = eps
        self.restart_unused_codes = restart_unused_codes
        self.n_embed = n_embed
        if self.ema:
            _ = [p.requires_grad_(False) for p in self.parameters()]
            self.register_buffer('cluster_size_ema', torch.zeros(n_embed))
            self.register_buffer('embed_ema', self.weight[:-1, :].detach().clone())
        self._prep = _CodebookPrep()

    @property
    def codes(self):
        """the K live codebook rows (weight[:-1], quantize2_mask.py:31)"""
        return self.weight[:-1, :]

    def invalidate_codebook_cache(self):
        """call after writing the codebook through `.data` (which does not bump the version counter)"""
        self._prep.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._prep.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._prep.invalidate()
        return out

    @torch.no_grad()
    def compute_distances(self, inputs):
        """Dense squared distances [..., K] of channel-last inputs [..., D] to the K live rows
        (quantize2_mask.py:29-48).  Only get_soft_codes needs the full matrix; it is a vendor GEMM
        (torch / hipBLASLt) at tolerance level -- the assignment itself never builds it."""
        rows = self.weight[:-1, :]
        if inputs.shape[-1] != rows.shape[1]:
            raise ValueError("last dim %d != codebook dim %d" % (inputs.shape[-1], rows.shape[1]))
        x = inputs.reshape(-1, rows.shape[1])
        norms = (x * x).sum(1, keepdim=True) + (rows * rows).sum(1).unsqueeze(0)
        return torch.addmm(norms, x, rows.t(), alpha=-2.0).reshape(*inputs.shape[:-1], rows.shape[0])

    @torch.no_grad()
    def find_nearest_embedding(self, inputs):
        """inputs [..., D] channel-last -> indices [...] (quantize2_mask.py:50-55), by the kernel."""
        flat = inputs.reshape(-1, inputs.shape[-1])
        _, codes, _ = vq_assign(flat.contiguous(), self.codes, self._prep, want_zq=False, want_loss=False)
        return codes.reshape(inputs.shape[:-1])

    @torch.no_grad()
    def _tile_with_noise(self, x, target_n):
        """at least target_n rows: x repeated, plus uniform noise of scale 0.01/sqrt(D) per element
        (quantize2_mask.py:57-64)"""
        n, dim = x.shape
        tiled = x.repeat(-(-target_n // n), 1)
        return tiled + torch.rand_like(tiled) * (0.01 / math.sqrt(dim))

    @torch.no_grad()
    def _cluster_sums(self, vectors, idxs, nchw=None):
        """cluster_size [K] and vectors_sum_per_cluster [K, D] (quantize2_mask.py:74-84), returned as
        views of ONE flat [K*D + K] buffer so the data-parallel reduction is a single collective.
        The reference builds a dense one-hot [K, N] matrix and multiplies.  With the NCHW latents at
        hand (`nchw` = z [B, D, *spatial], the layout the quantizer receives) one HIP kernel
        (`dvq_ema_accumulate_nchw_f32`) produces both; otherwise a bincount and an index_add.
        Either way the summation order differs from the reference: tolerance-level parity."""
        n_embed, embed_dim = self.weight.shape[0] - 1, self.weight.shape[-1]
        flat = torch.empty(n_embed * embed_dim + n_embed, dtype=torch.float32, device=vectors.device)
        vsum = flat[:n_embed * embed_dim].view(n_embed, embed_dim)
        cluster_size = flat[n_embed * embed_dim:]
        if nchw is not None and nchw.is_cuda and nchw.dtype == torch.float32:
            z = nchw.contiguous()
            B, HW = z.shape[0], z[0, 0].numel()
            codes = idxs.reshape(B, HW).contiguous()
            with _lib.on_device(z.device):
                _lib.check(_lib_handle.dvq_ema_accumulate_nchw_f32(
                    z.data_ptr(), codes.data_ptr(), B, embed_dim, HW, n_embed, cluster_size.data_ptr(),
                    vsum.data_ptr(), _lib.stream_ptr(z.device)), "dvq_ema_accumulate_nchw_f32")
        else:
            cluster_size.copy_(torch.bincount(idxs, minlength=n_embed))
            vsum.zero_().index_add_(0, idxs, vectors.to(torch.float32))
        return cluster_size, vsum, flat

    @torch.no_grad()
    def _update_buffers(self, vectors, idxs, nchw=None):
        """Training-mode EMA statistics + dead-code restart (quantize2_mask.py:66-105).  Data-parallel:
        the reference's two all_reduce calls (:87-88) are one all_reduce over the flat [K*D + K]
        statistics buffer; the restart vectors are broadcast from rank 0 as in :100."""
        n_embed, embed_dim = self.weight.shape[0] - 1, self.weight.shape[-1]
        idxs = idxs.reshape(-1)
        if nchw is None or not (nchw.is_cuda and nchw.dtype == torch.float32):
            vectors = vectors.reshape(-1, embed_dim)      # token rows (with NCHW latents on the GPU they are never materialised:
            nchw = None                                   # `vectors` stays the permuted VIEW [B, HW, D], rows are picked by index)
        cluster_size, vectors_sum_per_cluster, flat = self._cluster_sums(vectors, idxs, nchw)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        self.cluster_size_ema.mul_(self.decay).add_(cluster_size, alpha=1 - self.decay)
        self.embed_ema.mul_(self.decay).add_(vectors_sum_per_cluster, alpha=1 - self.decay)
        if self.restart_unused_codes:
            self._restart_dead_codes(vectors, self._draw_restart_vectors(vectors))

    @torch.no_grad()
    def _ema_step(self, vectors, idxs, nchw=None):
        """`_update_buffers` + `_update_embedding` (quantize2_mask.py:66-115) for NCHW latents on the GPU: the statistics kernel,
        the (single) all-reduce, the restart pick, then ONE kernel for the two EMA updates, the dead-code restart and the
        normalised weight (`dvq_ema_update_f32`; as torch ops ~20 launches of 4-5 us per step).  Anything else -- CPU tensors,
        fewer input vectors than codes (the reference tiles them with noise), a replaced torch.randperm with too few vectors for
        the pick kernel -- takes the two methods one after the other."""
        n_embed, embed_dim = self.weight.shape[0] - 1, self.weight.shape[-1]
        ok = (nchw is not None and nchw.is_cuda and nchw.dtype == torch.float32 and self.weight.is_cuda
              and self.weight.dtype == torch.float32 and self.weight.is_contiguous()
              and self.embed_ema.is_contiguous() and self.embed_ema.dtype == torch.float32)
        n_vectors = idxs.numel()
        if not ok or (self.restart_unused_codes and n_vectors < n_embed):
            self._update_buffers(vectors, idxs, nchw=nchw)
            self._update_embedding()
            return
        idxs = idxs.reshape(-1)
        cluster_size, vectors_sum, flat = self._cluster_sums(vectors, idxs, nchw)
        ddp = dist.is_available() and dist.is_initialized()
        if ddp:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        z = nchw.contiguous()
        B, HW = z.shape[0], z[0, 0].numel()
        restart, rows, pick = 0, None, None
        if self.restart_unused_codes:
            pick = _restart_pick(n_vectors, n_embed, z.device).contiguous()
            restart = 2
            if ddp:                                          # every rank restarts from rank 0's vectors (quantize2_mask.py:100)
                rows = vectors[torch.div(pick, HW, rounding_mode="floor"), pick % HW].contiguous()
                dist.broadcast(rows, 0)
                restart = 1
        cs_new = torch.empty_like(self.cluster_size_ema)
        with _lib.on_device(z.device):
            _lib.check(_lib_handle.dvq_ema_update_f32(
                vectors_sum.data_ptr(), cluster_size.data_ptr(), float(self.decay), float(self.eps), n_embed, embed_dim,
                self.cluster_size_ema.data_ptr(), cs_new.data_ptr(), self.embed_ema.data_ptr(), self.weight.data_ptr(),
                restart, _lib.ptr(rows), z.data_ptr(), B, HW, _lib.ptr(pick), _lib.stream_ptr(z.device)), "dvq_ema_update_f32")
        self.cluster_size_ema.copy_(cs_new)
        self._prep.invalidate()

    @torch.no_grad()
    def _draw_restart_vectors(self, vectors):
        """n_embed input vectors in random order (tiled with noise if the batch has fewer), the same on
        every rank (quantize2_mask.py:93-100)"""
        n_embed = self.weight.shape[0] - 1
        n_vectors = vectors.numel() // vectors.shape[-1]
        if n_vectors < n_embed:
            vectors = self._tile_with_noise(vectors.reshape(-1, vectors.shape[-1]), n_embed)
            n_vectors = vectors.shape[0]
        pick = _restart_pick(n_vectors, n_embed, vectors.device)
        if vectors.dim() == 3:                            # the permuted view [B, HW, D] of NCHW latents: K rows by (image, position)
            hw = vectors.shape[1]
            chosen = vectors[torch.div(pick, hw, rounding_mode="floor"), pick % hw].contiguous()
        else:
            chosen = vectors[pick]
        if dist.is_available() and dist.is_initialized():
            dist.broadcast(chosen, 0)
        return chosen

    @torch.no_grad()
    def _restart_dead_codes(self, vectors, restart_vectors):
        """codes whose EMA count fell below 1 restart from `restart_vectors` with count 1
        (quantize2_mask.py:102-105)"""
        dead = self.cluster_size_ema < 1
        self.embed_ema.copy_(torch.where(dead.unsqueeze(1), restart_vectors.to(self.embed_ema.dtype), self.embed_ema))
        self.cluster_size_ema.masked_fill_(dead, 1.0)

    @torch.no_grad()
    def _update_embedding(self):
        n_embed = self.weight.shape[0] - 1
        n = self.cluster_size_ema.sum()
        normalized_cluster_size = n * (self.cluster_size_ema + self.eps) / (n + n_embed * self.eps)
        self.weight[:-1, :] = self.embed_ema / normalized_cluster_size.reshape(-1, 1)
        self._prep.invalidate()

    def forward(self, inputs):
        """inputs [..., D] -> (embeds [..., D], idxs [...]) (quantize2_mask.py:117-128)."""
        embed_idxs = self.find_nearest_embedding(inputs)
        if self.training and self.ema:
            self._update_buffers(inputs, embed_idxs)
        embeds = self.embed(embed_idxs)
        if self.ema and self.training:
            self._update_embedding()
        return embeds, embed_idxs

    def embed(self, idxs):
        if idxs.is_cuda and not (torch.is_grad_enabled() and self.weight.requires_grad):
            return embed_gather(self.weight, idxs)
        return super().forward(idxs)


class _CodebookOps:
    """what VectorQuantize2 and VectorQuantize2List share besides forward (quantize2_mask.py:193-208, quantize2_list.py)"""

    @torch.no_grad()
    def get_soft_codes(self, x, temp=1.0, stochastic=False):
        """x [..., D] channel-last -> (softmax(-d / temp) over the K codes [..., K], hard code [...]):
        a multinomial draw per token when `stochastic`, else the nearest code (quantize2_mask.py:193-205)."""
        d = self.codebook.compute_distances(x)
        soft = torch.softmax(d / (-temp), dim=-1)
        if stochastic:
            code = torch.multinomial(soft.reshape(-1, soft.shape[-1]), 1).reshape(soft.shape[:-1])
        else:
            code = torch.argmin(d, dim=-1)
        return soft, code

    def get_codebook_entry(self, indices, *kwargs):
        return self.codebook.embed(indices)

    def invalidate_codebook_cache(self):
        self.codebook.invalidate_codebook_cache()


class VectorQuantize2(_CodebookOps, nn.Module):
    """Reference modules/vector_quantization/quantize2_mask.py:135-210 (and quantize2.py:135)."""

    def __init__(self, codebook_size, codebook_dim=None, accept_image_fmap=True, commitment_beta=0.25,
                 decay=0.99, restart_unused_codes=True, channel_last=False):
        super().__init__()
        self.accept_image_fmap = accept_image_fmap
        self.beta = commitment_beta
        self.channel_last = channel_last
        self.restart_unused_codes = restart_unused_codes
        self.codebook = VQEmbedding(codebook_size, codebook_dim, decay=decay,
                                    restart_unused_codes=restart_unused_codes)
        self.codebook.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)
        self.assign_mode = _lib.MODE_FILTER

    def forward(self, x, codebook_mask=None, *ignorewargs, **ignorekwargs):
        need_transpose = not self.channel_last and not self.accept_image_fmap
        if self.accept_image_fmap:
            if x.dim() != 4:
                raise ValueError("accept_image_fmap=True expects x [B, C, H, W]")
            z = x                                        # NCHW read in place by the kernel
        elif need_transpose:                             # x is [B, D, N]: already channel-major
            z = x
        else:                                            # channel_last: x [B, ..., D] -> tokens [N, D]
            z = x.reshape(-1, x.shape[-1])
        K = self.codebook.n_embed
        mask = None
        if codebook_mask is not None:
            mask = codebook_mask
            if mask.dtype != torch.float32:
                mask = mask.float()
        if self.training:
            self.codebook._prep.invalidate()             # training: optimizers / EMA may write through .data
        self.codebook._prep.track_users = self.training
        zq, loss, codes = _vq_straight_through(z, self.codebook.weight, mask, self.codebook._prep, K,
                                                   float(self.beta), 1.0, self.assign_mode)
        if self.training and self.codebook.ema:
            with torch.no_grad():
                if self.accept_image_fmap or need_transpose:     # channel-major -> token rows (a view)
                    ztok = z.reshape(z.shape[0], z.shape[1], -1).permute(0, 2, 1)
                    self.codebook._ema_step(ztok, codes.reshape(-1), nchw=z.detach())
                else:
                    self.codebook._update_buffers(z, codes.reshape(-1))
                    self.codebook._update_embedding()
        if self.accept_image_fmap or need_transpose:
            x_q = zq
            x_code = codes                               # [B, H, W] / [B, N]
        else:
            x_q = zq.reshape(x.shape)
            x_code = codes.reshape(x.shape[:-1])
            if x_code.dim() > 2:                         # 'h ... d -> h (...) d' (quantize2_mask.py:167)
                x_code = x_code.reshape(x.shape[0], -1)
        return x_q, loss, (None, None, x_code)



class VectorQuantize2List(_CodebookOps, nn.Module):
    """Reference modules/vector_quantization/quantize2_list.py:135-170 (class `VectorQuantize2` there): the input is a
    LIST of channel-last tensors x_i [..., D] (one per image, any number of tokens each); returns
    (list of x_q_i, loss, (None, None, list of codes_i)) with loss = mean over the items of
    beta * mean((e - x_i)^2) + mean((e - x_i)^2).  Same parameters / buffers / state_dict keys as VectorQuantize2.

    Inference (no gradient, eval mode): ONE assign over the concatenated token rows, split afterwards.  Training or
    gradients: item by item through the same differentiable op as VectorQuantize2, so the EMA update after item i is
    seen by item i + 1 exactly as in the reference's loop."""

    def __init__(self, codebook_size, codebook_dim=None, commitment_beta=0.25, decay=0.99, restart_unused_codes=True):
        super().__init__()
        self.beta = commitment_beta
        self.restart_unused_codes = restart_unused_codes
        self.codebook = VQEmbedding(codebook_size, codebook_dim, decay=decay, restart_unused_codes=restart_unused_codes)
        self.codebook.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)
        self.assign_mode = _lib.MODE_FILTER

    def forward(self, x_list, *ignorewargs, **ignorekwargs):
        K, D = self.codebook.n_embed, self.codebook.weight.shape[1]
        n_items = len(x_list)
        if n_items == 0:
            raise ValueError("x_list is empty")
        needs_grad = torch.is_grad_enabled() and (any(x.requires_grad for x in x_list) or self.codebook.weight.requires_grad)
        if not self.training and not needs_grad:
            rows = [x.reshape(-1, D) for x in x_list]
            counts = [r.shape[0] for r in rows]
            z = torch.cat(rows, 0) if n_items > 1 else rows[0]
            zq, codes, _ = vq_assign(z, self.codebook.weight[:K], self.codebook._prep, None, beta=float(self.beta),
                                     want_loss=False, mode=self.assign_mode)
            e = embed_gather(self.codebook.weight, codes.reshape(1, -1)).reshape(-1, D)
            loss = z.new_zeros(())
            xq_list, code_list, o = [], [], 0
            for x, n in zip(x_list, counts):
                m = torch.mean((e[o:o + n] - rows[len(xq_list)]) ** 2)
                loss = loss + (self.beta * m + m)
                xq_list.append(zq[o:o + n].reshape(x.shape))
                code_list.append(codes[o:o + n].reshape(x.shape[:-1]))
                o += n
            return xq_list, loss / n_items, (None, None, code_list)
        xq_list, code_list = [], []
        loss = 0.0
        for x in x_list:
            z = x.reshape(-1, D)
            if self.training:
                self.codebook._prep.invalidate()
            self.codebook._prep.track_users = self.training
            zq, l_i, codes = _vq_straight_through(z, self.codebook.weight, None, self.codebook._prep, K,
                                                      float(self.beta), 1.0, self.assign_mode)
            if self.training and self.codebook.ema:
                with torch.no_grad():
                    self.codebook._update_buffers(z, codes.reshape(-1))
                    self.codebook._update_embedding()
            loss = loss + l_i
            xq_list.append(zq.reshape(x.shape))
            code_list.append(codes.reshape(x.shape[:-1]))
        return xq_list, loss / n_items, (None, None, code_list)



class VectorQuantizer2(nn.Module):
    """Reference modules/vector_quantization/quantize_vqgan.py:213-341 (taming-style quantizer,
    used by the fixed-granularity VQModel with beta=0.25, remap=None, legacy=False)."""

    def __init__(self, n_e, e_dim, beta, remap=None, unknown_index="random", sane_index_shape=False,
                 legacy=True):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.legacy = legacy
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)
        self.remap = remap
        if self.remap is not None:
            import numpy as np
            self.register_buffer("used", torch.tensor(np.load(self.remap)))
            self.re_embed = self.used.shape[0]
            self.unknown_index = unknown_index
            if self.unknown_index == "extra":
                self.unknown_index = self.re_embed
                self.re_embed = self.re_embed + 1
        else:
            self.re_embed = n_e
        self.sane_index_shape = sane_index_shape
        self._prep = _CodebookPrep()
        self.assign_mode = _lib.MODE_FILTER

    def invalidate_codebook_cache(self):
        """call after writing embedding.weight through `.data` in eval mode"""
        self._prep.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._prep.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._prep.invalidate()
        return out

    def remap_to_used(self, inds):
        """full-codebook indices [B, ...] -> positions in the `used` list (first occurrence); indices that
        are not in the list become `unknown_index` (an int, or the extra slot) or a random used slot
        (quantize_vqgan.py:247-259)"""
        if inds.dim() < 2:
            raise ValueError("remap_to_used expects [B, ...] indices")
        used = self.used.to(inds.device).long()
        slot = torch.arange(used.numel(), device=inds.device)
        first = torch.full((self.n_e,), used.numel(), dtype=torch.long, device=inds.device)
        first.scatter_reduce_(0, used, slot, reduce="amin")              # lookup table code -> first slot
        new = first[inds.long()]
        unknown = new == used.numel()
        if self.unknown_index == "random":
            new = torch.where(unknown, torch.randint(0, self.re_embed, new.shape, device=new.device), new)
        else:
            new = new.masked_fill(unknown, int(self.unknown_index))
        return new

    def unmap_to_all(self, inds):
        """inverse of remap_to_used; the extra (unknown) slot maps to used[0] (quantize_vqgan.py:261-268)"""
        if inds.dim() < 2:
            raise ValueError("unmap_to_all expects [B, ...] indices")
        used = self.used.to(inds.device).long()
        inds = inds.long()
        if self.re_embed > used.numel():
            inds = inds.masked_fill(inds >= used.numel(), 0)
        return used[inds]

    def forward(self, z, temp=None, rescale_logits=False, return_logits=False):
        assert temp is None or temp == 1.0, "Only for interface compatible with Gumbel"
        assert rescale_logits is False, "Only for interface compatible with Gumbel"
        assert return_logits is False, "Only for interface compatible with Gumbel"
        if z.dim() != 4:
            raise ValueError("VectorQuantizer2 expects z [B, C, H, W]")
        # legacy=False: beta*mean((zq.detach()-z)^2) + mean((zq-z.detach())^2); legacy=True swaps beta
        coef_z, coef_e = (1.0, float(self.beta)) if self.legacy else (float(self.beta), 1.0)
        if self.training:
            self._prep.invalidate()                      # the optimizer may have stepped through .data
        self._prep.track_users = self.training
        z_q, loss, codes = _vq_straight_through(z, self.embedding.weight, None, self._prep, self.n_e,
                                                    coef_z, coef_e, self.assign_mode)
        min_encoding_indices = codes.reshape(-1)
        if self.remap is not None:
            min_encoding_indices = min_encoding_indices.reshape(z.shape[0], -1)
            min_encoding_indices = self.remap_to_used(min_encoding_indices)
            min_encoding_indices = min_encoding_indices.reshape(-1, 1)
        if self.sane_index_shape:
            min_encoding_indices = min_encoding_indices.reshape(z_q.shape[0], z_q.shape[2], z_q.shape[3])
        return z_q, loss, (None, None, min_encoding_indices)

    def get_codebook_entry(self, indices, shape=None):
        if self.remap is not None:
            indices = indices.reshape(shape[0], -1)
            indices = self.unmap_to_all(indices)
            indices = indices.reshape(-1)
        z_q = embed_gather(self.embedding.weight, indices) if indices.is_cuda else self.embedding(indices)
        if shape is not None:
            z_q = z_q.view(shape)
            z_q = z_q.permute(0, 3, 1, 2).contiguous()
        return z_q
