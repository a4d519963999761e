"""The 1x1 quant_conv of the stage-1 models on the HIP kernels of libdvq.so, optionally with the router select
fused in (second half).

Reference: `self.quant_conv = torch.nn.Conv2d(z_channels, embed_dim, 1)` applied to h_dual / h_triple between the
routing tail and the quantizer (models/stage1_dynamic/dqvae_dual_feat.py:34,66, dqvae_triple_feat.py:39,75,
models/stage1/vqgan.py:42,70).  A 1x1 conv is pointwise, so it commutes with the select: `quant_conv_select`
computes W src + b per output position straight from the encoder branch that won the position's cell -- h_dual is
never written -- and returns the select's by-products (grain indices, codebook_mask, the int64 gate) with it.
fp16 matrix cores with both operands split hi + lo (fp32-grade, 2^-22 products): tolerance parity with the
reference's conv (1e-5 relative to |x||w|); the assign downstream is bit-exact GIVEN this tensor.
Inference only (no autograd); `usable(conv)` says whether a module qualifies, otherwise callers keep torch's conv.
"""
import weakref

import torch
from torch import nn

from . import _lib

_lib_handle = _lib.lib


class _ConvPrep:
    """hi / lo fp16 tile images of one conv weight, rebuilt when the weight tensor changes (storage / version)"""

    def __init__(self):
        self.key, self.buf = None, None
        self._retired = []                   # replaced image buffers other streams may still be reading
        self._built = None                   # (stream handle, event) of the last build: other streams wait for it once

    def get(self, conv):
        w = conv.weight
        bias = conv.bias
        D = w.shape[0]
        key = (w.data_ptr(), w._version, None if bias is None else (bias.data_ptr(), bias._version), D, w.device)
        if key != self.key:
            nbytes = _lib_handle.dvq_qconv_prep_bytes(D)
            if nbytes == 0:
                raise _lib.DvqError("quant_conv: unsupported channel count %d" % D)
            # a FRESH buffer per rebuild: streams that still have kernels queued against the old images keep reading the old buffer
            # (kept alive here until two more rebuilds have happened)
            if self.buf is not None:
                self._retired = (self._retired + [self.buf])[-2:]
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            w2 = _lib.require_cuda_f32(w.detach().reshape(D, D), "quant_conv.weight")
            b2 = None if bias is None else _lib.require_cuda_f32(bias.detach(), "quant_conv.bias")
            with _lib.on_device(w.device):
                _lib.check(_lib_handle.dvq_qconv_prepare_f32(w2.data_ptr(), _lib.ptr(b2), D, self.buf.data_ptr(),
                                                             self.buf.numel(), _lib.stream_ptr(w.device)),
                           "dvq_qconv_prepare_f32")
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(w.device))
                self._built = (_lib.stream_ptr(w.device), ev)
            self.key = key
        elif self._built is not None and not torch.cuda.is_current_stream_capturing():
            if self._built[0] != _lib.stream_ptr(w.device):
                if self._built[1].query():
                    self._built = None               # long done: nothing to order any more
                else:
                    torch.cuda.current_stream(w.device).wait_event(self._built[1])
        return self.buf


_PREPS = {}


def _prep_of(conv):
    p = _PREPS.get(id(conv))
    if p is None or p[0]() is not conv:
        if len(_PREPS) > 64:
            _PREPS.clear()
        p = (weakref.ref(conv), _ConvPrep())
        _PREPS[id(conv)] = p
    if conv.training:
        p[1].key = None                      # optimizers may write through .data
    return p[1]


def invalidate(conv):
    """call after writing conv.weight / conv.bias through `.data` in eval mode"""
    p = _PREPS.get(id(conv))
    if p is not None:
        p[1].key = None


def usable(conv):
    """an nn.Conv2d(D, D, kernel_size=1) with D in (64, 128, 256), plain stride / groups, on the GPU in fp32"""
    return (isinstance(conv, nn.Conv2d) and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (0, 0) and conv.groups == 1 and tuple(conv.dilation) == (1, 1)
            and conv.in_channels == conv.out_channels and conv.in_channels in (64, 128, 256)
            and conv.weight.is_cuda and conv.weight.dtype == torch.float32)


def quant_conv(conv, x):
    """conv(x) for x [B, D, *spatial] through `dvq_qconv_f32` (no autograd)."""
    if not usable(conv):
        raise _lib.DvqError("quant_conv: the module is not a 1x1 nn.Conv2d(D, D) on the GPU")
    x = _lib.require_cuda_f32(x, "x")
    B, D = x.shape[0], x.shape[1]
    if D != conv.in_channels:
        raise ValueError("x has %d channels, the conv expects %d" % (D, conv.in_channels))
    HW = int(torch.Size(x.shape[2:]).numel())
    h = torch.empty_like(x)
    if B * HW == 0:
        return h
    with _lib.on_device(x.device):
        pbuf = _prep_of(conv).get(conv)
        _lib.check(_lib_handle.dvq_qconv_f32(x.data_ptr(), pbuf.data_ptr(), B, D, HW, h.data_ptr(),
                                             _lib.stream_ptr(x.device)), "dvq_qconv_f32")
    return h


def quant_conv_select(conv, h_coarse, h_fine, h_median=None, gate=None, entropy=None, threshold=None, out=None):
    """route select + quant_conv as ONE kernel (`dvq_qconv_select_f32`).
    dual: h_coarse [B, D, hc, wc], h_fine [B, D, 2hc, 2wc], `gate` [B, hc, wc, 2] or `entropy` [B, hc, wc] + threshold;
    triple: plus h_median [B, D, 2hc, 2wc], h_fine [B, D, 4hc, 4wc], gate [B, hc, wc, 3].
    -> dict(h [B, D, S hc, S wc], indices [B, hc, wc] i64, codebook_mask [B, 1, S hc, S wc], gate).
    `out` = (h, indices, codebook_mask, gate_out) preallocated (benchmark / graph capture)."""
    if not usable(conv):
        raise _lib.DvqError("quant_conv_select: the module is not a 1x1 nn.Conv2d(D, D) on the GPU")
    nb = 2 if h_median is None else 3
    S = 2 if nb == 2 else 4
    h_coarse = _lib.require_cuda_f32(h_coarse, "h_coarse")
    h_fine = _lib.require_cuda_f32(h_fine, "h_fine")
    if h_median is not None:
        h_median = _lib.require_cuda_f32(h_median, "h_median")
    B, D, hc, wc = h_coarse.shape
    if tuple(h_fine.shape) != (B, D, S * hc, S * wc) or D != conv.in_channels or \
            (h_median is not None and tuple(h_median.shape) != (B, D, 2 * hc, 2 * wc)):
        raise ValueError("shape mismatch between the branches / the conv")
    if (gate is None) == (entropy is None):
        raise ValueError("give exactly one of gate / entropy")
    dev = h_fine.device
    gate_out = None
    if entropy is not None:
        if nb != 2 or threshold is None:
            raise ValueError("the entropy gate is a dual-granularity router and needs a threshold")
        g, kind, thr = _lib.require_cuda_f32(entropy, "entropy"), _lib.GATE_ENTROPY, float(threshold)
        if tuple(g.shape) != (B, hc, wc):
            raise ValueError("entropy must be [B, hc, wc]")
        gate_out = out[3] if out is not None else torch.empty((B, hc, wc, 2), dtype=torch.int64, device=dev)
    else:
        if gate.dim() != 4 or tuple(gate.shape) != (B, hc, wc, nb) or not gate.is_cuda:
            raise ValueError("gate must be a GPU tensor [B, hc, wc, %d]" % nb)
        kind = _lib.GATE_I64 if gate.dtype == torch.int64 else _lib.GATE_F32
        g = gate.contiguous() if gate.dtype in (torch.int64, torch.float32) else gate.float().contiguous()
        thr = 0.0
    if out is not None:
        h, indices, cmask = out[0], out[1], out[2]
        if tuple(h.shape) != tuple(h_fine.shape) or not (h.is_contiguous() and indices.is_contiguous() and cmask.is_contiguous()):
            raise ValueError("out tensors must be contiguous and shaped like the op's outputs")
    else:
        h = torch.empty_like(h_fine)
        indices = torch.empty((B, hc, wc), dtype=torch.int64, device=dev)
        cmask = torch.empty((B, 1, S * hc, S * wc), dtype=torch.float32, device=dev)
    if h.numel() > 0:
        with _lib.on_device(dev):
            pbuf = _prep_of(conv).get(conv)
            _lib.check(_lib_handle.dvq_qconv_select_f32(
                nb, g.data_ptr(), kind, thr, h_coarse.data_ptr(), _lib.ptr(h_median), h_fine.data_ptr(), pbuf.data_ptr(),
                B, D, hc, wc, h.data_ptr(), indices.data_ptr(), cmask.data_ptr(), _lib.ptr(gate_out),
                _lib.stream_ptr(dev)), "dvq_qconv_select_f32")
    return {"h": h, "indices": indices, "codebook_mask": cmask, "gate": gate_out if entropy is not None else gate}
