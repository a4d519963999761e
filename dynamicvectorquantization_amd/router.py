"""Drop-in routers and the routing tail ("scatter/gather") backed by libdvq.so.

Mirrors
  * DualGrainFixedEntropyRouter, DualGrainFeatureRouter -- reference modules/dynamic_modules/RouterDual.py:6-57
  * TripleGrainFeatureRouter                            -- reference modules/dynamic_modules/RouterTriple.py:6-56
  * route_select_dual / route_select_triple             -- the eval-mode tails of
    DualGrainEncoder.forward (EncoderDual.py:134-156) and TripleGrainEncoder.forward (EncoderTriple.py:148-183)
Same constructor kwargs (including the reference's misspelt `fine_grain_ratito`), forward
signatures, return structures and state_dict keys (`gate.*`, `feature_norm_{fine,median,coarse}.*`).

The entropy gate and both selects are HIP kernels (one launch each).  The feature routers' gate
(GroupNorm -> AvgPool -> concat -> Linear[/act/Linear]) runs as the fused `dvq_router_gate_f32`
kernel whenever no gradient is needed; with autograd recording (router
training) the same math runs as differentiable torch ops on the GPU.
"""
import json

import torch
import torch.nn as nn

from . import _lib

_lib_handle = _lib.lib


def entropy_gate(entropy, threshold):
    """[...] f32 entropy -> [..., 2] int64 gate = cat[(ent <= thr), (ent > thr)] (RouterDual.py:54-56)."""
    entropy = _lib.require_cuda_f32(entropy, "entropy")
    gate = torch.empty(tuple(entropy.shape) + (2,), dtype=torch.int64, device=entropy.device)
    if entropy.numel() == 0:
        return gate
    with _lib.on_device(entropy.device):
        _lib.check(_lib_handle.dvq_entropy_gate_f32(entropy.data_ptr(), entropy.numel(), float(threshold),
                                                    gate.data_ptr(), _lib.stream_ptr(entropy.device)),
                   "dvq_entropy_gate_f32")
    return gate


def _gate_arg(gate, G):
    if not gate.is_cuda:
        raise _lib.DvqError("gate is on %s: the dvq kernels run on the GPU only" % gate.device)
    if gate.dim() != 4 or gate.shape[-1] != G:
        raise ValueError("gate must be [B, h, w, %d], got %s" % (G, tuple(gate.shape)))
    if gate.dtype == torch.int64:
        return gate.contiguous(), _lib.GATE_I64
    if gate.dtype != torch.float32:
        gate = gate.float()
    return gate.contiguous(), _lib.GATE_F32


def _route_select_dual_raw(gate, h_coarse, h_fine, out=None):
    """gate [B, hc, wc, 2] (router output, f32 logits or int64), h_coarse [B, C, hc, wc],
    h_fine [B, C, 2hc, 2wc] -> dict(h_dual, indices, codebook_mask, gate) exactly as
    DualGrainEncoder.forward returns it in eval mode (EncoderDual.py:151-156):
    indices [B, hc, wc] int64 (0 coarse / 1 fine), codebook_mask [B, 1, 2hc, 2wc] f32 (0.25 / 1.0),
    gate permuted to [B, 2, hc, wc]."""
    g, gdt = _gate_arg(gate, 2)
    h_coarse = _lib.require_cuda_f32(h_coarse, "h_coarse")
    h_fine = _lib.require_cuda_f32(h_fine, "h_fine")
    B, C, hc, wc = h_coarse.shape
    if tuple(h_fine.shape) != (B, C, 2 * hc, 2 * wc) or tuple(g.shape[:3]) != (B, hc, wc):
        raise ValueError("shape mismatch: gate %s h_coarse %s h_fine %s" %
                         (tuple(gate.shape), tuple(h_coarse.shape), tuple(h_fine.shape)))
    if out is not None:
        h_dual, indices, cmask = out
    else:
        h_dual = torch.empty_like(h_fine)
        indices = torch.empty((B, hc, wc), dtype=torch.int64, device=h_fine.device)
        cmask = torch.empty((B, 1, 2 * hc, 2 * wc), dtype=torch.float32, device=h_fine.device)
    if h_dual.numel() == 0:
        return {"h_dual": h_dual, "indices": indices, "codebook_mask": cmask, "gate": gate.permute(0, 3, 1, 2)}
    with _lib.on_device(h_fine.device):
        _lib.check(_lib_handle.dvq_route_select_dual_f32(
            g.data_ptr(), gdt, h_coarse.data_ptr(), h_fine.data_ptr(), B, C, hc, wc,
            h_dual.data_ptr(), indices.data_ptr(), cmask.data_ptr(), _lib.stream_ptr(h_fine.device)),
            "dvq_route_select_dual_f32")
    return {"h_dual": h_dual, "indices": indices, "codebook_mask": cmask, "gate": gate.permute(0, 3, 1, 2)}


def _route_select_dual_entropy_raw(entropy, threshold, h_coarse, h_fine, out=None):
    """DualGrainFixedEntropyRouter.forward + the routing tail of DualGrainEncoder.forward in ONE kernel
    (RouterDual.py:53-57 + EncoderDual.py:134-156): entropy [B, hc, wc] f32 -> the same dict as
    route_select_dual, "gate" being the router's int64 gate permuted to [B, 2, hc, wc].
    out = (h_dual, indices, codebook_mask, gate[B, hc, wc, 2] int64) to reuse buffers."""
    entropy = _lib.require_cuda_f32(entropy, "entropy")
    h_coarse = _lib.require_cuda_f32(h_coarse, "h_coarse")
    h_fine = _lib.require_cuda_f32(h_fine, "h_fine")
    B, C, hc, wc = h_coarse.shape
    if tuple(h_fine.shape) != (B, C, 2 * hc, 2 * wc) or tuple(entropy.shape) != (B, hc, wc):
        raise ValueError("shape mismatch: entropy %s h_coarse %s h_fine %s" %
                         (tuple(entropy.shape), tuple(h_coarse.shape), tuple(h_fine.shape)))
    if out is not None:
        h_dual, indices, cmask, gate = out
    else:
        h_dual = torch.empty_like(h_fine)
        indices = torch.empty((B, hc, wc), dtype=torch.int64, device=h_fine.device)
        cmask = torch.empty((B, 1, 2 * hc, 2 * wc), dtype=torch.float32, device=h_fine.device)
        gate = torch.empty((B, hc, wc, 2), dtype=torch.int64, device=h_fine.device)
    if h_dual.numel() > 0:
        with _lib.on_device(h_fine.device):
            _lib.check(_lib_handle.dvq_route_select_dual_entropy_f32(
                entropy.data_ptr(), float(threshold), h_coarse.data_ptr(), h_fine.data_ptr(), B, C, hc, wc,
                h_dual.data_ptr(), indices.data_ptr(), cmask.data_ptr(), gate.data_ptr(),
                _lib.stream_ptr(h_fine.device)), "dvq_route_select_dual_entropy_f32")
    return {"h_dual": h_dual, "indices": indices, "codebook_mask": cmask, "gate": gate.permute(0, 3, 1, 2)}


def _route_select_triple_raw(gate, h_coarse, h_median, h_fine, out=None):
    """gate [B, hc, wc, 3]; h_coarse [B,C,hc,wc], h_median [B,C,2hc,2wc], h_fine [B,C,4hc,4wc]
    -> dict(h_triple, indices, codebook_mask, gate) as TripleGrainEncoder.forward (EncoderTriple.py:178-183);
    mask values 0.0625 / 0.25 / 1.0."""
    g, gdt = _gate_arg(gate, 3)
    h_coarse = _lib.require_cuda_f32(h_coarse, "h_coarse")
    h_median = _lib.require_cuda_f32(h_median, "h_median")
    h_fine = _lib.require_cuda_f32(h_fine, "h_fine")
    B, C, hc, wc = h_coarse.shape
    if (tuple(h_median.shape) != (B, C, 2 * hc, 2 * wc) or tuple(h_fine.shape) != (B, C, 4 * hc, 4 * wc)
            or tuple(g.shape[:3]) != (B, hc, wc)):
        raise ValueError("shape mismatch: gate %s h_coarse %s h_median %s h_fine %s" %
                         (tuple(gate.shape), tuple(h_coarse.shape), tuple(h_median.shape), tuple(h_fine.shape)))
    if out is not None:
        h_triple, indices, cmask = out
    else:
        h_triple = torch.empty_like(h_fine)
        indices = torch.empty((B, hc, wc), dtype=torch.int64, device=h_fine.device)
        cmask = torch.empty((B, 1, 4 * hc, 4 * wc), dtype=torch.float32, device=h_fine.device)
    if h_triple.numel() == 0:
        return {"h_triple": h_triple, "indices": indices, "codebook_mask": cmask, "gate": gate.permute(0, 3, 1, 2)}
    with _lib.on_device(h_fine.device):
        _lib.check(_lib_handle.dvq_route_select_triple_f32(
            g.data_ptr(), gdt, h_coarse.data_ptr(), h_median.data_ptr(), h_fine.data_ptr(), B, C, hc, wc,
            h_triple.data_ptr(), indices.data_ptr(), cmask.data_ptr(), _lib.stream_ptr(h_fine.device)),
            "dvq_route_select_triple_f32")
    return {"h_triple": h_triple, "indices": indices, "codebook_mask": cmask, "gate": gate.permute(0, 3, 1, 2)}



class _RouteSelectGrad(torch.autograd.Function):
    """The select kernels under autograd.  The reference's `torch.where(indices_repeat == g, h_g_upsampled, ...)`
    (EncoderDual.py:137-140, EncoderTriple.py:151-159) is differentiable in every branch: the gradient of a
    fine cell goes to h_fine, that of a coarse / median cell is the SUM over the cell's 2x2 / 4x4 output
    positions (backward of repeat_interleave).  Forward = the HIP kernel; backward = three torch ops
    (training only)."""

    @staticmethod
    def forward(ctx, run, h_coarse, h_median, h_fine):
        sel = run()
        h_out = sel["h_triple"] if h_median is not None else sel["h_dual"]
        ctx.save_for_backward(sel["indices"])
        ctx.triple = h_median is not None
        ctx.mark_non_differentiable(sel["indices"], sel["codebook_mask"])
        ctx.gate = sel["gate"]
        return h_out, sel["indices"], sel["codebook_mask"]

    @staticmethod
    def backward(ctx, g, _gi, _gm):
        (ind,) = ctx.saved_tensors
        fine_id, s = (2, 4) if ctx.triple else (1, 2)
        up = ind.repeat_interleave(s, 1).repeat_interleave(s, 2).unsqueeze(1)        # [B, 1, H, W]
        zero = g.new_zeros(())
        pool = torch.nn.functional.avg_pool2d
        g_fine = torch.where(up == fine_id, g, zero) if ctx.needs_input_grad[3] else None
        g_coarse = pool(torch.where(up == 0, g, zero), s) * float(s * s) if ctx.needs_input_grad[1] else None
        g_median = None
        if ctx.triple and ctx.needs_input_grad[2]:
            g_median = pool(torch.where(up == 1, g, zero), 2) * 4.0
        return None, g_coarse, g_median, g_fine


def _select(run, key, h_coarse, h_median, h_fine):
    feats = [t for t in (h_coarse, h_median, h_fine) if t is not None]
    if torch.is_grad_enabled() and any(t.requires_grad for t in feats):
        h_out, indices, cmask = _RouteSelectGrad.apply(run, h_coarse, h_median, h_fine)
        sel = run.last
        return {key: h_out, "indices": indices, "codebook_mask": cmask, "gate": sel["gate"]}
    return run()


class _Run:
    """callable that keeps the dict of its last call (the autograd wrapper returns tensors only)"""

    def __init__(self, fn):
        self.fn, self.last = fn, None

    def __call__(self):
        self.last = self.fn()
        return self.last


def route_select_dual(gate, h_coarse, h_fine, out=None):
    """DualGrainEncoder.forward's eval-mode routing tail (EncoderDual.py:134-156), differentiable in
    h_coarse / h_fine like the reference's torch.where; see _route_select_dual_raw for shapes."""
    return _select(_Run(lambda: _route_select_dual_raw(gate, h_coarse.detach(), h_fine.detach(), out)),
                   "h_dual", h_coarse, None, h_fine)


def route_select_dual_entropy(entropy, threshold, h_coarse, h_fine, out=None):
    """fixed-entropy router + routing tail in one kernel, differentiable in h_coarse / h_fine."""
    return _select(_Run(lambda: _route_select_dual_entropy_raw(entropy, threshold, h_coarse.detach(),
                                                               h_fine.detach(), out)),
                   "h_dual", h_coarse, None, h_fine)


def route_select_triple(gate, h_coarse, h_median, h_fine, out=None):
    """TripleGrainEncoder.forward's eval-mode routing tail (EncoderTriple.py:148-183), differentiable in
    the three branches."""
    return _select(_Run(lambda: _route_select_triple_raw(gate, h_coarse.detach(), h_median.detach(),
                                                         h_fine.detach(), out)),
                   "h_triple", h_coarse, h_median, h_fine)


class DualGrainFixedEntropyRouter(nn.Module):
    """RouterDual.py:46-57.  threshold = json[str(int(100 - ratio*100))]."""

    def __init__(self, json_path, fine_grain_ratito):
        super().__init__()
        with open(json_path, "r", encoding="utf-8") as f:
            content = json.load(f)
        self.fine_grain_threshold = content["{}".format(str(int(100 - fine_grain_ratito * 100)))]

    def forward(self, h_fine=None, h_coarse=None, entropy=None):
        return entropy_gate(entropy, self.fine_grain_threshold)


_ACT_OF_GATE = {"1layer-fc": _lib.ACT_NONE, "2layer-fc-SiLu": _lib.ACT_SILU, "2layer-fc-ReLu": _lib.ACT_RELU}


def _needs_autograd(module, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


class _GateWeightPrep:
    """The hidden-layer weight of a router's gate MLP as split fp16 matrix-core tile images
    (`dvq_router_gate_prepare_f32`), rebuilt only when the weight changes: keyed on (data_ptr, _version, shape,
    device) like quantize._CodebookPrep, with the same caveat -- writes through `.data` do not bump the version, so
    the router modules call invalidate() from `_load_from_state_dict`, `_apply` and every training-mode forward."""

    def __init__(self):
        self.key = None
        self.buf = None
        self._retired = []
        self._built = None                   # (stream handle, event) of the last build: other streams wait for it once

    def invalidate(self):
        self.key = None

    def get(self, w1, nb, C, gn=None, owners=None):
        """gn: (weights, biases) of the branches' GroupNorms, coarse -> fine (None: no normalisation): their per-branch maxima
        are prepared into the same buffer (`dvq_router_gate_prepare_norm_f32`) and refreshed when a parameter's version changes.
        owners: the module PARAMETERS the tensors were derived from (w1's, then the GroupNorms'): the cache is keyed on THEIR
        (data_ptr, _version, dtype) -- a converted copy (non-fp32 or non-contiguous parameters) is a fresh temporary on every
        call, whose address and version say nothing"""
        hidden = w1.shape[0]
        ow = owners if owners is not None else [w1] + (list(gn[0] + gn[1]) if gn is not None else [])
        key = (ow[0].data_ptr(), ow[0]._version, ow[0].dtype, tuple(w1.shape), w1.device)
        gkey = None if gn is None else tuple((t.data_ptr(), t._version, t.dtype) for t in ow[1:])
        if key == self.key and gkey != getattr(self, "gkey", None):
            self.key = None                              # (simplest: a changed GroupNorm parameter rebuilds the whole prep)
        if key != self.key:
            nbytes = _lib_handle.dvq_router_gate_prep_bytes(nb, C, hidden)
            if nbytes == 0:
                raise _lib.DvqError("unsupported gate shape nb=%d C=%d hidden=%d" % (nb, C, hidden))
            # a FRESH buffer per rebuild (other streams may still have kernels queued against the old images; the last two
            # replaced buffers are kept alive)
            if self.buf is not None:
                self._retired = (self._retired + [self.buf])[-2:]
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
            with _lib.on_device(w1.device):
                _lib.check(_lib_handle.dvq_router_gate_prepare_f32(
                    w1.data_ptr(), nb, C, hidden, self.buf.data_ptr(), self.buf.numel(), _lib.stream_ptr(w1.device)),
                    "dvq_router_gate_prepare_f32")
                if gn is not None:
                    gw, gb = gn
                    p = lambda t: t.data_ptr()
                    _lib.check(_lib_handle.dvq_router_gate_prepare_norm_f32(
                        nb, C, hidden, p(gw[0]), p(gb[0]), p(gw[1]) if nb == 3 else None, p(gb[1]) if nb == 3 else None,
                        p(gw[-1]), p(gb[-1]), self.buf.data_ptr(), self.buf.numel(), _lib.stream_ptr(w1.device)),
                        "dvq_router_gate_prepare_norm_f32")
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(w1.device))
                self._built = (_lib.stream_ptr(w1.device), ev)
            self.key = key
            self.gkey = gkey
        elif self._built is not None and not torch.cuda.is_current_stream_capturing():
            if self._built[0] != _lib.stream_ptr(w1.device):     # built on another stream: order this one behind it, once
                if self._built[1].query():
                    self._built = None
                else:
                    torch.cuda.current_stream(w1.device).wait_event(self._built[1])
        return self.buf


def fused_router_gate(gate, gate_type, norms, branches, weight_prep=None):
    """gate: the router's nn.Linear / nn.Sequential; norms / branches: coarse -> fine lists of the
    GroupNorm (or Identity) modules and of the [B, C, rows, cols] feature maps.
    -> logits [B, hc, wc, len(branches)] f32 (RouterDual.py:35-43, RouterTriple.py:46-56): one pass over the features
    (statistics + per-cell averages), then the gate MLP on the averages.  weight_prep: a _GateWeightPrep kept by the
    module (None: the weight images are rebuilt inside the call)."""
    nb = len(branches)
    hs = [_lib.require_cuda_f32(h, "router input") for h in branches]
    B, C, hc, wc = hs[0].shape
    for i, h in enumerate(hs):
        if tuple(h.shape) != (B, C, hc << i, wc << i):
            raise ValueError("router branch %d has shape %s, expected %s" % (i, tuple(h.shape), (B, C, hc << i, wc << i)))
    groups, eps, gw, gb = 0, 0.0, [None] * nb, [None] * nb
    if isinstance(norms[0], nn.GroupNorm):
        groups, eps = norms[0].num_groups, norms[0].eps
        gw = [n.weight.detach().float().contiguous() for n in norms]
        gb = [n.bias.detach().float().contiguous() for n in norms]
    act = _ACT_OF_GATE[gate_type]
    if act == _lib.ACT_NONE:
        w1 = b1 = None
        w2, b2, hidden = gate.weight, gate.bias, 0
    else:
        w1, b1, w2, b2 = gate[0].weight, gate[0].bias, gate[2].weight, gate[2].bias
        hidden = w1.shape[0]
    w1, b1, w2, b2 = [None if t is None else (t.detach() if (t.dtype == torch.float32 and t.is_contiguous())
                                               else t.detach().float().contiguous()) for t in (w1, b1, w2, b2)]
    dev = hs[0].device
    out = torch.empty((B, hc, wc, nb), dtype=torch.float32, device=dev)
    ws_bytes = _lib_handle.dvq_router_gate_workspace_bytes(nb, B, C, hc, wc, groups, hidden)
    if ws_bytes == 0:
        raise _lib.DvqError("unsupported router shape")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    med = hs[1] if nb == 3 else None
    owners = None
    if w1 is not None:
        owners = [gate[0].weight] + ([n.weight for n in norms] + [n.bias for n in norms] if groups > 0 else [])
    prep = (weight_prep.get(w1, nb, C, gn=(gw, gb) if groups > 0 else None, owners=owners)
            if (weight_prep is not None and w1 is not None) else None)
    with _lib.on_device(dev):
        _lib.check(_lib_handle.dvq_router_gate_f32(
            nb, hs[0].data_ptr(), ptr(med), hs[-1].data_ptr(), B, C, hc, wc, groups, float(eps),
            ptr(gw[0]), ptr(gb[0]), ptr(gw[1]) if nb == 3 else None, ptr(gb[1]) if nb == 3 else None,
            ptr(gw[-1]), ptr(gb[-1]), ptr(w1), ptr(b1), ptr(w2), ptr(b2), hidden, act, ptr(prep),
            out.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev)), "dvq_router_gate_f32")
    return out


class _GateCacheMixin:
    """invalidation hooks of the cached gate-weight images (see _GateWeightPrep)"""

    def _gate_prep(self):
        if self.training:
            self._gate_weight_prep.invalidate()
        return self._gate_weight_prep

    def invalidate_gate_cache(self):
        self._gate_weight_prep.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):
        self._gate_weight_prep.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._gate_weight_prep.invalidate()
        return super()._apply(fn, *args, **kwargs)


def _make_gate(gate_type, width, splits, allow_relu):
    if gate_type == "1layer-fc":
        return nn.Linear(width, splits)
    if gate_type == "2layer-fc-SiLu":
        return nn.Sequential(nn.Linear(width, width), nn.SiLU(inplace=True), nn.Linear(width, splits))
    if allow_relu and gate_type == "2layer-fc-ReLu":
        return nn.Sequential(nn.Linear(width, width), nn.ReLU(inplace=True), nn.Linear(width, splits))
    raise NotImplementedError()


def _make_norm(normalization_type, num_channels):
    if normalization_type == "none":
        return nn.Identity()
    if "group" in normalization_type:  # like "group-32"
        num_groups = int(normalization_type.split("-")[-1])
        return nn.GroupNorm(num_groups=num_groups, num_channels=num_channels, eps=1e-6, affine=True)
    raise NotImplementedError()


class DualGrainFeatureRouter(_GateCacheMixin, nn.Module):
    """RouterDual.py:6-43: GroupNorm both branches, 2x2 average-pool the fine one, concat channels,
    NHWC, gate MLP -> logits [B, hc, wc, 2]."""

    def __init__(self, num_channels, normalization_type="none", gate_type="1layer-fc"):
        super().__init__()
        self.gate_pool = nn.AvgPool2d(2, 2)
        self.gate_type = gate_type
        self.gate = _make_gate(gate_type, num_channels * 2, 2, allow_relu=False)
        self.num_splits = 2
        self.normalization_type = normalization_type
        self.feature_norm_fine = _make_norm(normalization_type, num_channels)
        self.feature_norm_coarse = _make_norm(normalization_type, num_channels)
        self._gate_weight_prep = _GateWeightPrep()

    def forward(self, h_fine, h_coarse, entropy=None):
        if h_fine.is_cuda and not _needs_autograd(self, h_fine, h_coarse):
            return fused_router_gate(self.gate, self.gate_type, [self.feature_norm_coarse, self.feature_norm_fine],
                                     [h_coarse, h_fine], weight_prep=self._gate_prep())
        h_fine = self.feature_norm_fine(h_fine)
        h_coarse = self.feature_norm_coarse(h_coarse)
        avg_h_fine = self.gate_pool(h_fine)
        h_logistic = torch.cat([h_coarse, avg_h_fine], dim=1).permute(0, 2, 3, 1)
        return self.gate(h_logistic)


class TripleGrainFeatureRouter(_GateCacheMixin, nn.Module):
    """RouterTriple.py:6-56: three GroupNorms, 4x4 / 2x2 pools, concat -> MLP -> logits [B, hc, wc, 3]."""

    def __init__(self, num_channels, normalization_type="none", gate_type="1layer-fc"):
        super().__init__()
        self.gate_median_pool = nn.AvgPool2d(2, 2)
        self.gate_fine_pool = nn.AvgPool2d(4, 4)
        self.num_splits = 3
        self.gate_type = gate_type
        self.gate = _make_gate(gate_type, num_channels * 3, 3, allow_relu=True)
        self.normalization_type = normalization_type
        self.feature_norm_fine = _make_norm(normalization_type, num_channels)
        self.feature_norm_median = _make_norm(normalization_type, num_channels)
        self.feature_norm_coarse = _make_norm(normalization_type, num_channels)
        self._gate_weight_prep = _GateWeightPrep()

    def forward(self, h_fine, h_median, h_coarse, entropy=None):
        if h_fine.is_cuda and not _needs_autograd(self, h_fine, h_median, h_coarse):
            return fused_router_gate(self.gate, self.gate_type,
                                     [self.feature_norm_coarse, self.feature_norm_median, self.feature_norm_fine],
                                     [h_coarse, h_median, h_fine], weight_prep=self._gate_prep())
        h_fine = self.feature_norm_fine(h_fine)
        h_median = self.feature_norm_median(h_median)
        h_coarse = self.feature_norm_coarse(h_coarse)
        avg_h_fine = self.gate_fine_pool(h_fine)
        avg_h_median = self.gate_median_pool(h_median)
        h_logistic = torch.cat([h_coarse, avg_h_median, avg_h_fine], dim=1).permute(0, 2, 3, 1)
        return self.gate(h_logistic)
