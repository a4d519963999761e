"""Drop-in DualGrainSeperatePermuter (reference modules/dynamic_modules/permuter.py:7-135; the class
name keeps the reference's spelling) backed by the stream-compaction kernels of libdvq.so.

forward(indices [B, fine_hw, fine_hw], grain_indices [B, coarse_hw, coarse_hw]) -> dict of
coarse/fine content, position and segment sequences (int64, EOS-terminated, PAD-filled to the
longest sequence of the batch -- the one device->host read `pad_sequence` implies; with `max_len=(Lc, Lf)`
the sequences are padded to those lengths instead and nothing is read back: ONE kernel, no host sync);
forward_back(...) -> dense [B, fine_hw, fine_hw] codes.  Pure integer work, bit-exact.
"""
import ctypes

import torch
from torch import nn

from . import _lib

_L = _lib.lib


def _i64_cuda(t, name):
    if not t.is_cuda:
        raise _lib.DvqError("%s is on %s: the dvq kernels run on the GPU only" % (name, t.device))
    if t.dtype != torch.int64:
        t = t.long()
    return t.contiguous()


class DualGrainSeperatePermuter(nn.Module):
    def __init__(self, coarse_hw=16, fine_hw=32, content_pad_code=1024, content_eos_code=1025,
                 coarse_position_pad_code=256, coarse_position_eos_code=257,
                 fine_position_pad_code=1024, fine_position_eos_code=1025,
                 fine_position_order="region-first"):
        super().__init__()
        self.hw1 = coarse_hw
        self.hw2 = fine_hw // coarse_hw
        self.fine_hw = fine_hw
        self.hw2_square = int(self.hw2 * self.hw2)
        if self.hw2 != 2 or fine_hw != 2 * coarse_hw:
            raise NotImplementedError("dual granularity: fine_hw must be 2 * coarse_hw (as in the reference's forward_back)")
        self.content_pad_code = content_pad_code
        self.content_eos_code = content_eos_code
        self.coarse_position_pad_code = coarse_position_pad_code
        self.coarse_position_eos_code = coarse_position_eos_code
        self.fine_position_pad_code = fine_position_pad_code
        self.fine_position_eos_code = fine_position_eos_code
        self.fine_position_order = fine_position_order
        assert self.fine_position_order in ["row-first", "region-first"]
        self._special = (ctypes.c_int64 * 6)(content_pad_code, content_eos_code, coarse_position_pad_code,
                                            coarse_position_eos_code, fine_position_pad_code, fine_position_eos_code)

    def max_lengths(self):
        """(Lc, Lf) that hold any grain map: every cell coarse / every cell fine, plus the EOS"""
        return self.hw1 * self.hw1 + 1, self.hw2_square * self.hw1 * self.hw1 + 1

    def forward(self, indices, grain_indices, max_len=None, out=None):
        """max_len = (Lc, Lf): pad to these lengths (>= the batch maxima + 1; `max_lengths()` always suffices) instead of
        reading the batch maxima back -- the extra columns hold PAD, exactly what pad_sequence would put there in a longer
        batch.  The lengths are the CALLER's promise (checking them would need the host round trip this argument exists to avoid):
        entries that do not fit are dropped by the kernel, EOS included; lengths below 1 are rejected.
        out: six preallocated contiguous int64 [B, Lc] / [B, Lf] tensors on the inputs' device, in the order coarse content /
        position / segment, fine content / position / segment (benchmark / graph capture)."""
        indices = _i64_cuda(indices, "indices")
        grain = _i64_cuda(grain_indices, "grain_indices")
        B = indices.shape[0]
        hc = self.hw1
        if tuple(indices.shape) != (B, self.fine_hw, self.fine_hw) or tuple(grain.shape) != (B, hc, hc):
            raise ValueError("indices %s / grain_indices %s do not match coarse_hw=%d fine_hw=%d" %
                             (tuple(indices.shape), tuple(grain.shape), hc, self.fine_hw))
        dev = indices.device
        with _lib.on_device(dev):
            st = _lib.stream_ptr(dev)
            if max_len is not None:
                Lc, Lf = int(max_len[0]), int(max_len[1])
                if Lc < 1 or Lf < 1:
                    raise ValueError("max_len must be at least (1, 1), got %s" % (tuple(max_len),))
            else:
                counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
                maxes = torch.empty(2, dtype=torch.int32, device=dev)
                _lib.check(_L.dvq_permute_dual_count_i64(grain.data_ptr(), B, hc, hc, counts.data_ptr(), maxes.data_ptr(), st),
                           "dvq_permute_dual_count_i64")
                mc, mf = maxes.tolist()                               # the sync pad_sequence implies
                Lc, Lf = mc + 1, 4 * mf + 1
            if out is not None:
                outs = list(out)
                if [tuple(o.shape) for o in outs] != [(B, Lc)] * 3 + [(B, Lf)] * 3 or \
                        any(o.dtype != torch.int64 or o.device != dev or not o.is_contiguous() for o in outs):
                    raise ValueError("out tensors must be 3 x [B, Lc] and 3 x [B, Lf], contiguous int64 on %s" % dev)
            else:
                outs = [torch.empty((B, Lc), dtype=torch.int64, device=dev) for _ in range(3)] + \
                       [torch.empty((B, Lf), dtype=torch.int64, device=dev) for _ in range(3)]
            order = 0 if self.fine_position_order == "region-first" else 1
            _lib.check(_L.dvq_permute_dual_forward_i64(indices.data_ptr(), grain.data_ptr(), B, hc, hc, order, Lc, Lf,
                                                       self._special, *[o.data_ptr() for o in outs], st),
                       "dvq_permute_dual_forward_i64")
        return {"coarse_content": outs[0], "fine_content": outs[3], "coarse_position": outs[1],
                "fine_position": outs[4], "coarse_segment": outs[2], "fine_segment": outs[5]}

    def forward_back(self, coarse_content, fine_content, coarse_position, fine_position):
        cc, fc = _i64_cuda(coarse_content, "coarse_content"), _i64_cuda(fine_content, "fine_content")
        cp, fp = _i64_cuda(coarse_position, "coarse_position"), _i64_cuda(fine_position, "fine_position")
        B, Lc = cc.shape
        Lf = fc.shape[1]
        if tuple(cp.shape) != (B, Lc) or tuple(fp.shape) != (B, Lf):
            raise ValueError("content / position shapes differ")
        target = torch.empty((B, self.fine_hw, self.fine_hw), dtype=torch.int64, device=cc.device)
        with _lib.on_device(cc.device):
            _lib.check(_L.dvq_permute_dual_backward_i64(cc.data_ptr(), fc.data_ptr(), cp.data_ptr(), fp.data_ptr(),
                                                        B, Lc, Lf, self.hw1, self.hw1,
                                                        self.coarse_position_eos_code, self.fine_position_eos_code,
                                                        target.data_ptr(), _lib.stream_ptr(cc.device)),
                       "dvq_permute_dual_backward_i64")
        return target
