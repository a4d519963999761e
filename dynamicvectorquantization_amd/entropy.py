"""Patch-entropy map that feeds the fixed-entropy router (BASELINE configs[2]).

Drop-in for `Entropy` of the reference (models/stage1_dynamic/dqvae_dual_entropy.py:13-63: grayscale,
non-overlapping 16x16 patches, 32-bin Gaussian-KDE histogram over [0, 1] with sigma 0.01, entropy
-sum p ln p with the reference's 1e-40 epsilons -- an fp32 SUBNORMAL: denormals must stay enabled, which
is PyTorch-ROCm's and hipcc's default).  The forward is ONE fused kernel, `dvq_entropy_map_f32`
(SURVEY.md section 8 row f3): the image is read once, the reference's [B*P, 256, 32] intermediate (2.1 GB
at B = 256) never exists.  Transcendental fp32 math: parity to 1e-5, grain maps equal away from the
threshold.  GPU only (CPU tensors raise: the package has no CPU path).  The fused kernel covers patch size 16, the
size every reference config uses; any other patch size (scripts/tools/calculate_entropy_thresholds.py takes
--patch_size) runs the same arithmetic as chunked torch ops ON THE GPU.
The reference's tensor-op sequence used as the comparator in tests lives in oracle/entropy_torch.py.
"""
import torch
from torch import nn

from . import _lib


class Entropy(nn.Sequential):
    def __init__(self, patch_size, image_width, image_height):
        super().__init__()
        self.width = image_width
        self.height = image_height
        self.psize = patch_size
        self.patch_num = int(self.width * self.height / self.psize ** 2)
        self.hw = int(self.width // self.psize)

    def forward(self, inputs):
        x = _lib.require_cuda_f32(inputs, "inputs")              # raises on CPU tensors: no silent fallback
        B, C, H, W = x.shape
        if C != 3:
            raise ValueError("Entropy expects RGB images [B, 3, H, W]")
        if self.psize != 16:
            return _entropy_ops(x, self.psize)
        out = torch.empty((B, H // 16, W // 16), dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        with _lib.on_device(x.device):
            _lib.check(_lib.lib.dvq_entropy_map_f32(x.data_ptr(), B, H, W, 16, out.data_ptr(),
                                                    _lib.stream_ptr(x.device)), "dvq_entropy_map_f32")
        return out


def _entropy_ops(x, patch, chunk=8):
    """patch sizes the fused kernel does not cover: gray -> patches -> 32-bin Gaussian KDE -> entropy as device tensor
    ops, a few images at a time (the [b*P, patch^2, 32] intermediate is what the fused kernel avoids)"""
    B, _, H, W = x.shape
    bins = torch.linspace(0, 1, 32, device=x.device)
    outs = []
    for s in range(0, B, chunk):
        xs = x[s:s + chunk]
        gray = 0.2989 * xs[:, 0] + 0.5870 * xs[:, 1] + 0.1140 * xs[:, 2]
        b = gray.shape[0]
        p = gray.reshape(b, H // patch, patch, W // patch, patch).permute(0, 1, 3, 2, 4).reshape(-1, patch * patch)
        pdf = torch.exp(-0.5 * ((p.unsqueeze(2) - bins) / 0.01) ** 2).mean(1)
        pdf = pdf / (pdf.sum(1, keepdim=True) + 1e-40) + 1e-40
        outs.append((-(pdf * torch.log(pdf)).sum(1)).reshape(b, H // patch, W // patch))
    return torch.cat(outs, 0) if outs else x.new_empty((0, H // patch, W // patch))


def calibrate_thresholds(batches, patch_size=16):
    """The offline table DualGrainFixedEntropyRouter reads (reference scripts/tools/calculate_entropy_thresholds.py:
    92-110): patch entropies of every image of an iterable of [B, 3, H, W] GPU batches, sorted; threshold "k"
    (k = 1 .. 99) = sorted[(size * k) // 100].  Returns the dict the reference dumps as JSON (keys are strings).
    The entropy map is the fused kernel; sorting stays on the GPU."""
    vals = []
    ent = None
    for x in batches:
        if ent is None:
            ent = Entropy(patch_size, x.shape[-1], x.shape[-2])
        vals.append(ent(x).reshape(-1))
    if not vals:
        raise ValueError("calibrate_thresholds: no images")
    allv, _ = torch.sort(torch.cat(vals))
    size = allv.numel()
    idx = torch.tensor([(size * (i + 1)) // 100 for i in range(99)], device=allv.device)
    picked = allv[idx].cpu().tolist()
    return {str(i + 1): float(picked[i]) for i in range(99)}
