"""Patch-entropy map that feeds the fixed-entropy router (BASELINE configs[2]).

Mirrors `Entropy` of the reference (models/stage1_dynamic/dqvae_dual_entropy.py:13-63): grayscale,
non-overlapping patch x patch unfold, 32-bin Gaussian-KDE histogram over [0, 1] (sigma 0.01), entropy
-sum p ln p with the reference's 1e-40 epsilons (an fp32 SUBNORMAL: denormals must stay enabled, which
is PyTorch-ROCm's and hipcc's default).  Transcendental fp32 math: parity to 1e-5, grain maps equal
away from the threshold.

On the GPU (patch 16, fp32) the forward is ONE fused kernel, `dvq_entropy_map_f32` (SURVEY.md section 8
row f3): the image is read once, nothing is materialised.  `fused=False` keeps the reference's tensor
op sequence (CPU tensors, other patch sizes); its [B*P, patch^2, 32] intermediate (2.1 GB at B = 256 in
the reference) is bounded by processing `chunk` images at a time.
"""
import torch
from torch import nn

from . import _lib


class Entropy(nn.Sequential):
    def __init__(self, patch_size, image_width, image_height, chunk=32, fused=True):
        super().__init__()
        self.width = image_width
        self.height = image_height
        self.psize = patch_size
        self.patch_num = int(self.width * self.height / self.psize ** 2)
        self.hw = int(self.width // self.psize)
        self.unfold = torch.nn.Unfold(kernel_size=(self.psize, self.psize), stride=self.psize)
        self.chunk = chunk
        self.fused = fused

    def entropy(self, values, bins, sigma, batch):
        epsilon = 1e-40
        values = values.unsqueeze(2)
        residuals = values - bins.unsqueeze(0).unsqueeze(0)
        kernel_values = torch.exp(-0.5 * (residuals / sigma).pow(2))
        pdf = torch.mean(kernel_values, dim=1)
        normalization = torch.sum(pdf, dim=1).unsqueeze(1) + epsilon
        pdf = pdf / normalization + epsilon
        entropy = -torch.sum(pdf * torch.log(pdf), dim=1)
        return entropy.reshape(batch, self.hw, self.hw)

    def forward(self, inputs):
        if self.fused and self.psize == 16:
            x = _lib.require_cuda_f32(inputs, "inputs")              # raises on CPU tensors: no silent fallback
            B, C, H, W = x.shape
            if C != 3:
                raise ValueError("Entropy expects RGB images [B, 3, H, W]")
            out = torch.empty((B, H // 16, W // 16), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib.dvq_entropy_map_f32(x.data_ptr(), B, H, W, 16, out.data_ptr(),
                                                        _lib.stream_ptr(x.device)), "dvq_entropy_map_f32")
            return out
        outs = []
        bins = torch.linspace(0, 1, 32).to(device=inputs.device)
        sigma = torch.tensor(0.01, device=inputs.device)
        for s in range(0, inputs.shape[0], self.chunk):
            x = inputs[s:s + self.chunk]
            gray = 0.2989 * x[:, 0:1, :, :] + 0.5870 * x[:, 1:2, :, :] + 0.1140 * x[:, 2:, :, :]
            u = self.unfold(gray).transpose(1, 2)
            u = torch.reshape(u.unsqueeze(2), (u.shape[0] * self.patch_num, u.shape[2]))
            outs.append(self.entropy(u, bins, sigma, x.shape[0]))
        return torch.cat(outs, 0)
