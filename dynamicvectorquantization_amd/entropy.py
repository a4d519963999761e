"""Patch-entropy map that feeds the fixed-entropy router (BASELINE configs[2]).

Mirrors `Entropy` of the reference (models/stage1_dynamic/dqvae_dual_entropy.py:13-63): grayscale,
non-overlapping patch x patch unfold, 32-bin Gaussian-KDE histogram over [0, 1] (sigma 0.01), entropy
-sum p ln p with the reference's 1e-40 epsilons (an fp32 SUBNORMAL: denormals must stay enabled, which
is PyTorch-ROCm's and hipcc's default).  Runs as PyTorch-ROCm tensor ops (SURVEY.md section 8 row a12;
a fused kernel is row f3) -- transcendental fp32 math, parity to 1e-5, grain maps equal away from the
threshold.  The [B*P, patch^2, 32] intermediate of the reference (2.1 GB at B = 256) is bounded by
processing `chunk` images at a time.
"""
import torch
from torch import nn


class Entropy(nn.Sequential):
    def __init__(self, patch_size, image_width, image_height, chunk=32):
        super().__init__()
        self.width = image_width
        self.height = image_height
        self.psize = patch_size
        self.patch_num = int(self.width * self.height / self.psize ** 2)
        self.hw = int(self.width // self.psize)
        self.unfold = torch.nn.Unfold(kernel_size=(self.psize, self.psize), stride=self.psize)
        self.chunk = chunk

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
