"""Torch-op restatement of the reference's patch-entropy map (models/stage1_dynamic/dqvae_dual_entropy.py:13-63)
-- TEST INFRASTRUCTURE (comparator for the fused kernel in tests/ and tools/; runs on CPU or GPU tensors).
gray = .2989 R + .5870 G + .1140 B; 16x16 patches; p_k ~ mean_i exp(-((v_i - bin_k) / sigma)^2 / 2) over 32
bins on [0, 1], sigma = 0.01; normalised with the reference's 1e-40 epsilons; H = -sum p ln p."""
import torch


def entropy_map(images, patch=16, chunk=8):
    B, _, H, W = images.shape
    bins = torch.linspace(0, 1, 32, device=images.device)
    outs = []
    for s in range(0, B, chunk):
        x = images[s:s + chunk]
        gray = 0.2989 * x[:, 0] + 0.5870 * x[:, 1] + 0.1140 * x[:, 2]                       # [b, H, W]
        b = gray.shape[0]
        p = gray.reshape(b, H // patch, patch, W // patch, patch).permute(0, 1, 3, 2, 4).reshape(-1, patch * patch)
        k = torch.exp(-0.5 * ((p.unsqueeze(2) - bins) / 0.01) ** 2)                         # [b*P, 256, 32]
        pdf = k.mean(1)
        pdf = pdf / (pdf.sum(1, keepdim=True) + 1e-40) + 1e-40
        outs.append((-(pdf * torch.log(pdf)).sum(1)).reshape(b, H // patch, W // patch))
    return torch.cat(outs, 0)
