"""Golden vectors for the autograd and training-mode paths of the quantizers (BUILD container only;
imports the reference read-only):

  vq2_train_grad    VectorQuantize2 in train mode, z requires grad, restart_unused_codes=True with
                    torch.randperm replaced by a fixed permutation (reversed arange) so the dead-code
                    restart is reproducible: codes, loss, z.grad, EMA buffers, codebook after the step
                    (quantize2_mask.py:57-115,157-191)
  vqgan_grad        VectorQuantizer2 legacy=True / False: z.grad and embedding.weight.grad
                    (quantize_vqgan.py:271-312)
  vq2_soft_codes    VectorQuantize2.get_soft_codes at temp 0.7 (quantize2_mask.py:193-205)
  vqgan_remap       VectorQuantizer2.remap_to_used / unmap_to_all with unknown_index "extra" and an integer
                    (quantize_vqgan.py:247-268); the remap file is a tiny .npy written to a temp dir

Usage: python oracle/gen_golden_grad.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import crc, save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402


def fixed_randperm(n, device=None, **_kw):
    return torch.arange(n - 1, -1, -1, device=device)


def train_grad():
    VQ2, _ = refimport.quantizers()
    K, D, B, H, W = 64, 256, 2, 8, 8
    E = synth.codebook_trained(K, D, seed=7101)
    z = synth.z_tokens(E, B, H, W, 7102)
    mask = np.where(synth.bernoulli(7103, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
    gw = synth.normal(7104, z.shape)                       # upstream gradient on x_q
    m = VQ2(K, D, restart_unused_codes=True)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.codebook.embed_ema.copy_(torch.from_numpy(E))
    m.train()
    zt = torch.from_numpy(z).requires_grad_(True)
    real = torch.randperm
    torch.randperm = fixed_randperm
    try:
        xq, loss, (_, _, codes) = m(zt, codebook_mask=torch.from_numpy(mask))
    finally:
        torch.randperm = real
    ((xq * torch.from_numpy(gw)).sum() + 3.0 * loss).backward()
    save("vq2_train_grad", K=K, D=D, B=B, H=H, W=W, z_crc=crc(z), cb_crc=crc(E), mask_crc=crc(mask), gw_crc=crc(gw),
         codes=codes.numpy().astype(np.int16), loss=np.float32(loss.item()), z_grad=zt.grad.numpy(),
         cluster_size_ema=m.codebook.cluster_size_ema.numpy(), embed_ema=m.codebook.embed_ema.numpy(),
         weight_after=m.codebook.weight.detach().numpy()[:K])


def vqgan_grad():
    _, VQG = refimport.quantizers()
    K, D, B, H, W = 128, 256, 2, 8, 8
    E = synth.codebook_default_init(K, D, seed=7205)
    z = synth.z_tokens(synth.codebook_trained(K, D), B, H, W, 7202) * np.float32(0.002)
    gw = synth.normal(7204, z.shape)
    out = dict(K=K, D=D, B=B, H=H, W=W, z_crc=crc(z), cb_crc=crc(E), gw_crc=crc(gw))
    for legacy in (False, True):
        m = VQG(K, D, beta=0.25, legacy=legacy)
        m.embedding.weight.data.copy_(torch.from_numpy(E))
        zt = torch.from_numpy(z).requires_grad_(True)
        zq, loss, (_, _, idx) = m(zt)
        ((zq * torch.from_numpy(gw)).sum() + 5.0 * loss).backward()
        s = "_legacy%d" % int(legacy)
        out["codes" + s] = idx.numpy().astype(np.int16)
        out["loss" + s] = np.float32(loss.item())
        out["z_grad" + s] = zt.grad.numpy()
        out["w_grad" + s] = m.embedding.weight.grad.numpy()
    save("vqgan_grad", **out)


def soft_codes():
    VQ2, _ = refimport.quantizers()
    K, D = 48, 256
    E = synth.codebook_trained(K, D, seed=7301)
    x = synth.z_tokens(E, 1, 4, 6, 7302).transpose(0, 2, 3, 1).copy()        # channel-last [1, 4, 6, D]
    m = VQ2(K, D).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    soft, code = m.get_soft_codes(torch.from_numpy(x), temp=0.7, stochastic=False)
    d = m.codebook.compute_distances(torch.from_numpy(x))
    save("vq2_soft_codes", K=K, D=D, x=x, cb_crc=crc(E), soft=soft.numpy(), code=code.numpy().astype(np.int16),
         dist=d.numpy(), temp=np.float32(0.7))


def remap():
    _, VQG = refimport.quantizers()
    K, D = 32, 64
    used = np.array([3, 7, 7, 0, 19, 31, 4], dtype=np.int64)               # a duplicate: first match wins
    inds = np.array([[0, 3, 5, 7, 31], [19, 4, 30, 7, 1]], dtype=np.int64)
    out = dict(K=K, D=D, used=used, inds=inds)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "used.npy")
        np.save(path, used)
        for tag, unk in (("extra", "extra"), ("int", 2)):
            m = VQG(K, D, beta=0.25, remap=path, unknown_index=unk)
            new = m.remap_to_used(torch.from_numpy(inds.copy()))
            out["to_used_" + tag] = new.numpy()
            out["re_embed_" + tag] = np.int64(m.re_embed)
            back = m.unmap_to_all(new.clone())
            out["to_all_" + tag] = back.numpy()
    save("vqgan_remap", **out)


if __name__ == "__main__":
    train_grad()
    vqgan_grad()
    soft_codes()
    remap()
