"""Golden vectors for the list-input quantizer (BUILD container only; imports the reference read-only):

  vq2_list_eval    quantize2_list.VectorQuantize2 in eval mode on three items of different token counts
                   ([5, 7, D], [33, D], [2, 3, 4, D] channel-last): codes per item, loss, CRC of every x_q item
                   (quantize2_list.py:153-170)
  vq2_list_train   the same module in train mode (EMA update after every item: item i + 1 sees the codebook item i
                   left behind), inputs requiring grad, torch.randperm pinned: codes, loss, input gradients, EMA
                   buffers and the codebook after the step

Usage: python oracle/gen_golden_list.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import crc, save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402

SHAPES = [(5, 7), (33,), (2, 3, 4)]


def items(E, seed):
    """channel-last token tensors drawn like the NCHW latents of the other fixtures"""
    out = []
    for i, shp in enumerate(SHAPES):
        n = int(np.prod(shp))
        z = synth.z_tokens(E, 1, n, 1, seed + i)               # [1, D, n, 1]
        out.append(np.ascontiguousarray(z[0, :, :, 0].T).reshape(shp + (E.shape[1],)))
    return out


def ref_class():
    refimport.setup()
    from modules.vector_quantization.quantize2_list import VectorQuantize2
    return VectorQuantize2


def eval_case():
    VQL = ref_class()
    K, D = 96, 256
    E = synth.codebook_trained(K, D, seed=7301)
    xs = items(E, 7310)
    m = VQL(K, D)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.eval()
    with torch.no_grad():
        xq, loss, (_, _, codes) = m([torch.from_numpy(x) for x in xs])
    kw = {"codes%d" % i: c.numpy().astype(np.int16) for i, c in enumerate(codes)}
    kw.update({"xq_crc%d" % i: crc(q.numpy()) for i, q in enumerate(xq)})
    save("vq2_list_eval", K=K, D=D, cb_crc=crc(E), x_crc=np.array([crc(x) for x in xs], dtype=np.uint32),
         loss=np.float32(loss.item()), **kw)


BIG_SHAPES = [(16, 32, 32), (3001,), (40, 50, 13)]            # 16384 + 3001 + 26000 = 45385 tokens: the filter path's dispatch size


def big_items(E, seed):
    out = []
    for i, shp in enumerate(BIG_SHAPES):
        n = int(np.prod(shp))
        z = synth.z_tokens(E, 1, n, 1, seed + i)               # [1, D, n, 1]
        out.append(np.ascontiguousarray(z[0, :, :, 0].T).reshape(shp + (E.shape[1],)))
    return out


def eval_big_case():
    """the list quantizer at dispatch size (K = 1024, 45 385 tokens in three items): per item the CRC of all codes, the first 512
    codes in full and the CRC of x_q (quantize2_list.py:153-170) -- what the row-major pass-1 form is pinned by"""
    VQL = ref_class()
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    xs = big_items(E, 7510)
    m = VQL(K, D)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.eval()
    with torch.no_grad():
        xq, loss, (_, _, codes) = m([torch.from_numpy(x) for x in xs])
    kw = {"codes_head%d" % i: c.numpy().reshape(-1)[:512].astype(np.int16) for i, c in enumerate(codes)}
    kw.update({"codes_crc%d" % i: crc(c.numpy().astype(np.int64)) for i, c in enumerate(codes)})
    kw.update({"xq_crc%d" % i: crc(q.numpy()) for i, q in enumerate(xq)})
    save("vq2_list_eval_big", K=K, D=D, cb_crc=crc(E), x_crc=np.array([crc(x) for x in xs], dtype=np.uint32),
         loss=np.float32(loss.item()), **kw)


def train_case():
    VQL = ref_class()
    K, D = 16, 256                                             # every item has >= K tokens: no noise-tiling (unseeded RNG) in the restart
    E = synth.codebook_trained(K, D, seed=7401)
    xs = items(E, 7410)
    gws = [synth.normal(7420 + i, x.shape) for i, x in enumerate(xs)]
    m = VQL(K, D, restart_unused_codes=True)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.codebook.embed_ema.copy_(torch.from_numpy(E))
    m.train()
    xt = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    real = torch.randperm
    torch.randperm = lambda n, device=None, **kw: torch.arange(n - 1, -1, -1, device=device)
    try:
        xq, loss, (_, _, codes) = m(xt)
    finally:
        torch.randperm = real
    (sum((q * torch.from_numpy(g)).sum() for q, g in zip(xq, gws)) + 2.0 * loss).backward()
    kw = {"codes%d" % i: c.numpy().astype(np.int16) for i, c in enumerate(codes)}
    kw.update({"grad%d" % i: x.grad.numpy() for i, x in enumerate(xt)})
    save("vq2_list_train", K=K, D=D, cb_crc=crc(E), x_crc=np.array([crc(x) for x in xs], dtype=np.uint32),
         loss=np.float32(loss.item()), cluster_size_ema=m.codebook.cluster_size_ema.numpy(),
         embed_ema=m.codebook.embed_ema.numpy(), weight_after=m.codebook.weight.detach().numpy()[:K], **kw)


if __name__ == "__main__":
    eval_case()
    eval_big_case()
    train_case()
