"""Golden vector for one training-mode step of VectorQuantize2 (row f2), BUILD container only:
EMA cluster statistics and the re-normalised codebook after the step, restart_unused_codes=False
(the restart draws torch.randperm: no RNG parity possible).  Usage: python oracle/gen_golden_train.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import crc, save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402

if __name__ == "__main__":
    VQ2, _ = refimport.quantizers()
    K, D, B, H, W = 64, 256, 2, 16, 16
    E = synth.codebook_trained(K, D, seed=7001)
    z = synth.z_tokens(E, B, H, W, 7002)
    mask = np.where(synth.bernoulli(7003, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
    m = VQ2(K, D, restart_unused_codes=False)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.codebook.embed_ema.copy_(torch.from_numpy(E))
    m.train()
    xq, loss, (_, _, codes) = m(torch.from_numpy(z), codebook_mask=torch.from_numpy(mask))
    save("vq2_train_step", K=K, D=D, B=B, H=H, W=W, z_crc=crc(z), cb_crc=crc(E),
         codes=codes.numpy().astype(np.int16), loss=np.float32(loss.item()),
         cluster_size_ema=m.codebook.cluster_size_ema.numpy(), embed_ema=m.codebook.embed_ema.numpy(),
         weight_after=m.codebook.weight.detach().numpy())
