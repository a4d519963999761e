"""Golden CRCs at kernel-DISPATCH size (BUILD container only; imports the reference read-only):

  vq2_K16384_B128_crc   VectorQuantize2, K = 16384, D = 256, 32x32, B = 128 (131072 tokens): the size at
                        which DVQ_MODE_FILTER takes the two-blocks-per-wave pass-1 kernel and the 8-slice
                        resolver (quantize2_mask.py:157-191).  The reference's dense [N, K] distance
                        matrix would be 8.6 GB, so it is run 16 images at a time (tokens are independent).

  vq2_K16384_B512_crc   the same at BASELINE configs[4]'s full size, B = 512 (524288 tokens; its first 128 images are
                        the images of the B = 128 fixture).

Usage: python oracle/gen_golden_dispatch.py [B ...]        (default: 128)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import crc, per_image_crc, save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402

def generate(B):
    VQ2, _ = refimport.quantizers()
    H, W, K, D, seed = 32, 32, 16384, 256, 2605
    E = synth.codebook_trained(K, D)
    z = synth.z_tokens(E, B, H, W, seed)
    mask = np.where(synth.bernoulli(seed + 1, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
    m = VQ2(K, D).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    codes, zq_crc, sq = [], [], 0.0
    with torch.no_grad():
        for s in range(0, B, 16):
            xq, loss, (_, _, c) = m(torch.from_numpy(z[s:s + 16]), codebook_mask=torch.from_numpy(mask[s:s + 16]))
            codes.append(c.numpy())
            zq_crc.append(per_image_crc(xq.numpy()))
            sq += float(loss) / 1.25 * (16 * H * W * D)               # loss = 1.25 * mean over the chunk
    codes = np.concatenate(codes, 0)
    save("vq2_K16384_B%d_crc" % B, cls="VectorQuantize2", B=B, H=H, W=W, K=K, D=D, cb_kind="trained", seed=seed, masked=1,
         beta=np.float32(0.25), z_crc=crc(z), cb_crc=crc(E), mask_crc=crc(mask), codes_crc=per_image_crc(codes),
         zq_crc=np.concatenate(zq_crc, 0), codes_image0=codes[0].astype(np.int16),
         loss=np.float32(1.25 * sq / (B * H * W * D)))


if __name__ == "__main__":
    for b in ([int(x) for x in sys.argv[1:]] or [128]):
        generate(b)
