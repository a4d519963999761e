"""Timing + error of the patch-entropy kernel (dvq_entropy_map_f32) at B = 64 and B = 256 (BASELINE configs[2]'s batch):
HIP events over blocks of 100 back-to-back launches, the MEDIAN of 8 blocks after 2 warm-up blocks (the first blocks after idle
run ~15 % slower: clock ramp), algorithmic bytes (the image once + the map) against 8 TB/s, max |dH| vs the reference's op
sequence on the CPU for a sample.  python tools/entropy_time.py > profiles/r06_entropy_time.json"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.entropy import Entropy
from oracle.entropy_torch import entropy_map as ref
dev = torch.device("cuda:0")
out = {}
base = torch.from_numpy(synth.images_flat_noise(5000, 32)[0]).to(dev)
ent = Entropy(16, 256, 256).to(dev)


def timed(img, blocks=10, n=100):
    ms = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ent(img)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / n)
    return float(np.median(ms[2:])), [round(m, 5) for m in ms]


for B in (64, 256):
    img = torch.cat([torch.roll(base, 16 * k, -1) for k in range(B // 32)], 0).contiguous()
    ms, blocks = timed(img)
    nbytes = img.numel() * 4 + B * 256 * 4
    got = ent(img[:8]).cpu()
    want = ref(img[:8].cpu())
    out["B%d" % B] = {"ms": ms, "ms_blocks": blocks, "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000,
                      "max_abs_err_8_images": float((got - want).abs().max())}
# natural-image-like input: smooth gradients in [-1, 1] + a little noise
yy, xx = np.meshgrid(np.linspace(-1, 1, 256, dtype=np.float32), np.linspace(-1, 1, 256, dtype=np.float32), indexing="ij")
nat = np.stack([yy, xx, 0.5 * (xx + yy)])[None].repeat(64, 0) + 0.02 * synth.normal(77, (64, 3, 256, 256))
nat = torch.from_numpy(nat.astype(np.float32)).to(dev)
ms, blocks = timed(nat)
out["smooth_B64"] = {"ms": ms, "ms_blocks": blocks, "max_abs_err_8_images": float((ent(nat[:8]).cpu() - ref(nat[:8].cpu())).abs().max())}
print(json.dumps(out, indent=1))
