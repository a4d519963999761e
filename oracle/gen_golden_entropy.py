"""Golden vector for the patch-entropy map (row a12), BUILD container only: the reference `Entropy`
module on the flat/noise synthetic images of SURVEY.md section 8d, plus the grain map the fixed-entropy
router derives from it.  Usage: python oracle/gen_golden_entropy.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import crc, save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402

refimport.setup()
# the module imports names from files that need torchvision / PIL at import time: provide them lazily
import types  # noqa: E402
for name in ("modules.dynamic_modules.utils", "models.stage1.utils", "models.stage2.utils"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.draw_dual_grain_256res = m.draw_dual_grain_256res_color = None
        m.Scheduler_LinearWarmup = m.Scheduler_LinearWarmup_CosineDecay = None
        m.disabled_train = lambda self, mode=True: self
        sys.modules[name] = m
from models.stage1_dynamic.dqvae_dual_entropy import Entropy  # noqa: E402

if __name__ == "__main__":
    img, noisy = synth.images_flat_noise(5000, 2)
    with torch.no_grad():
        ent = Entropy(16, 256, 256)(torch.from_numpy(img)).numpy()
    thr = 1.6777750253677368
    assert np.array_equal(ent > thr, noisy), "grain map must equal the noise mask on this data"
    save("entropy_map_B2", seed=5000, img_crc=crc(img), entropy=ent.astype(np.float32), noisy=noisy.astype(np.int8),
         thr=np.float64(thr))
    print("entropy range flat [%.3g, %.3g] noise [%.3g, %.3g]" % (ent[~noisy].min(), ent[~noisy].max(),
                                                                 ent[noisy].min(), ent[noisy].max()))
