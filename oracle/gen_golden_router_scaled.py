"""Golden vectors for the feature routers with LARGE GroupNorm parameters (BUILD container only; imports the reference read-only).

The fused gate kernel writes its fp16 operand images with a power-of-two scale derived from the GroupNorm parameters (|w| sqrt(n) +
|b| can exceed the fp16 range); until round 5 that path was only compared with the package's own torch-op path.  Here the reference's
DualGrainFeatureRouter / TripleGrainFeatureRouter (modules/dynamic_modules/RouterDual.py:6-43, RouterTriple.py:6-56) run on CPU with the
seeded parameters of the other router fixtures, GroupNorm weights x 3000, biases x 200, and the hidden layer's weight / 3000 (so the logits
stay O(1)):  feature_router_{dual,triple}_scaled.npz = logits.   Usage: python oracle/gen_golden_router_scaled.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from oracle.gen_golden import save  # noqa: E402
from dynamicvectorquantization_amd import synth  # noqa: E402


def scaled_state(mod, seed):
    sd = {}
    for i, (k, v) in enumerate(mod.state_dict().items()):
        p = synth.seeded_param(seed, i, k, tuple(v.shape))
        if k.startswith("feature_norm") and k.endswith("weight"):
            p = p * np.float32(3000.0)
        elif k.startswith("feature_norm") and k.endswith("bias"):
            p = p * np.float32(200.0)
        elif k == "gate.0.weight":
            p = p * np.float32(1.0 / 3000.0)
        sd[k] = torch.from_numpy(np.ascontiguousarray(p))
    return sd


def main():
    DF, _, TF = refimport.routers()
    B, C = 2, 256
    r = DF(256, "group-32", "2layer-fc-SiLu").eval()
    r.load_state_dict(scaled_state(r, 6300))
    with torch.no_grad():
        lg = r(h_fine=torch.from_numpy(synth.features(3102, B, C, 32, 32)), h_coarse=torch.from_numpy(synth.features(3112, B, C, 16, 16)))
    save("feature_router_dual_scaled", B=B, seed=6300, logits=lg.numpy())
    r = TF(256, "group-32", "2layer-fc-ReLu").eval()
    r.load_state_dict(scaled_state(r, 6400))
    with torch.no_grad():
        lg = r(h_fine=torch.from_numpy(synth.features(3104, B, C, 32, 32)), h_median=torch.from_numpy(synth.features(3114, B, C, 16, 16)),
               h_coarse=torch.from_numpy(synth.features(3124, B, C, 8, 8)))
    save("feature_router_triple_scaled", B=B, seed=6400, logits=lg.numpy())


if __name__ == "__main__":
    main()
