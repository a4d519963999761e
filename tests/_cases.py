"""Rebuild the seeded inputs of a golden fixture (the recipe in oracle/gen_golden.py)."""
import os
import zlib

import numpy as np

from dynamicvectorquantization_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def per_image_crc(a):
    return np.array([crc(a[i]) for i in range(a.shape[0])], dtype=np.uint32)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def vq2_inputs(g):
    """-> (z [B,D,H,W], E [K,D], mask or None) for a VectorQuantize2 fixture, CRC-checked."""
    B, H, W, K, D, seed = (int(g[k]) for k in ("B", "H", "W", "K", "D", "seed"))
    kind = str(g["cb_kind"])
    E = synth.codebook_trained(K, D) if kind == "trained" else synth.codebook_default_init(K, D)
    z = synth.z_tokens(E, B, H, W, seed)
    mask = None
    if int(g["masked"]):
        mask = np.where(synth.bernoulli(seed + 1, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
        assert crc(mask) == g["mask_crc"]
    assert crc(z) == g["z_crc"] and crc(E) == g["cb_crc"], "synthetic inputs do not regenerate bit-identically"
    return z, E, mask


def vqgan_inputs(g):
    B, H, W, K, D, seed = (int(g[k]) for k in ("B", "H", "W", "K", "D", "seed"))
    E = synth.codebook_default_init(K, D, seed=seed + 5)
    z = synth.z_tokens(synth.codebook_trained(K, D), B, H, W, seed) * np.float32(0.002)
    assert crc(z) == g["z_crc"] and crc(E) == g["cb_crc"]
    return z, E


def route_dual_inputs(g):
    B, C = int(g["B"]), int(g["C"])
    hf = synth.features(3002, B, C, 32, 32)
    hc = synth.features(3012, B, C, 16, 16)
    gate = synth.grain_gate_dual(4002, B, 16, 16)
    assert crc(hf) == g["hf_crc"] and crc(hc) == g["hc_crc"] and crc(gate) == g["gate_crc"]
    return gate, hc, hf


def route_triple_inputs(g):
    B, C = int(g["B"]), int(g["C"])
    hf = synth.features(3004, B, C, 32, 32)
    hm = synth.features(3014, B, C, 16, 16)
    hc = synth.features(3024, B, C, 8, 8)
    assert crc(hf) == g["hf_crc"] and crc(hm) == g["hm_crc"] and crc(hc) == g["hc_crc"]
    return g["logits"], hc, hm, hf


VQ2_FULL = ["vq2_cfg2_B4", "vq2_tiestress_B2", "vq2_nomask_B2", "vq2_16x16_B2", "vq2_K16384_B2", "vq2_D64_B2", "vq2_D128_B2"]
VQ2_CRC = ["vq2_cfg2_B64_crc", "vq2_cfg3_B256_crc"]
VQGAN = ["vqgan_cfg1_B4", "vqgan_legacy_sane_B2"]


def loss_close(a, b, rel=1e-5):
    """loss parity: 1e-5 relative (north_star tolerance); NaN == NaN, inf == inf."""
    a, b = float(a), float(b)
    if np.isnan(a) or np.isnan(b):
        return np.isnan(a) and np.isnan(b)
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= rel * abs(b)
