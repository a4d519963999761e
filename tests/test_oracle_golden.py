"""CPU: the oracle (oracle/dvq_oracle.c) against every golden vector captured from the reference."""
import numpy as np
import pytest

from tests import _cases as C


@pytest.mark.parametrize("name", C.VQ2_FULL + C.VQ2_CRC[:1])
def test_vq2_oracle_matches_reference(oracle_mod, name):
    g = C.load(name)
    z, E, mask = C.vq2_inputs(g)
    o = oracle_mod.vq_assign_nchw(z, E, mask)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    codes = o["codes"].reshape(B, H, W)
    if "codes" in g:
        assert np.array_equal(codes, g["codes"].astype(np.int64))
    else:
        assert np.array_equal(codes[0], g["codes_image0"].astype(np.int64))
    assert np.array_equal(C.per_image_crc(codes), g["codes_crc"])
    assert np.array_equal(C.per_image_crc(o["zq"]), g["zq_crc"])          # z_q bit-exact
    loss = oracle_mod.vq_loss(o["sqerr"], o["numel"], float(g["beta"]))
    assert C.loss_close(loss, g["loss"])


def test_vq2_oracle_matches_reference_k16384_dispatch_size(oracle_mod):
    """the K = 16384, B = 128 fixture (the size that dispatches to the wide pass-1 kernel on the GPU): a sample of
    8 whole images keeps the CPU suite short"""
    g = C.load("vq2_K16384_B128_crc")
    z, E, mask = C.vq2_inputs(g)
    sel = np.arange(0, int(g["B"]), 16)
    o = oracle_mod.vq_assign_nchw(z[sel], E, mask[sel])
    codes = o["codes"].reshape(len(sel), 32, 32)
    assert np.array_equal(codes[0], g["codes_image0"].astype(np.int64))
    assert np.array_equal(C.per_image_crc(codes), g["codes_crc"][sel])
    assert np.array_equal(C.per_image_crc(o["zq"]), g["zq_crc"][sel])


@pytest.mark.parametrize("name", C.VQGAN)
def test_vqgan_oracle_matches_reference(oracle_mod, name):
    g = C.load(name)
    z, E = C.vqgan_inputs(g)
    o = oracle_mod.vq_assign_nchw(z, E, None)
    assert np.array_equal(o["codes"].reshape(g["idx_shape"]), g["codes"].astype(np.int64))
    assert np.array_equal(C.per_image_crc(o["zq"]), g["zq_crc"])
    loss = oracle_mod.vq_loss(o["sqerr"], o["numel"], float(g["beta"]), legacy=bool(g["legacy"]))
    assert C.loss_close(loss, g["loss"])


def test_tiny_full_special_values(oracle_mod):
    """duplicates (first index wins), NaN / +-inf tokens, zero tokens -- explicit inputs/outputs"""
    g = C.load("vq2_tiny_full")
    o = oracle_mod.vq_assign_nchw(g["z"], g["codebook"], g["mask"])
    assert np.array_equal(o["codes"].reshape(g["codes"].shape), g["codes"])
    assert np.array_equal(o["zq"], g["zq"], equal_nan=True)
    assert g["codes"][0, 0, 0] == 3 and g["codes"][0, 0, 1] == 0      # tie -> first; NaN row -> 0
    o1 = oracle_mod.vq_assign_nchw(g["z"][1:], g["codebook"], g["mask"][1:])
    loss = oracle_mod.vq_loss(o1["sqerr"], o1["numel"], 0.25)
    assert C.loss_close(loss, g["loss_img1_flatmask"])
    o0 = oracle_mod.vq_assign_nchw(g["z"], g["codebook"], g["mask"])
    assert C.loss_close(oracle_mod.vq_loss(o0["sqerr"], o0["numel"], 0.25), g["loss"])


def test_entropy_gate(oracle_mod, golden_dir):
    import json
    import os
    from dynamicvectorquantization_amd import synth
    g = C.load("entropy_router")
    ent = synth.entropy_map(int(g["seed"]), int(g["B"]), 16, 16)
    assert C.crc(ent) == g["ent_crc"]
    with open(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json")) as f:
        table = json.load(f)
    assert table["50"] == 1.6777750253677368
    for ratio in (0.5, 0.3, 0.85):
        key = "r%02d" % int(ratio * 100)
        thr = table[str(int(100 - ratio * 100))]
        assert thr == float(g["thr_" + key])
        e2 = ent.copy()
        e2[0, 0, 0] = np.float32(thr)                                  # == thr -> coarse
        gate = oracle_mod.entropy_gate(e2, thr)
        assert np.array_equal(gate, g["gate_" + key].astype(np.int64))
        assert gate[0, 0, 0, 0] == 1 and gate[0, 0, 0, 1] == 0


def test_route_select_dual(oracle_mod):
    g = C.load("route_dual_B2")
    gate, hc, hf = C.route_dual_inputs(g)
    o = oracle_mod.route_select_dual(gate, hc, hf)
    assert np.array_equal(o["indices"], g["indices"].astype(np.int64))
    assert np.array_equal(o["codebook_mask"], g["cmask"]) and o["codebook_mask"].dtype == np.float32
    assert np.array_equal(C.per_image_crc(o["h_dual"]), g["h_crc"])
    o = oracle_mod.route_select_dual(g["logits"], hc, hf)               # f32 logits with ties and NaN
    assert np.array_equal(o["indices"], g["indices_logits"].astype(np.int64))
    assert np.array_equal(o["codebook_mask"], g["cmask_logits"])
    assert np.array_equal(C.per_image_crc(o["h_dual"]), g["h_crc_logits"])


def test_route_select_triple(oracle_mod):
    g = C.load("route_triple_B2")
    lg, hc, hm, hf = C.route_triple_inputs(g)
    o = oracle_mod.route_select_triple(lg, hc, hm, hf)
    assert np.array_equal(o["indices"], g["indices"].astype(np.int64))
    assert np.array_equal(o["codebook_mask"], g["cmask"])
    assert np.array_equal(C.per_image_crc(o["h_triple"]), g["h_crc"])
    assert set(np.unique(o["codebook_mask"])) <= {0.0625, 0.25, 1.0}


def test_oracle_properties(oracle_mod):
    """idempotence: quantising z_q's winning rows again returns the same codes with distance ~0;
    coarse 2x2 cells (bit-identical inputs) get identical codes."""
    from dynamicvectorquantization_amd import synth
    K, D = 256, 256
    E = synth.codebook_trained(K, D, seed=71)
    idx = synth.randint(72, (1, 16, 16), K)
    z = np.ascontiguousarray(E[idx].transpose(0, 3, 1, 2))
    o = oracle_mod.vq_assign_nchw(z, E, None, want_dmin=True)
    assert np.array_equal(o["codes"].reshape(1, 16, 16), idx)
    assert np.array_equal(o["zq"], z)
    zc = synth.normal(73, (1, D, 8, 8))
    zr = zc.repeat(2, axis=2).repeat(2, axis=3)
    o = oracle_mod.vq_assign_nchw(zr, E, None)
    c = o["codes"].reshape(16, 16)
    assert np.array_equal(c[0::2, 0::2], c[1::2, 1::2]) and np.array_equal(c[0::2, 0::2], c[0::2, 1::2])


@pytest.mark.parametrize("fixture", ["encode_dual_entropy_model_B1.npz", "encode_dual_entropy_model_B4.npz"])
def test_oracle_chain_against_the_reference_models_own_encode(oracle_mod, golden_dir, fixture):
    """the oracle's gate -> select -> (float64 conv, rounded to f32) -> assign chain against the golden captured from the reference's
    own `DualGrainVQModel.encode` on CPU (oracle/gen_golden_encode.py; round 4: B = 4 with mixed grains): grain map and gate
    bit-equal, codes equal up to near-ties of the conv's rounding (> 99.5 %), loss within 1e-4"""
    import json
    import os
    import zlib
    from dynamicvectorquantization_amd import synth
    g = np.load(os.path.join(golden_dir, fixture))
    crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    cw, cb = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0), synth.normal(9502, (D,), 0.0, 0.1)
    assert crc(E) == g["cb_crc"] and crc(cw) == g["conv_w_crc"] and crc(cb) == g["conv_b_crc"]
    meta = json.loads(str(g["meta"]))
    if "image_crc" in meta:                                   # the PIXELS regenerate from the seed (the GPU test starts from them)
        img, _ = synth.images_flat_noise(meta["seeds"]["image"], g["h_fine"].shape[0])
        assert int(crc(img)) == meta["image_crc"]
        assert 0.2 < float(g["fine_ratio"]) < 0.8            # mixed grains
    thr = json.load(open(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json")))["50"]
    og = oracle_mod.entropy_gate(g["x_entropy"], thr)
    osel = oracle_mod.route_select_dual(og, g["h_coarse"], g["h_fine"])
    assert np.array_equal(osel["indices"], g["grain"].astype(np.int64))
    assert np.array_equal(og.transpose(0, 3, 1, 2), g["gate"].astype(np.int64))
    h = (np.einsum("ok,bkhw->bohw", cw[:, :, 0, 0].astype(np.float64), osel["h_dual"].astype(np.float64))
         + cb.astype(np.float64)[None, :, None, None]).astype(np.float32)
    o = oracle_mod.vq_assign_nchw(h, E, osel["codebook_mask"])
    rate = float((o["codes"].reshape(-1) == g["codes"].astype(np.int64).reshape(-1)).mean())
    assert rate > 0.995, rate
    ol = float(oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    assert abs(ol - float(g["emb_loss"])) <= 1e-4 * abs(float(g["emb_loss"]))


def test_fold_identity_in_float64(oracle_mod):
    """host logic behind the conv-folded codebook (csrc/vq_fold.hip): with h = W x + b,
    argmin_j ||h - e_j||^2 = argmax_j [ x.(W^T e_j) + (b.e_j - ||e_j||^2 / 2) ]  -- checked in float64 on seeded data, and the
    oracle's fp32 chain on fp32(h) picks the same codes wherever the float64 margin is not a near-tie"""
    from dynamicvectorquantization_amd import synth
    D, K, n = 256, 512, 200
    E = synth.codebook_trained(K, D, seed=4401).astype(np.float64)
    Wc = synth.normal(4402, (D, D), 0.0, 1.0 / 16.0).astype(np.float64)
    b = synth.normal(4403, (D,), 0.0, 0.1).astype(np.float64)
    x = synth.normal(4404, (n, D)).astype(np.float64)
    h = x @ Wc.T + b
    d = ((h[:, None, :] - E[None]) ** 2).sum(-1)
    score = x @ (E @ Wc).T + (E @ b - 0.5 * (E ** 2).sum(1))[None]
    assert np.array_equal(d.argmin(1), score.argmax(1))
    assert np.allclose(-0.5 * (d - (h ** 2).sum(1, keepdims=True)), score, rtol=1e-10, atol=1e-9)
    o = oracle_mod.vq_assign_nchw(np.ascontiguousarray(h.astype(np.float32).T[None]), E.astype(np.float32), None)
    srt = np.sort(d, 1)
    clear = (srt[:, 1] - srt[:, 0]) > 1e-3
    assert clear.sum() > n // 2 and np.array_equal(o["codes"].reshape(-1)[clear], d.argmin(1)[clear])
