"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs, against the golden vectors captured from the reference, and -- at BASELINE.json's
full sizes -- through size-independent properties.  Bar: code indices, grain indices, routed
features and z_q BIT-EXACT; loss within 1e-5 relative (north_star tolerance)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import _cases as C

pytestmark = pytest.mark.gpu

MODES = [0, 1]   # DVQ_MODE_EXACT, DVQ_MODE_FILTER -- must produce identical outputs


def _vq2(dev, E, mode):
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    K, D = E.shape
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    vq.assign_mode = mode
    return vq


def _run_vq2(dev, z, E, mask, mode):
    vq = _vq2(dev, E, mode)
    with torch.no_grad():
        xq, loss, (a, b, codes) = vq(torch.from_numpy(z).to(dev),
                                     codebook_mask=None if mask is None else torch.from_numpy(mask).to(dev))
    assert a is None and b is None and codes.dtype == torch.int64
    return xq.cpu().numpy(), float(loss), codes.cpu().numpy()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", C.VQ2_FULL + C.VQ2_CRC)
def test_vq2_golden(dev, name, mode):
    """reference golden vectors: cfg-2/3 shapes, tie-stress default-init codebook, K=16384, 16x16"""
    g = C.load(name)
    z, E, mask = C.vq2_inputs(g)
    xq, loss, codes = _run_vq2(dev, z, E, mask, mode)
    assert codes.shape == (int(g["B"]), int(g["H"]), int(g["W"]))
    if "codes" in g:
        assert np.array_equal(codes, g["codes"].astype(np.int64))
    else:
        assert np.array_equal(codes[0], g["codes_image0"].astype(np.int64))
    assert np.array_equal(C.per_image_crc(codes), g["codes_crc"])       # int64 codes, bit-exact
    assert np.array_equal(C.per_image_crc(xq), g["zq_crc"])             # z_q bit-exact
    assert C.loss_close(loss, g["loss"])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", C.VQGAN)
def test_vqgan_golden(dev, name, mode):
    """BASELINE configs[0]: fixed-granularity VectorQuantizer2, 16x16x256, K=1024, B=4"""
    from dynamicvectorquantization_amd.quantize import VectorQuantizer2
    g = C.load(name)
    z, E = C.vqgan_inputs(g)
    m = VectorQuantizer2(int(g["K"]), int(g["D"]), beta=0.25, legacy=bool(g["legacy"]),
                         sane_index_shape=bool(g["sane"])).to(dev).eval()
    m.embedding.weight.data.copy_(torch.from_numpy(E))
    m.assign_mode = mode
    with torch.no_grad():
        zq, loss, (p, me, idx) = m(torch.from_numpy(z).to(dev))
    assert p is None and me is None and idx.dtype == torch.int64
    assert tuple(idx.shape) == tuple(g["idx_shape"])
    assert np.array_equal(idx.cpu().numpy(), g["codes"].astype(np.int64))
    assert np.array_equal(C.per_image_crc(zq.cpu().numpy()), g["zq_crc"])
    assert C.loss_close(float(loss), g["loss"])
    with pytest.raises(AssertionError):
        m(torch.from_numpy(z).to(dev), temp=0.5)


@pytest.mark.parametrize("mode", MODES)
def test_special_values(dev, mode):
    """duplicated code rows (first index), NaN / +-inf tokens (NaN = minimum), zero and -0 tokens"""
    g = C.load("vq2_tiny_full")
    xq, loss, codes = _run_vq2(dev, g["z"], g["codebook"], g["mask"], mode)
    assert np.array_equal(codes, g["codes"])
    assert np.array_equal(xq, g["zq"], equal_nan=True)
    assert C.loss_close(loss, g["loss"])
    # 2-D mask branch (quantize2_mask.py:177) on image 1 only
    vq = _vq2(dev, g["codebook"], mode)
    with torch.no_grad():
        xq1, loss1, (_, _, c1) = vq(torch.from_numpy(g["z"][1:]).to(dev),
                                    codebook_mask=torch.from_numpy(g["mask"][1:, 0].reshape(1, 64)).to(dev))
    assert np.array_equal(c1.cpu().numpy(), g["codes_img1"]) and C.loss_close(float(loss1), g["loss_img1_flatmask"])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", [(1, 1, 1, 33), (3, 5, 7, 100), (2, 8, 8, 1000), (1, 3, 3, 31), (5, 1, 1, 64)])
def test_ragged_shapes_vs_oracle(dev, oracle_mod, shape, mode):
    """token counts that are not multiples of the 32/128-token tiles, K not a multiple of 32, K < 32"""
    from dynamicvectorquantization_amd import synth
    B, H, W, K = shape
    E = synth.codebook_trained(K, 256, seed=800 + K)
    z = synth.z_tokens(E, B, H, W, 810 + K)
    mask = np.where(synth.bernoulli(820 + K, (B, 1, H, W), 0.5), 1.0, 0.0625).astype(np.float32)
    xq, loss, codes = _run_vq2(dev, z, E, mask, mode)
    o = oracle_mod.vq_assign_nchw(z, E, mask)
    assert np.array_equal(codes.reshape(B, -1), o["codes"])
    assert np.array_equal(xq, o["zq"])
    assert C.loss_close(loss, oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("D", [64, 128])
def test_other_dims_vs_oracle(dev, oracle_mod, D, mode):
    from dynamicvectorquantization_amd import synth
    E = synth.codebook_trained(512, D, seed=900 + D)
    z = synth.z_tokens(E, 2, 16, 16, 910 + D)
    xq, loss, codes = _run_vq2(dev, z, E, None, mode)
    o = oracle_mod.vq_assign_nchw(z, E, None)
    assert np.array_equal(codes.reshape(2, -1), o["codes"]) and np.array_equal(xq, o["zq"])
    assert C.loss_close(loss, oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))


@pytest.mark.parametrize("mode", MODES)
def test_near_tie_adversarial(dev, oracle_mod, mode):
    """tokens placed (almost) exactly between two codes, duplicated / nearly duplicated codes,
    huge and tiny magnitudes: the cases where an approximate filter must fall back to the exact chain"""
    from dynamicvectorquantization_amd import synth
    K, D = 256, 256
    E = synth.codebook_trained(K, D, seed=61)
    E[100] = E[40]                                             # exact duplicate
    E[101] = E[41] * (1 + np.float32(2 ** -22))                # 1-ulp-scale near duplicate
    E[102:110] = E[50] + synth.normal(62, (8, D), 0, 1e-6)     # tight cluster
    z = synth.z_tokens(E, 2, 16, 16, 63)
    zt = z.transpose(0, 2, 3, 1)
    zt[0, 0, :, :] = ((E[0:16] + E[16:32]) * np.float32(0.5))           # midpoints
    zt[0, 1, :, :] = E[40:56]                                            # on a code (one duplicated)
    zt[0, 2, :, :] = E[96:112]
    zt[0, 3, :, :] = E[0:16] * np.float32(1e4)                           # large magnitude
    zt[0, 4, :, :] = E[0:16] * np.float32(1e-6)                          # tiny magnitude
    zt[0, 5, :, :] = E[0:16] * np.float32(3e38 / 1e4)                    # overflow territory
    z = np.ascontiguousarray(zt.transpose(0, 3, 1, 2))
    xq, loss, codes = _run_vq2(dev, z, E, None, mode)
    o = oracle_mod.vq_assign_nchw(z, E, None)
    assert np.array_equal(codes.reshape(2, -1), o["codes"])
    assert np.array_equal(xq, o["zq"], equal_nan=True)


def test_channel_last_and_sequence_layouts(dev, oracle_mod):
    """accept_image_fmap=False: channel_last [B, N, D] (flat tokens) and [B, D, N] inputs"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    K, D = 128, 256
    E = synth.codebook_trained(K, D, seed=66)
    z = synth.z_tokens(E, 2, 4, 8, 67)                        # [2, D, 4, 8]
    o = oracle_mod.vq_assign_nchw(z, E, None)
    zs = torch.from_numpy(z.reshape(2, D, 32)).to(dev)
    for kw, x in ((dict(accept_image_fmap=False, channel_last=True), zs.permute(0, 2, 1).contiguous()),
                  (dict(accept_image_fmap=False, channel_last=False), zs)):
        vq = VectorQuantize2(K, D, **kw).to(dev).eval()
        vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
        with torch.no_grad():
            xq, loss, (_, _, codes) = vq(x)
        assert tuple(xq.shape) == tuple(x.shape) and tuple(codes.shape) == (2, 32)
        assert np.array_equal(codes.cpu().numpy(), o["codes"])
        xq_cm = xq.permute(0, 2, 1) if kw["channel_last"] else xq
        assert np.array_equal(xq_cm.cpu().numpy().reshape(z.shape), o["zq"])
        assert C.loss_close(float(loss), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))


def test_codebook_entry_and_nearest(dev, oracle_mod):
    from dynamicvectorquantization_amd import synth
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    vq = _vq2(dev, E, 1)
    idx = torch.from_numpy(synth.randint(5, (2, 32, 32), K)).to(dev)
    zq = vq.get_codebook_entry(idx)                           # [B, H, W, D] like the reference
    assert tuple(zq.shape) == (2, 32, 32, D)
    assert np.array_equal(zq.cpu().numpy(), E[idx.cpu().numpy()])
    emb, ids = vq.codebook(zq)                                # VQEmbedding.forward on channel-last input
    assert np.array_equal(ids.cpu().numpy(), idx.cpu().numpy()) and np.array_equal(emb.cpu().numpy(), zq.cpu().numpy())
    # codebook edits are picked up (prep cache keyed on the tensor version)
    with torch.no_grad():
        vq.codebook.weight[:-1].mul_(0.5)
    z = synth.z_tokens(E, 1, 8, 8, 6)
    with torch.no_grad():
        _, _, (_, _, c) = vq(torch.from_numpy(z).to(dev))
    o = oracle_mod.vq_assign_nchw(z, E * np.float32(0.5), None)
    assert np.array_equal(c.cpu().numpy().reshape(1, -1), o["codes"])


def test_entropy_router_golden(dev, golden_dir):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    g = C.load("entropy_router")
    ent = synth.entropy_map(int(g["seed"]), int(g["B"]), 16, 16)
    js = os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json")
    for ratio in (0.5, 0.3, 0.85):
        r = DualGrainFixedEntropyRouter(js, ratio)
        e2 = ent.copy()
        e2[0, 0, 0] = np.float32(r.fine_grain_threshold)
        gate = r(entropy=torch.from_numpy(e2).to(dev))
        assert gate.dtype == torch.int64 and tuple(gate.shape) == (int(g["B"]), 16, 16, 2)
        assert np.array_equal(gate.cpu().numpy(), g["gate_r%02d" % int(ratio * 100)].astype(np.int64))


def test_route_select_dual_golden(dev):
    from dynamicvectorquantization_amd.router import route_select_dual
    g = C.load("route_dual_B2")
    gate, hc, hf = C.route_dual_inputs(g)
    t = lambda a: torch.from_numpy(a).to(dev)
    for gt, suffix in ((gate, ""), (g["logits"], "_logits")):     # int64 gate; f32 logits with ties / NaN
        out = route_select_dual(t(gt), t(hc), t(hf))
        assert out["indices"].dtype == torch.int64 and out["codebook_mask"].dtype == torch.float32
        assert tuple(out["gate"].shape) == (2, 2, 16, 16) and tuple(out["codebook_mask"].shape) == (2, 1, 32, 32)
        assert np.array_equal(out["indices"].cpu().numpy(), g["indices" + suffix].astype(np.int64))
        assert np.array_equal(out["codebook_mask"].cpu().numpy(), g["cmask" + suffix])
        assert np.array_equal(C.per_image_crc(out["h_dual"].cpu().numpy()), g["h_crc" + suffix])


def test_route_select_triple_golden(dev):
    from dynamicvectorquantization_amd.router import route_select_triple
    g = C.load("route_triple_B2")
    lg, hc, hm, hf = C.route_triple_inputs(g)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = route_select_triple(t(lg), t(hc), t(hm), t(hf))
    assert np.array_equal(out["indices"].cpu().numpy(), g["indices"].astype(np.int64))
    assert np.array_equal(out["codebook_mask"].cpu().numpy(), g["cmask"])
    assert np.array_equal(C.per_image_crc(out["h_triple"].cpu().numpy()), g["h_crc"])
    assert tuple(out["gate"].shape) == (2, 3, 8, 8)


@pytest.mark.parametrize("shape", [(1, 4, 1, 2), (3, 5, 3, 6), (2, 256, 16, 16), (2, 7, 1, 1), (3, 6, 5, 7), (2, 32, 15, 15)])
def test_route_select_shapes_vs_oracle(dev, oracle_mod, shape):
    """incl. odd coarse widths (rows of the dual select then move as 8-byte pieces: 240-px images, 15 x 15 cells)"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import route_select_dual, route_select_dual_entropy, route_select_triple
    B, Cc, hc_, wc_ = shape
    t = lambda a: torch.from_numpy(a).to(dev)
    hf, hc = synth.features(1, B, Cc, 2 * hc_, 2 * wc_), synth.features(2, B, Cc, hc_, wc_)
    gate = synth.grain_gate_dual(3, B, hc_, wc_)
    out = route_select_dual(t(gate), t(hc), t(hf))
    o = oracle_mod.route_select_dual(gate, hc, hf)
    for k in ("h_dual", "indices", "codebook_mask"):
        assert np.array_equal(out[k].cpu().numpy(), o[k]), k
    ent = synth.entropy_map(7, B, hc_, wc_)
    thr = 1.6777750253677368
    out = route_select_dual_entropy(t(ent), thr, t(hc), t(hf))
    og = oracle_mod.entropy_gate(ent, thr)
    o = oracle_mod.route_select_dual(og, hc, hf)
    for k in ("h_dual", "indices", "codebook_mask"):
        assert np.array_equal(out[k].cpu().numpy(), o[k]), k
    assert np.array_equal(out["gate"].cpu().numpy(), og.transpose(0, 3, 1, 2))      # the encoders hand the gate on as [B, 2, hc, wc]
    hf, hm = synth.features(4, B, Cc, 4 * hc_, 4 * wc_), synth.features(5, B, Cc, 2 * hc_, 2 * wc_)
    lg = synth.grain_logits_triple(6, B, hc_, wc_)
    out = route_select_triple(t(lg), t(hc), t(hm), t(hf))
    o = oracle_mod.route_select_triple(lg, hc, hm, hf)
    for k in ("h_triple", "indices", "codebook_mask"):
        assert np.array_equal(out[k].cpu().numpy(), o[k]), k


@pytest.mark.parametrize("which", ["dual", "triple"])
def test_feature_router_logits(dev, which):
    """feature routers (fused dvq_router_gate_f32 kernel under no_grad; GroupNorm, exp, k=512/768 GEMM are
    not bit-reproducible): logits within 1e-4 of the reference capture, grain indices equal wherever the
    reference margin exceeds 1e-3"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    g = C.load("feature_router_" + which)
    B, Cc = int(g["B"]), 256
    r = (DualGrainFeatureRouter if which == "dual" else TripleGrainFeatureRouter)(256, "group-32", "2layer-fc-SiLu")
    sd = {k: torch.from_numpy(synth.seeded_param(int(g["seed"]), i, k, tuple(v.shape)))
          for i, (k, v) in enumerate(r.state_dict().items())}
    r.load_state_dict(sd)
    r = r.to(dev).eval()
    t = lambda a: torch.from_numpy(a).to(dev)
    with torch.no_grad():
        if which == "dual":
            lg = r(h_fine=t(synth.features(3002, B, Cc, 32, 32)), h_coarse=t(synth.features(3012, B, Cc, 16, 16)))
        else:
            lg = r(h_fine=t(synth.features(3004, B, Cc, 32, 32)), h_median=t(synth.features(3014, B, Cc, 16, 16)),
                   h_coarse=t(synth.features(3024, B, Cc, 8, 8)))
    ref = g["logits"]
    got = lg.cpu().numpy()
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-4, "logit max-abs-diff %.3g" % np.max(np.abs(got - ref))
    srt = np.sort(ref, axis=-1)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert np.array_equal(got.argmax(-1)[clear], ref.argmax(-1)[clear])


@pytest.mark.parametrize("which", ["dual", "triple"])
def test_feature_router_logits_large_groupnorm_parameters(dev, which):
    """the fp16 range path of the gate's operand images (GroupNorm weights x 3000, biases x 200: |w| sqrt(n) + |b| far beyond
    65504; the pooling pass scales by a power of two derived from the parameters -- since round 5 from their per-branch maxima
    prepared by `dvq_router_gate_prepare_norm_f32`) against the REFERENCE's routers run on CPU with the same parameters
    (oracle/gen_golden_router_scaled.py; RouterDual.py:6-43, RouterTriple.py:6-56): logits within 1e-4, grain indices equal where the
    reference margin exceeds 1e-3; twice (the second call takes the prepared maxima from the cached weight prep)"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    g = C.load("feature_router_%s_scaled" % which)
    B, Cc = int(g["B"]), 256
    r = (DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu") if which == "dual"
         else TripleGrainFeatureRouter(256, "group-32", "2layer-fc-ReLu"))
    sd = {}
    for i, (k, v) in enumerate(r.state_dict().items()):
        p = synth.seeded_param(int(g["seed"]), i, k, tuple(v.shape))
        if k.startswith("feature_norm") and k.endswith("weight"):
            p = p * np.float32(3000.0)
        elif k.startswith("feature_norm") and k.endswith("bias"):
            p = p * np.float32(200.0)
        elif k == "gate.0.weight":
            p = p * np.float32(1.0 / 3000.0)
        sd[k] = torch.from_numpy(np.ascontiguousarray(p))
    r.load_state_dict(sd)
    r = r.to(dev).eval()
    t = lambda a: torch.from_numpy(a).to(dev)
    ref = g["logits"]
    if which == "dual":
        hs = [synth.features(3112, B, Cc, 16, 16), synth.features(3102, B, Cc, 32, 32)]                     # coarse -> fine
        names = ["coarse", "fine"]
    else:
        hs = [synth.features(3124, B, Cc, 8, 8), synth.features(3114, B, Cc, 16, 16), synth.features(3104, B, Cc, 32, 32)]
        names = ["coarse", "median", "fine"]
    # the routers' formula in float64 (GroupNorm(32, eps 1e-6) per branch, average pooling onto the coarse grid, concat coarse ->
    # fine, Linear -> SiLU | ReLU -> Linear): with parameters this large the reference's own fp32 result sits ~1e-3 off it (values of
    # a few thousand meet in sums of 512 / 768 products), so "as close to the exact value as the reference is" is the honest bar
    feats = []
    for i, (nm, h) in enumerate(zip(names, hs)):
        x = h.astype(np.float64)
        Bq, Cq, Hh, Ww = x.shape
        xg = x.reshape(Bq, 32, -1)
        xn = ((xg - xg.mean(-1, keepdims=True)) / np.sqrt(xg.var(-1, keepdims=True) + 1e-6)).reshape(x.shape)
        xn = xn * sd["feature_norm_%s.weight" % nm].numpy().astype(np.float64)[None, :, None, None] + \
            sd["feature_norm_%s.bias" % nm].numpy().astype(np.float64)[None, :, None, None]
        sc = 1 << i
        feats.append(xn.reshape(Bq, Cq, Hh // sc, sc, Ww // sc, sc).mean((3, 5)))
    X = np.concatenate(feats, 1).transpose(0, 2, 3, 1)
    hid = X @ sd["gate.0.weight"].numpy().astype(np.float64).T + sd["gate.0.bias"].numpy().astype(np.float64)
    hid = hid / (1.0 + np.exp(-hid)) if which == "dual" else np.maximum(hid, 0.0)
    truth = hid @ sd["gate.2.weight"].numpy().astype(np.float64).T + sd["gate.2.bias"].numpy().astype(np.float64)
    err_ref = float(np.max(np.abs(ref - truth)))
    assert err_ref < 2e-2                                           # the float64 restatement IS the reference's formula
    for _ in range(2):
        with torch.no_grad():
            if which == "dual":
                lg = r(h_fine=t(hs[1]), h_coarse=t(hs[0]))
            else:
                lg = r(h_fine=t(hs[2]), h_median=t(hs[1]), h_coarse=t(hs[0]))
        got = lg.cpu().numpy()
        assert got.shape == ref.shape
        err_got = float(np.max(np.abs(got - truth)))
        assert err_got <= max(3.0 * err_ref, 1e-4), (err_got, err_ref)
        assert float(np.max(np.abs(got - ref))) <= 4.0 * err_ref + 1e-4
        srt = np.sort(ref, axis=-1)
        clear = (srt[..., -1] - srt[..., -2]) > 50.0 * err_ref + 1e-3
        assert np.array_equal(got.argmax(-1)[clear], ref.argmax(-1)[clear])


@pytest.mark.parametrize("which,norm,gate_type,B,hc,wc", [
    ("dual", "group-32", "2layer-fc-SiLu", 3, 5, 6),       # ragged: 90 cells, last workgroup partly empty
    ("dual", "none", "1layer-fc", 2, 16, 16),
    ("dual", "group-8", "1layer-fc", 1, 4, 4),
    ("triple", "group-32", "2layer-fc-ReLu", 2, 8, 8),
    ("triple", "none", "2layer-fc-SiLu", 1, 3, 2),
    ("triple", "group-16", "1layer-fc", 5, 8, 8),
    ("dual", "none", "2layer-fc-SiLu", 2, 4, 4),
    # >= 64 cells per CU: the gate kernel takes two blocks of 32 cells per workgroup (ragged last workgroup, odd row
    # length -> scalar pooling path; triple with three logits; the fp16-range rescale across both blocks)
    ("dual", "group-32", "2layer-fc-SiLu", 70, 16, 15),
    ("triple", "group-16", "2layer-fc-ReLu", 260, 8, 8),
    ("dual", "none", "2layer-fc-SiLu", 64, 16, 16),
])
def test_fused_router_gate_matches_torch_ops(dev, which, norm, gate_type, B, hc, wc):
    """the fused kernel (no_grad) against the same module evaluated with differentiable torch ops
    (grad enabled), every gate / normalisation type the reference accepts"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    Cc = 64
    torch.manual_seed(11)
    r = (DualGrainFeatureRouter if which == "dual" else TripleGrainFeatureRouter)(Cc, norm, gate_type).to(dev)
    with torch.no_grad():
        for n_, p_ in r.named_parameters():                # non-trivial GroupNorm affine
            if "feature_norm" in n_:
                p_.copy_(torch.randn_like(p_) * 0.5 + (1.0 if n_.endswith("weight") else 0.0))
    t = lambda a: torch.from_numpy(a).to(dev)
    nbr = 2 if which == "dual" else 3
    # un-normalised 2-layer case: features far beyond the fp16 range exercise the tile rescale
    mag = np.float32(3.0e5) if (norm == "none" and gate_type != "1layer-fc" and which == "dual") else np.float32(1.5)
    feats = [t(synth.features(700 + i, B, Cc, hc << i, wc << i) * mag + np.float32(0.3)) for i in range(nbr)]
    kw = dict(h_coarse=feats[0], h_fine=feats[-1])
    if nbr == 3:
        kw["h_median"] = feats[1]
    with torch.no_grad():
        fused = r(**kw)
    ref = r(**kw)                                          # parameters require grad -> torch-op path
    assert ref.requires_grad and not fused.requires_grad
    assert fused.shape == ref.shape == (B, hc, wc, nbr)
    err = float((fused - ref.detach()).abs().max())
    assert err < 1e-4 * max(1.0, float(ref.detach().abs().max())), err


def test_encode_dual_end_to_end(dev, oracle_mod, golden_dir):
    """BASELINE configs[2] shape at B=4: entropy router -> select -> VectorQuantize2 (encode glue)"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    B, K, D = 4, 1024, 256
    E = synth.codebook_trained(K, D)
    hf, hc = synth.z_tokens(E, B, 32, 32, 2103), synth.z_tokens(E, B, 16, 16, 2113)
    ent = synth.entropy_map(5103, B, 16, 16)
    r = DualGrainFixedEntropyRouter(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    vq = _vq2(dev, E, 1)
    t = lambda a: torch.from_numpy(a).to(dev)
    with torch.no_grad():
        quant, emb_loss, info, grain, gate = encode_dual(r, vq, t(hf), t(hc), entropy=t(ent))
    og = oracle_mod.entropy_gate(ent, r.fine_grain_threshold)
    osel = oracle_mod.route_select_dual(og, hc, hf)
    o = oracle_mod.vq_assign_nchw(osel["h_dual"], E, osel["codebook_mask"])
    assert np.array_equal(grain.cpu().numpy(), osel["indices"]) and tuple(gate.shape) == (B, 2, 16, 16)
    assert np.array_equal(info[2].cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(quant.cpu().numpy(), o["zq"])
    assert C.loss_close(float(emb_loss), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    c = info[2].cpu().numpy()
    coarse = osel["indices"] == 0                               # all 2x2 codes of a coarse cell are equal
    for dy in (0, 1):
        for dx in (0, 1):
            assert np.array_equal(c[:, dy::2, dx::2][coarse], c[:, 0::2, 0::2][coarse])


def test_encode_triple_end_to_end(dev, oracle_mod):
    """BASELINE configs[3] shape at B=2 (per-rank slice of the 8-GPU job)"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_triple
    B, K, D = 2, 1024, 256
    E = synth.codebook_trained(K, D)
    hf, hm, hc = (synth.z_tokens(E, B, 32, 32, 2104), synth.z_tokens(E, B, 16, 16, 2114),
                  synth.z_tokens(E, B, 8, 8, 2124))
    lg = synth.grain_logits_triple(4104, B, 8, 8)
    t = lambda a: torch.from_numpy(a).to(dev)
    router = lambda h_fine, h_median, h_coarse, entropy=None: t(lg)
    vq = _vq2(dev, E, 1)
    with torch.no_grad():
        quant, emb_loss, info, grain, gate = encode_triple(router, vq, t(hf), t(hm), t(hc))
    osel = oracle_mod.route_select_triple(lg, hc, hm, hf)
    o = oracle_mod.vq_assign_nchw(osel["h_triple"], E, osel["codebook_mask"])
    assert np.array_equal(grain.cpu().numpy(), osel["indices"])
    assert np.array_equal(info[2].cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(quant.cpu().numpy(), o["zq"])
    assert C.loss_close(float(emb_loss), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))


@pytest.mark.parametrize("mode", MODES)
def test_full_size_properties_cfg3(dev, oracle_mod, mode):
    """BASELINE configs[2] at full size (B=256, 32x32x256, K=1024): properties that need no dense
    CPU run -- (1) encode -> gather -> re-encode is idempotent with z_q == e, (2) exact and filter
    modes agree, (3) a random sample of tokens equals the oracle, (4) loss equals the mean of
    (z_q - z)^2 recomputed by torch."""
    from dynamicvectorquantization_amd import synth
    B, K, D = 256, 1024, 256
    E = synth.codebook_trained(K, D)
    vq = _vq2(dev, E, mode)
    z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 2003)).to(dev)
    with torch.no_grad():
        xq, loss, (_, _, codes) = vq(z)
        e = vq.get_codebook_entry(codes).permute(0, 3, 1, 2).contiguous()
        xq2, loss2, (_, _, codes2) = vq(e)
    assert torch.equal(codes2, codes) and torch.equal(xq2, e) and float(loss2) == 0.0
    assert float(torch.max(torch.abs(xq - e))) <= 1e-5            # z + (e - z) vs e
    ref_loss = (1.25 * torch.mean((e.double() - z.double()) ** 2)).item()
    assert abs(float(loss) - ref_loss) <= 1e-5 * ref_loss
    g = C.load("vq2_cfg3_B256_crc")
    assert np.array_equal(C.per_image_crc(codes.cpu().numpy()), g["codes_crc"])
    sample = np.arange(0, B, 37)
    zs = z[sample].cpu().numpy()
    o = oracle_mod.vq_assign_nchw(zs, E, None)
    assert np.array_equal(codes[sample].cpu().numpy().reshape(len(sample), -1), o["codes"])
    assert np.array_equal(xq[sample].cpu().numpy(), o["zq"])


def test_autograd_straight_through(dev):
    """gradient = identity through z_q plus the commitment term 2*beta*(z - e)*m/numel"""
    from dynamicvectorquantization_amd import synth
    K, D = 64, 256
    E = synth.codebook_trained(K, D, seed=3)
    vq = _vq2(dev, E, 1)
    z = torch.from_numpy(synth.z_tokens(E, 1, 4, 4, 4)).to(dev).requires_grad_(True)
    mask = torch.full((1, 1, 4, 4), 0.25, device=dev)
    xq, loss, (_, _, codes) = vq(z, codebook_mask=mask)
    (xq.sum() + 3.0 * loss).backward()
    e = vq.get_codebook_entry(codes).permute(0, 3, 1, 2)
    expect = 1.0 + 3.0 * 0.25 * 2.0 * (z.detach() - e) * mask / z.numel()
    assert torch.allclose(z.grad, expect, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_filter_mode_matches_exact_mode_on_random_sweep():
    """The exact mode is pinned to the oracle above; the filter mode must agree with it bit for bit on
    random shapes, scales (fp16-range overflow / underflow), duplicated codes, NaN / Inf latents."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_modes", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_modes.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, n = fuzz.run(300, seed=777, verbose=False)           # K up to 16384, every pass-1 form
    assert n == 300 and bad == 0


def test_hot_path_is_graph_capturable(dev):
    """gate + select + VQ assign captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed on new
    inputs: same bits as the eager launches (the ABI never allocates or synchronises)"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    from dynamicvectorquantization_amd.router import entropy_gate, route_select_dual
    B, K, D = 4, 1024, 256
    E = torch.from_numpy(synth.codebook_trained(K, D)).to(dev)
    En = E.cpu().numpy()
    mk = lambda seed: (torch.from_numpy(synth.z_tokens(En, B, 32, 32, seed)).to(dev),
                       torch.from_numpy(synth.z_tokens(En, B, 16, 16, seed + 1)).to(dev),
                       torch.from_numpy(synth.entropy_map(seed + 2, B, 16, 16)).to(dev))
    hf, hc, ent = mk(9100)
    prep = _CodebookPrep()
    h_dual = torch.empty_like(hf); grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev)
    cmask = torch.empty((B, 1, 32, 32), device=dev); zq = torch.empty_like(hf)
    codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
    gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)

    def step():
        gate.copy_(entropy_gate(ent, 1.6777750253677368))
        route_select_dual(gate, hc, hf, out=(h_dual, grain, cmask))
        vq_assign(h_dual, E, prep, cmask, beta=0.25, out=(zq, codes, loss))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()                                   # warm-up: prep, workspace, kernel attributes
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    hf2, hc2, ent2 = mk(9200)
    hf.copy_(hf2); hc.copy_(hc2); ent.copy_(ent2)
    g.replay()
    torch.cuda.synchronize()
    got = (zq.clone(), codes.clone(), loss.clone(), grain.clone())
    step()
    torch.cuda.synchronize()
    assert torch.equal(got[1], codes) and torch.equal(got[0], zq) and torch.equal(got[3], grain)
    assert torch.equal(got[2], loss)


@pytest.mark.parametrize("B", [16, 8, 4])
def test_filter_queue_overflow_falls_back_to_exact(dev, B):
    """a codebook with widely mixed norms leaves most tokens undecided: every shard of the resolver queue
    fills up and the remainder goes through the exact list; output still equals the exact mode bit for bit
    (B = 8, 4: the split form of pass 1 -- several workgroups per token block -- with full shards)"""
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    rng = np.random.default_rng(3)
    K, D = 1024, 256
    E = synth.codebook_trained(K, D, seed=77)
    E = np.ascontiguousarray(E * np.exp2(rng.integers(-5, 5, size=(K, 1))).astype(np.float32))
    zt = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 4242)).to(dev)
    Et = torch.from_numpy(E).to(dev)
    mask = torch.from_numpy(np.where(rng.random((B, 1, 32, 32)) < 0.5, 1.0, 0.25).astype(np.float32)).to(dev)
    pe, pf = _CodebookPrep(), _CodebookPrep()
    zq0, c0, l0 = vq_assign(zt, Et, pe, mask, mode=_lib.MODE_EXACT)
    zq1, c1, l1 = vq_assign(zt, Et, pf, mask, mode=_lib.MODE_FILTER)
    queued, listed = pf.fallback_count()
    if B == 16:
        assert queued == 4096 and listed > 1000, (queued, listed)    # 64 shards x 64 slots full, the rest listed
    else:
        assert queued >= 1024 and listed > 1000, (queued, listed)    # full shards
    assert torch.equal(c0, c1) and torch.equal(zq0, zq1)
    assert abs(float(l0[1]) - float(l1[1])) <= 1e-6 * abs(float(l0[1]))


def test_empty_batch(dev):
    """B = 0: empty codes / z_q / routed features and a NaN loss, as the reference's torch ops return"""
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import entropy_gate, route_select_dual
    vq = VectorQuantize2(64, 64).to(dev).eval()
    z = torch.empty((0, 64, 8, 8), device=dev)
    with torch.no_grad():
        zq, loss, (_, _, codes) = vq(z, codebook_mask=torch.empty((0, 1, 8, 8), device=dev))
    assert zq.shape == (0, 64, 8, 8) and codes.shape == (0, 8, 8) and codes.dtype == torch.int64
    assert bool(torch.isnan(loss))
    gate = entropy_gate(torch.empty((0, 4, 4), device=dev), 1.0)
    assert gate.shape == (0, 4, 4, 2) and gate.dtype == torch.int64
    out = route_select_dual(gate, torch.empty((0, 64, 4, 4), device=dev), z)
    assert out["h_dual"].shape == (0, 64, 8, 8) and out["indices"].shape == (0, 4, 4)


def test_route_select_dual_entropy_fused(dev, oracle_mod):
    """entropy router fused into the select kernel == entropy_gate + route_select_dual == oracle, incl. NaN / threshold ties"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import entropy_gate, route_select_dual, route_select_dual_entropy
    B, Cc, hc_, wc_ = 3, 40, 6, 8
    thr = 1.6777750253677368
    ent = synth.entropy_map(881, B, hc_, wc_)
    ent[0, 0, 0] = np.float32(thr); ent[0, 0, 1] = np.nan; ent[1, 2, 3] = np.inf; ent[2, 5, 7] = -np.inf
    hc, hf = synth.features(882, B, Cc, hc_, wc_), synth.features(883, B, Cc, 2 * hc_, 2 * wc_)
    t = lambda a: torch.from_numpy(a).to(dev)
    fused = route_select_dual_entropy(t(ent), thr, t(hc), t(hf))
    gate = entropy_gate(t(ent), thr)
    ref = route_select_dual(gate, t(hc), t(hf))
    for k in ("h_dual", "indices", "codebook_mask"):
        assert torch.equal(fused[k], ref[k]), k
    assert torch.equal(fused["gate"], gate.permute(0, 3, 1, 2)) and fused["gate"].dtype == torch.int64
    og = oracle_mod.entropy_gate(ent, thr)
    o = oracle_mod.route_select_dual(og, hc, hf)
    for k in ("h_dual", "indices", "codebook_mask"):
        assert np.array_equal(fused[k].cpu().numpy(), o[k]), k


def test_wide_kernel_codes_only_and_ragged(dev):
    """K >= 2048 with >= 131072 tokens takes the two-blocks-per-wave pass-1 kernel (here K = 8192): codes-only call
    (no z_q, no loss) and a token count that is not a multiple of 64 agree with the exact mode; the oracle / golden
    check of that kernel is tests/test_routed.py::test_k16384_dispatch_size_vs_oracle_and_golden"""
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    K, D = 8192, 256
    E = synth.codebook_trained(K, D, seed=31)
    Et = torch.from_numpy(E).to(dev)
    z = torch.from_numpy(synth.z_tokens(E, 131, 31, 33, 4321)).to(dev)         # N = 134013 tokens (odd)
    pe, pf = _CodebookPrep(), _CodebookPrep()
    _, c0, _ = vq_assign(z, Et, pe, None, want_zq=False, want_loss=False, mode=_lib.MODE_EXACT)
    _, c1, _ = vq_assign(z, Et, pf, None, want_zq=False, want_loss=False, mode=_lib.MODE_FILTER)
    assert torch.equal(c0, c1)
    zq0, c2, l0 = vq_assign(z, Et, pe, None, mode=_lib.MODE_EXACT)
    zq1, c3, l1 = vq_assign(z, Et, pf, None, mode=_lib.MODE_FILTER)
    assert torch.equal(c2, c3) and torch.equal(zq0, zq1) and torch.equal(c0, c2)
    assert abs(float(l0[1]) - float(l1[1])) <= 1e-6 * abs(float(l0[1]))


@pytest.mark.gpu
def test_stream_slots_pipelined_encode_matches_serial(dev, oracle_mod):
    """encode.StreamSlots: independent batches round-robin on 3 streams through ONE quantizer / router (per-stream
    workspaces) give the bits of the serial calls, inputs produced on the caller's stream are ordered before the slot's
    work, join() orders the caller after it"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import StreamSlots, encode_dual
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    B, K, D = 8, 512, 256
    E = synth.codebook_trained(K, D, seed=31)
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E).to(dev))
    router = DualGrainFixedEntropyRouter(os.path.join(os.path.dirname(__file__), "golden",
                                                      "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    t = lambda a: torch.from_numpy(a).to(dev)
    batches = [(synth.z_tokens(E, B, 32, 32, 7000 + i), synth.z_tokens(E, B, 16, 16, 7100 + i), synth.entropy_map(7200 + i, B, 16, 16))
               for i in range(7)]
    with torch.no_grad():
        serial = []
        for hf, hc, ent in batches:
            q, loss, info, grain, gate = encode_dual(router, vq, t(hf), t(hc), t(ent))
            serial.append((q.clone(), float(loss), info[2].clone(), grain.clone()))
        torch.cuda.synchronize()
        slots = StreamSlots(3)
        outs = [None] * len(batches)
        for i, (hf, hc, ent) in enumerate(batches):
            a, b, c = t(hf) * 1.0, t(hc) * 1.0, t(ent) * 1.0          # produced on the caller's stream just before
            with slots.next() as slot:
                outs[i] = encode_dual(router, vq, a, b, c)
                for x in (a, b, c):
                    x.record_stream(slot.stream)
        slots.join()
        for i, (q, loss, info, grain, gate) in enumerate(outs):
            assert torch.equal(q, serial[i][0]) and torch.equal(info[2], serial[i][2]) and torch.equal(grain, serial[i][3])
            assert abs(float(loss) - serial[i][1]) <= 1e-6 * abs(serial[i][1])


@pytest.mark.parametrize("which,Cc,norm,gate_type,B,hc,wc", [
    ("dual", 256, "group-32", "2layer-fc-SiLu", 3, 5, 6),      # 90 cells: the last 32-cell block is partly empty (zero-filled image)
    ("triple", 256, "group-32", "2layer-fc-ReLu", 2, 8, 8),    # 48 k-steps: 384 resident fragment registers
    ("dual", 128, "group-16", "2layer-fc-SiLu", 5, 4, 4),      # 16 k-steps, hidden 256 = two hidden groups
    ("triple", 64, "group-8", "2layer-fc-SiLu", 4, 3, 5),      # 12 k-steps (ring chunks of 4), hidden 192: the last group has empty row tiles
    ("dual", 256, "group-8", "2layer-fc-SiLu", 40, 16, 16),    # 320 cell blocks: several per workgroup, 32 channels per group
    ("triple", 256, "group-32", "2layer-fc-SiLu", 20, 8, 8),
])
def test_router_gate_gemm_form_matches_torch_ops(dev, which, Cc, norm, gate_type, B, hc, wc):
    """the GEMM form of the gate (gate_pool_kernel<true> -> gate_gemm_kernel -> gate_finalize_kernel: hidden-layer weights resident
    in registers, feature images streamed through an LDS ring) on shapes that take it -- GroupNorm with whole octets of channels
    per group, a hidden layer -- against the module's own differentiable torch ops; deterministic across calls"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    torch.manual_seed(23)
    nbr = 2 if which == "dual" else 3
    r = (DualGrainFeatureRouter if nbr == 2 else TripleGrainFeatureRouter)(Cc, norm, gate_type).to(dev)
    with torch.no_grad():
        for n_, p_ in r.named_parameters():
            if "feature_norm" in n_:
                p_.copy_(torch.randn_like(p_) * 0.5 + (1.0 if n_.endswith("weight") else 0.0))
    t = lambda a: torch.from_numpy(a).to(dev)
    feats = [t(synth.features(7700 + i, B, Cc, hc << i, wc << i) * np.float32(1.5) + np.float32(0.3)) for i in range(nbr)]
    kw = dict(h_coarse=feats[0], h_fine=feats[-1])
    if nbr == 3:
        kw["h_median"] = feats[1]
    with torch.no_grad():
        fused = r(**kw)
        again = r(**kw)
    ref = r(**kw).detach()
    assert torch.equal(fused, again)
    err = float((fused - ref).abs().max())
    assert err < 1e-4 * max(1.0, float(ref.abs().max())), err
    # GroupNorm parameters large enough that the normalised features leave the fp16 range: the a-priori power-of-two scale
    with torch.no_grad():
        for n_, p_ in r.named_parameters():
            if "feature_norm" in n_ and n_.endswith("weight"):
                p_.mul_(3000.0)
        big = r(**kw)
    ref_big = r(**kw).detach()
    assert torch.isfinite(big).all()
    assert float((big - ref_big).abs().max()) < 1e-4 * max(1.0, float(ref_big.abs().max()))


@pytest.mark.gpu
def test_fused_router_gate_fuzz(dev):
    """40 random router configurations (branches, channels, grid, batch, normalisation, gate type): the fused gate
    (one pooling pass + MLP kernel, either cell-block form, vector / scalar pooling) against the module's own torch ops"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    rng = np.random.default_rng(20261003)
    t = lambda a: torch.from_numpy(a).to(dev)
    for case in range(40):
        nbr = int(rng.integers(2, 4))
        Cc = int(rng.choice([32, 64, 128, 256]))
        hc, wc, B = int(rng.integers(1, 13)), int(rng.integers(1, 13)), int(rng.integers(1, 10))
        if case % 10 == 9:                                   # now and then enough cells for two blocks per workgroup
            hc, wc, B = 16, 16, 64 + int(rng.integers(0, 3))
            Cc = int(rng.choice([32, 64]))
        norm = str(rng.choice(["none", "group-8", "group-32"]))
        gate_type = str(rng.choice(["1layer-fc", "2layer-fc-SiLu"] + (["2layer-fc-ReLu"] if nbr == 3 else [])))
        torch.manual_seed(100 + case)
        r = (DualGrainFeatureRouter if nbr == 2 else TripleGrainFeatureRouter)(Cc, norm, gate_type).to(dev)
        with torch.no_grad():
            for n_, p_ in r.named_parameters():
                if "feature_norm" in n_:
                    p_.copy_(torch.randn_like(p_) * 0.5 + (1.0 if n_.endswith("weight") else 0.0))
        feats = [t(synth.features(900 + 3 * case + i, B, Cc, hc << i, wc << i) * np.float32(1.5) + np.float32(0.3)) for i in range(nbr)]
        kw = dict(h_coarse=feats[0], h_fine=feats[-1])
        if nbr == 3:
            kw["h_median"] = feats[1]
        with torch.no_grad():
            fused = r(**kw)
            again = r(**kw)                                  # second call: cached weight images
        ref = r(**kw).detach()
        assert torch.equal(fused, again), case
        err = float((fused - ref).abs().max())
        assert err < 1e-4 * max(1.0, float(ref.abs().max())), (case, nbr, Cc, hc, wc, B, norm, gate_type, err)



@pytest.mark.parametrize("D", [32, 96, 160, 192, 224])
def test_widths_served_by_zero_padding(dev, oracle_mod, D):
    """the reference classes take any codebook_dim (quantize2_mask.py:136-155); the kernels exist for 64 / 128 /
    256 channels and the drop-in serves every other multiple of 32 below 256 by appending ZERO channels (exact: fma(0, 0, acc) and
    + 0 in the norm's partial sums).  Codes and z_q bit-exact vs the oracle (itself pinned against the imported reference at these
    widths: oracle/validate_against_reference.py), loss 1e-5: VectorQuantize2 with a mask, tie-stress codebook, row-major input,
    VectorQuantizer2; a width that is NOT a multiple of 32 raises."""
    from dynamicvectorquantization_amd import _lib, synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, VectorQuantizer2
    K = 512
    for kind, seed in (("trained", 40 + D), ("default", 50 + D)):
        E = synth.codebook_trained(K, D) if kind == "trained" else synth.codebook_default_init(K, D)
        z = synth.z_tokens(E, 3, 16, 16, seed) * (np.float32(1.0) if kind == "trained" else np.float32(0.002))
        mask = np.where(synth.bernoulli(seed + 1, (3, 1, 16, 16), 0.5), 1.0, 0.25).astype(np.float32)
        o = oracle_mod.vq_assign_nchw(z, E, mask)
        for mode in (_lib.MODE_FILTER, _lib.MODE_EXACT):
            xq, loss, codes = _run_vq2(dev, z, E, mask, mode)
            assert np.array_equal(codes.reshape(3, -1), o["codes"]), (D, kind, mode)
            assert np.array_equal(xq, o["zq"]), (D, kind, mode)
            assert C.loss_close(loss, oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25)), (D, kind, mode)
    # row-major tokens (channel_last) and the taming class
    E = synth.codebook_trained(K, D)
    z = synth.z_tokens(E, 2, 8, 8, 70 + D)
    o = oracle_mod.vq_assign_nchw(z, E, None)
    vq = VectorQuantize2(K, D, accept_image_fmap=False, channel_last=True).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    with torch.no_grad():
        xq, loss, (_, _, codes) = vq(torch.from_numpy(z.reshape(2, D, 64)).to(dev).permute(0, 2, 1).contiguous())
    assert np.array_equal(codes.cpu().numpy(), o["codes"])
    assert np.array_equal(xq.permute(0, 2, 1).cpu().numpy().reshape(z.shape), o["zq"])
    g = VectorQuantizer2(K, D, beta=0.25, legacy=False).to(dev).eval()
    g.embedding.weight.data.copy_(torch.from_numpy(E))
    with torch.no_grad():
        zq, gl, (_, _, idx) = g(torch.from_numpy(z).to(dev))
    assert np.array_equal(idx.cpu().numpy().reshape(2, -1), o["codes"]) and np.array_equal(zq.cpu().numpy(), o["zq"])
    assert C.loss_close(float(gl), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25, legacy=False))
    # training-mode forward + backward runs at this width (EMA statistics, straight-through gradient)
    vt = VectorQuantize2(K, D).to(dev).train()
    vt.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    x = torch.from_numpy(z).to(dev).requires_grad_(True)
    xq, loss, _ = vt(x)
    (xq.sum() + loss).backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad).all())
    with pytest.raises(_lib.DvqError):
        VectorQuantize2(64, 48).to(dev).eval()(torch.zeros(1, 48, 4, 4, device=dev))
