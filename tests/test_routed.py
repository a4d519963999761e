"""GPU (-m gpu): the routed assign (dvq_vq_assign_routed_{dual,triple}_f32 -- routing tail + VectorQuantize2
as one op, the select fused into pass 1) against the CPU oracle's select + assign on the same seeded inputs, at
every form pass 1 takes (32-wide output grids stage the coarser branches through LDS, other grids address them per
lane), and at DISPATCH size: the exact bench step at
B = 256 on all images, configs[1] end to end at B = 64, the triple encode at the per-rank size B = 128, K = 16384
through the wide pass-1 kernel + sliced resolver at B = 128 and at configs[4]'s B = 512.
Bar: codes, grain indices, codebook_mask, gate and z_q bit-exact; loss 1e-5."""
import numpy as np
import pytest
import torch

from tests import _cases as C

pytestmark = pytest.mark.gpu

THR = 1.6777750253677368


def _t(dev):
    return lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _check_dual(r, o_sel, o, B, beta=0.25, oracle_mod=None):
    assert np.array_equal(r["indices"].cpu().numpy(), o_sel["indices"]), "grain indices"
    assert np.array_equal(r["codebook_mask"].cpu().numpy(), o_sel["codebook_mask"]), "codebook_mask"
    assert np.array_equal(r["codes"].cpu().numpy().reshape(B, -1), o["codes"]), "codes"
    if r["zq"] is not None:
        assert np.array_equal(r["zq"].cpu().numpy(), o["zq"], equal_nan=True), "z_q"
    if r["loss"] is not None:
        assert C.loss_close(float(r["loss"][1]), oracle_mod.vq_loss(o["sqerr"], o["numel"], beta)), "loss"


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(1, 1, 2), (3, 4, 6), (9, 16, 16), (17, 5, 8), (2, 32, 32), (3, 5, 7), (4, 2, 16)])
def test_routed_dual_vs_oracle(dev, oracle_mod, shape, mode):
    """ragged batches, grids from 1x2 to 32x32 cells incl. odd widths (15-wide grids: 240-px images); the 32-wide
    output grids (9, 16, 16) / (4, 2, 16) take the LDS-staged form, the others the per-lane form;
    int64 gate, f32 logits (ties / NaN) and the fused entropy router"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
    B, hc, wc = shape
    K, D = 333, 256
    E = synth.codebook_trained(K, D, seed=500 + hc)
    hf, hco = synth.z_tokens(E, B, 2 * hc, 2 * wc, 510 + hc), synth.z_tokens(E, B, hc, wc, 520 + hc)
    t = _t(dev)
    prep = _CodebookPrep()
    gate = synth.grain_gate_dual(530 + hc, B, hc, wc)
    gate[0] = np.array([1, 0])                                    # image 0 all coarse
    if B > 1:
        gate[1] = np.array([0, 1])                                # image 1 all fine
    logits = synth.normal(540 + hc, (B, hc, wc, 2))
    logits[0, 0, 0] = 0.5                                         # tie -> first index (coarse)
    logits[-1, -1, -1, 1] = np.nan                                # NaN counts as the maximum
    ent = synth.entropy_map(550 + hc, B, hc, wc)
    ent[0, 0, 0] = np.float32(THR)
    ent[-1, -1, -1] = np.nan
    for kind, g in (("gate", gate), ("gate", logits), ("entropy", ent)):
        if kind == "entropy":
            r = vq_assign_routed_dual(t(hco), t(hf), t(E), prep, entropy=t(g), threshold=THR, mode=mode)
            og = oracle_mod.entropy_gate(g, THR)
            assert np.array_equal(r["gate"].cpu().numpy(), og)
        else:
            r = vq_assign_routed_dual(t(hco), t(hf), t(E), prep, gate=t(g), mode=mode)
            og = g
        o_sel = oracle_mod.route_select_dual(og, hco, hf)
        o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
        _check_dual(r, o_sel, o, B, oracle_mod=oracle_mod)
    # codes-only call
    r = vq_assign_routed_dual(t(hco), t(hf), t(E), prep, gate=t(gate), mode=mode, want_zq=False, want_loss=False)
    o_sel = oracle_mod.route_select_dual(gate, hco, hf)
    o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
    _check_dual(r, o_sel, o, B, oracle_mod=oracle_mod)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 2, 3), (10, 8, 8), (2, 16, 16), (5, 3, 8)])
def test_routed_triple_vs_oracle(dev, oracle_mod, shape, mode):
    """(10, 8, 8) and (5, 3, 8) are 32 positions wide: median and coarse branches staged through LDS"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_triple
    B, hc, wc = shape
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    hf, hm, hco = (synth.z_tokens(E, B, 4 * hc, 4 * wc, 610 + hc), synth.z_tokens(E, B, 2 * hc, 2 * wc, 620 + hc),
                   synth.z_tokens(E, B, hc, wc, 630 + hc))
    lg = synth.grain_logits_triple(640 + hc, B, hc, wc)
    lg[0, 0, 0] = 0.25                                            # three-way tie -> coarse
    if B > 1:
        lg[1] = np.array([0.0, 0.0, 1.0])                         # image 1 all fine
    gi = np.stack([(lg.argmax(-1) == k) for k in range(3)], -1).astype(np.int64)
    t = _t(dev)
    prep = _CodebookPrep()
    for g in (lg, gi):
        r = vq_assign_routed_triple(t(hco), t(hm), t(hf), t(E), prep, t(g), mode=mode)
        o_sel = oracle_mod.route_select_triple(g, hco, hm, hf)
        o = oracle_mod.vq_assign_nchw(o_sel["h_triple"], E, o_sel["codebook_mask"])
        _check_dual(r, o_sel, o, B, oracle_mod=oracle_mod)


def test_staged_form_special_tokens(dev, oracle_mod):
    """the LDS-staged select (32-wide grid), dual and triple: NaN / Inf / huge tokens in every branch (exact list),
    zero tokens, all-coarse and all-fine images, codes-only and loss-only calls"""
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual, vq_assign_routed_triple
    t = _t(dev)
    B, K, D = 6, 300, 256
    E = synth.codebook_trained(K, D, seed=91)
    E[7] = E[3]
    hf, hm, hco = (synth.z_tokens(E, B, 32, 32, 92), synth.z_tokens(E, B, 16, 16, 93), synth.z_tokens(E, B, 8, 8, 94))
    for a in (hf, hm, hco):
        a[0, 5, 0, 0] = np.nan
        a[1, :, 1, 1] = np.inf
        a[2, 9, 2, 2] = -np.inf
        a[3, :, 3, 3] *= np.float32(1e6)
        a[4, :, 0, 1] = 0.0
        a[5, :, 7, 7] = np.nan
    lg = synth.grain_logits_triple(95, B, 8, 8)
    lg[4] = np.array([1.0, 0.0, -1.0], dtype=np.float32)          # all coarse
    lg[5] = np.array([-1.0, 0.0, 1.0], dtype=np.float32)          # all fine
    prep = _CodebookPrep()
    r = vq_assign_routed_triple(t(hco), t(hm), t(hf), t(E), prep, t(lg))
    o_sel = oracle_mod.route_select_triple(lg, hco, hm, hf)
    o = oracle_mod.vq_assign_nchw(o_sel["h_triple"], E, o_sel["codebook_mask"])
    assert np.array_equal(r["indices"].cpu().numpy(), o_sel["indices"])
    assert np.array_equal(r["codebook_mask"].cpu().numpy(), o_sel["codebook_mask"])
    assert np.array_equal(r["codes"].cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(r["zq"].cpu().numpy(), o["zq"], equal_nan=True)
    assert prep.fallback_count()[1] > 0
    # dual, clean data: all outputs, then codes-only and the op's loss against the oracle's
    hf2, hc2 = synth.z_tokens(E, B, 32, 32, 96), synth.z_tokens(E, B, 16, 16, 97)
    g2 = synth.grain_gate_dual(98, B, 16, 16)
    g2[0] = np.array([1, 0]); g2[1] = np.array([0, 1])            # all coarse / all fine images
    r2 = vq_assign_routed_dual(t(hc2), t(hf2), t(E), _CodebookPrep(), gate=t(g2))
    o_sel = oracle_mod.route_select_dual(g2, hc2, hf2)
    o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
    _check_dual(r2, o_sel, o, B, oracle_mod=oracle_mod)
    r3 = vq_assign_routed_dual(t(hc2), t(hf2), t(E), _CodebookPrep(), gate=t(g2), want_zq=False)
    assert np.array_equal(r3["codes"].cpu().numpy().reshape(B, -1), o["codes"])
    ol = float(oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    assert abs(float(r3["loss"][1]) - ol) <= 1e-5 * abs(ol)


def test_routed_special_tokens_and_queue_overflow(dev, oracle_mod):
    """NaN / Inf / huge tokens in every branch go through the exact list (routed ids); a codebook with widely
    mixed norms overflows the resolver queue; D = 64 / 128"""
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual, vq_assign_routed_triple
    t = _t(dev)
    B, hc, wc, K, D = 5, 4, 4, 200, 256
    E = synth.codebook_trained(K, D, seed=71)
    E[7] = E[3]                                                    # duplicate code: first index wins
    hf, hm, hco = (synth.z_tokens(E, B, 16, 16, 72), synth.z_tokens(E, B, 8, 8, 73), synth.z_tokens(E, B, 4, 4, 74))
    for a in (hf, hm, hco):
        a[0, 5, 0, 0] = np.nan
        a[1, :, 1, 1] = np.inf
        a[2, 9, 2, 2] = -np.inf
        a[3, :, 3, 3] *= np.float32(1e6)
        a[4, :, 0, 1] = 0.0
    lg = synth.grain_logits_triple(75, B, hc, wc)
    prep = _CodebookPrep()
    r = vq_assign_routed_triple(t(hco), t(hm), t(hf), t(E), prep, t(lg))
    o_sel = oracle_mod.route_select_triple(lg, hco, hm, hf)
    o = oracle_mod.vq_assign_nchw(o_sel["h_triple"], E, o_sel["codebook_mask"])
    assert np.array_equal(r["codes"].cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(r["zq"].cpu().numpy(), o["zq"], equal_nan=True)
    assert prep.fallback_count()[1] > 0                           # the exact list was used
    # queue overflow (dual), the routed analogue of test_filter_queue_overflow_falls_back_to_exact
    rng = np.random.default_rng(3)
    K2, B2 = 1024, 24
    E2 = synth.codebook_trained(K2, D, seed=77)
    E2 = np.ascontiguousarray(E2 * np.exp2(rng.integers(-5, 5, size=(K2, 1))).astype(np.float32))
    hf2, hc2 = synth.z_tokens(E2, B2, 32, 32, 4243), synth.z_tokens(E2, B2, 16, 16, 4244)
    g2 = synth.grain_gate_dual(4245, B2, 16, 16)
    p1, p0 = _CodebookPrep(), _CodebookPrep()
    r1 = vq_assign_routed_dual(t(hc2), t(hf2), t(E2), p1, gate=t(g2), mode=_lib.MODE_FILTER)
    r0 = vq_assign_routed_dual(t(hc2), t(hf2), t(E2), p0, gate=t(g2), mode=_lib.MODE_EXACT)
    queued, listed = p1.fallback_count()
    # 64 shards full, the rest through the exact list
    assert queued >= 4096 and listed > 500, (queued, listed)
    assert torch.equal(r0["codes"], r1["codes"]) and torch.equal(r0["zq"], r1["zq"])
    assert abs(float(r0["loss"][1]) - float(r1["loss"][1])) <= 1e-6 * abs(float(r0["loss"][1]))
    o_sel = oracle_mod.route_select_dual(g2, hc2, hf2)
    o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E2, o_sel["codebook_mask"])
    assert np.array_equal(r1["codes"].cpu().numpy().reshape(B2, -1), o["codes"])
    for Dd in (64, 128):
        E3 = synth.codebook_trained(300, Dd, seed=80 + Dd)
        hf3, hc3 = synth.z_tokens(E3, 3, 8, 8, 81 + Dd), synth.z_tokens(E3, 3, 4, 4, 82 + Dd)
        g3 = synth.grain_gate_dual(83 + Dd, 3, 4, 4)
        r3 = vq_assign_routed_dual(t(hc3), t(hf3), t(E3), _CodebookPrep(), gate=t(g3))
        o_sel = oracle_mod.route_select_dual(g3, hc3, hf3)
        o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E3, o_sel["codebook_mask"])
        _check_dual(r3, o_sel, o, 3, oracle_mod=oracle_mod)


def test_bench_step_full_size_all_images(dev, oracle_mod):
    """BASELINE configs[2] exactly as bench.py runs it (entropy gate + routing + masked assign, B = 256, K = 1024:
    2048 workgroups, LDS-staged select), every one of the 256 images against the oracle, both
    the routed op and the round-1 select + dense assign path; codes-only call as the tokenisation path makes it"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
    from dynamicvectorquantization_amd.router import route_select_dual_entropy
    B, K, D = 256, 1024, 256
    E = synth.codebook_trained(K, D)
    hf = synth.z_tokens(E, B, 32, 32, 2903)
    hco = synth.z_tokens(E, B, 16, 16, 2913)
    ent = synth.entropy_map(5903, B, 16, 16)
    t = _t(dev)
    thf, thc, tent, tE = t(hf), t(hco), t(ent), t(E)
    og = oracle_mod.entropy_gate(ent, THR)
    o_sel = oracle_mod.route_select_dual(og, hco, hf)
    o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
    prep = _CodebookPrep()
    r = vq_assign_routed_dual(thc, thf, tE, prep, entropy=tent, threshold=THR)
    _check_dual(r, o_sel, o, B, oracle_mod=oracle_mod)
    assert np.array_equal(r["gate"].cpu().numpy(), og)
    queued, listed = prep.fallback_count()
    assert 0 < queued < 0.2 * B * 1024 and listed == 0, (queued, listed)
    sel = route_select_dual_entropy(tent, THR, thc, thf)
    zq, codes, loss = vq_assign(sel["h_dual"], tE, _CodebookPrep(), sel["codebook_mask"])
    assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(zq.cpu().numpy(), o["zq"])
    assert torch.equal(zq, r["zq"]) and torch.equal(codes, r["codes"])
    assert C.loss_close(float(loss[1]), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    r2 = vq_assign_routed_dual(thc, thf, tE, _CodebookPrep(), entropy=tent, threshold=THR, want_zq=False, want_loss=False)
    assert torch.equal(r2["codes"], r["codes"]) and torch.equal(r2["indices"], r["indices"])


def test_triple_encode_per_rank_size(dev, oracle_mod):
    """BASELINE configs[3] at the per-rank size (B = 128 of the 8-GPU job's 1024): the
    triple encode glue (fused feature-router gate -> routed assign) against the oracle given the router's
    logits, all images"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_triple
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import TripleGrainFeatureRouter
    B, K, D = 128, 1024, 256
    E = synth.codebook_trained(K, D)
    hf, hm, hco = (synth.z_tokens(E, B, 32, 32, 2104), synth.z_tokens(E, B, 16, 16, 2114), synth.z_tokens(E, B, 8, 8, 2124))
    t = _t(dev)
    router = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu")
    sd = {k: torch.from_numpy(synth.seeded_param(6104, i, k, tuple(v.shape)))
          for i, (k, v) in enumerate(router.state_dict().items())}
    router.load_state_dict(sd)
    router = router.to(dev).eval()
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    with torch.no_grad():
        quant, emb_loss, info, grain, gate = encode_triple(router, vq, t(hf), t(hm), t(hco))
    lg = gate.permute(0, 2, 3, 1).contiguous().cpu().numpy()
    assert tuple(gate.shape) == (B, 3, 8, 8)
    o_sel = oracle_mod.route_select_triple(lg, hco, hm, hf)
    o = oracle_mod.vq_assign_nchw(o_sel["h_triple"], E, o_sel["codebook_mask"])
    assert np.array_equal(grain.cpu().numpy(), o_sel["indices"])
    assert np.array_equal(info[2].cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(quant.cpu().numpy(), o["zq"])
    assert C.loss_close(float(emb_loss), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    counts = np.bincount(o_sel["indices"].reshape(-1), minlength=3)
    assert counts.min() > 0.1 * counts.sum()                      # the router really mixes the three grains


@pytest.mark.parametrize("fixture", ["vq2_K16384_B128_crc", "vq2_K16384_B512_crc"])
def test_k16384_dispatch_size_vs_oracle_and_golden(dev, oracle_mod, fixture):
    """K = 16384, D = 256 through DVQ_MODE_FILTER: the wide (two-blocks-per-wave) pass-1 kernel + 8-slice resolver, at
    N = 131072 tokens (B = 128, the smallest size that dispatches them) and at BASELINE configs[4]'s full B = 512.
    All images against the golden CRCs captured from the imported reference; 8 whole images against the oracle"""
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    g = C.load(fixture)
    z, E, mask = C.vq2_inputs(g)
    B = int(g["B"])
    assert B * 1024 >= 131072 and int(g["K"]) == 16384
    prep = _CodebookPrep()
    t = _t(dev)
    zq, codes, loss = vq_assign(t(z), t(E), prep, t(mask), mode=_lib.MODE_FILTER)
    queued, listed = prep.fallback_count()
    assert queued > 0                                             # the sliced resolver had work
    codes_np, zq_np = codes.cpu().numpy(), zq.cpu().numpy()
    assert np.array_equal(codes_np[0], g["codes_image0"].astype(np.int64))
    assert np.array_equal(C.per_image_crc(codes_np), g["codes_crc"])
    assert np.array_equal(C.per_image_crc(zq_np), g["zq_crc"])
    assert C.loss_close(float(loss[1]), g["loss"])
    sel = np.arange(0, B, B // 8)                                 # 8 whole images
    o = oracle_mod.vq_assign_nchw(z[sel], E, mask[sel])
    assert np.array_equal(codes_np[sel].reshape(len(sel), -1), o["codes"]) and np.array_equal(zq_np[sel], o["zq"])


def test_dual_feature_router_config_end_to_end(dev, oracle_mod):
    """BASELINE configs[1] (DQ-VAE dual, feature router, F = 16 / 8, B = 64) end to end through the encode glue: fused
    feature-router gate -> routed op (no quant_conv), and -> select + 1x1 quant_conv kernel -> dense assign (the
    order of every reference checkpoint, dqvae_dual_feat.py:59-68).  Every image against the oracle GIVEN the
    router's logits (the gate is a 1e-4-tolerance kernel with its own tests) and, on the conv path, given the kernel's h"""
    from dynamicvectorquantization_amd import synth, qconv
    from dynamicvectorquantization_amd.encode import encode_dual
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter
    B, K, D = 64, 1024, 256
    E = synth.codebook_trained(K, D)
    hf, hco = synth.z_tokens(E, B, 32, 32, 3002), synth.z_tokens(E, B, 16, 16, 3012)
    t = _t(dev)
    router = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu")
    sd = {k: torch.from_numpy(synth.seeded_param(6002, i, k, tuple(v.shape)))
          for i, (k, v) in enumerate(router.state_dict().items())}
    router.load_state_dict(sd)
    router = router.to(dev).eval()
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    thf, thc = t(hf), t(hco)
    with torch.no_grad():
        quant, emb_loss, info, grain, gate = encode_dual(router, vq, thf, thc)
    assert tuple(gate.shape) == (B, 2, 16, 16)
    lg = gate.permute(0, 2, 3, 1).contiguous().cpu().numpy()
    o_sel = oracle_mod.route_select_dual(lg, hco, hf)
    o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
    assert np.array_equal(grain.cpu().numpy(), o_sel["indices"])
    assert np.array_equal(info[2].cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(quant.cpu().numpy(), o["zq"])
    assert C.loss_close(float(emb_loss), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    counts = np.bincount(o_sel["indices"].reshape(-1), minlength=2)
    assert counts.min() > 0.1 * counts.sum()                      # the router mixes the grains
    # the model's order: select -> quant_conv -> quantizer
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(t(synth.normal(6012, (D, D, 1, 1), 0.0, 1.0 / 16.0)))
        conv.bias.copy_(t(synth.normal(6013, (D,), 0.0, 0.1)))
        q2, l2, info2, grain2, gate2 = encode_dual(router, vq, thf, thc, quant_conv=conv)
        sel = qconv.quant_conv_select(conv, thc, thf, gate=gate2.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(grain2, grain) and torch.equal(gate2, gate)
    h = sel["h"].cpu().numpy()                                    # the kernel's h: codes / z_q are exact GIVEN it
    o2 = oracle_mod.vq_assign_nchw(h, E, o_sel["codebook_mask"])
    assert np.array_equal(info2[2].cpu().numpy().reshape(B, -1), o2["codes"])
    assert np.array_equal(q2.cpu().numpy(), o2["zq"])
    assert C.loss_close(float(l2), oracle_mod.vq_loss(o2["sqerr"], o2["numel"], 0.25))
    # and h itself against the conv in float64 (contract 1e-5 * sum |w||x|)
    w64 = conv.weight.detach().double().cpu().numpy()[:, :, 0, 0]
    x64 = o_sel["h_dual"].astype(np.float64)[:4]
    ref = np.einsum("ok,bkhw->bohw", w64, x64) + conv.bias.detach().double().cpu().numpy()[None, :, None, None]
    bound = np.einsum("ok,bkhw->bohw", np.abs(w64), np.abs(x64))
    assert np.all(np.abs(h[:4] - ref) <= 1e-5 * bound + 1e-30)


@pytest.mark.parametrize("with_conv", [False, True])
def test_routed_op_is_graph_capturable(dev, with_conv):
    """the routed op -- and the model-order op with the 1x1 quant_conv fused in -- captured into a hipGraph and replayed on
    new inputs: no host synchronisation, no allocation outside the capture's pool, same bits as an eager call"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
    B, K, D = 4, 1024, 256
    En = synth.codebook_trained(K, D)
    t = _t(dev)
    E = t(En)
    mk = lambda seed: (t(synth.z_tokens(En, B, 32, 32, seed)), t(synth.z_tokens(En, B, 16, 16, seed + 1)),
                       t(synth.entropy_map(seed + 2, B, 16, 16)))
    hf, hc, ent = mk(9300)
    prep = _CodebookPrep()
    out = (torch.empty_like(hf), torch.empty((B, 32, 32), dtype=torch.int64, device=dev), torch.empty(2, device=dev),
           torch.empty((B, 16, 16), dtype=torch.int64, device=dev), torch.empty((B, 1, 32, 32), device=dev),
           torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev))
    conv = None
    if with_conv:
        conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
        with torch.no_grad():
            conv.weight.copy_(t(synth.normal(9310, (D, D, 1, 1), 0.0, 1.0 / 16.0)))
    step = lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, out=out, conv=conv)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    hf2, hc2, ent2 = mk(9400)
    hf.copy_(hf2); hc.copy_(hc2); ent.copy_(ent2)
    g.replay()
    torch.cuda.synchronize()
    got = [x.clone() for x in out]
    step()
    torch.cuda.synchronize()
    for a, b in zip(got, out):
        assert torch.equal(a, b)


def test_encode_dual_uses_routed_op_and_matches_select_path(dev, oracle_mod, golden_dir):
    """encode glue: eval + no quant_conv -> routed op; a quant_conv (identity here) or autograd -> select + dense
    assign; identical outputs"""
    import os
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, DualGrainFixedEntropyRouter
    B, K, D = 6, 1024, 256
    E = synth.codebook_trained(K, D)
    t = _t(dev)
    hf, hc = t(synth.z_tokens(E, B, 32, 32, 2203)), t(synth.z_tokens(E, B, 16, 16, 2213))
    ent = t(synth.entropy_map(5203, B, 16, 16))
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    r_ent = DualGrainFixedEntropyRouter(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    torch.manual_seed(5)
    r_feat = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
    for router, e in ((r_ent, ent), (r_feat, None)):
        with torch.no_grad():
            a = encode_dual(router, vq, hf, hc, entropy=e)
            b = encode_dual(router, vq, hf, hc, entropy=e, quant_conv=torch.nn.Identity())
        assert torch.equal(a[0], b[0]) and torch.equal(a[2][2], b[2][2]) and torch.equal(a[3], b[3])
        assert torch.equal(a[4], b[4]) and a[4].shape == (B, 2, 16, 16)
        assert abs(float(a[1]) - float(b[1])) <= 1e-6 * abs(float(b[1]))
    # odd grid widths (240-px images: 15 x 15 cells) go through the routed op too
    hf15, hc15 = t(synth.z_tokens(E, 2, 30, 30, 2223)), t(synth.z_tokens(E, 2, 15, 15, 2233))
    ent15 = t(synth.entropy_map(5213, 2, 15, 15))
    with torch.no_grad():
        a = encode_dual(r_ent, vq, hf15, hc15, entropy=ent15)
    og = oracle_mod.entropy_gate(ent15.cpu().numpy(), r_ent.fine_grain_threshold)
    o_sel = oracle_mod.route_select_dual(og, hc15.cpu().numpy(), hf15.cpu().numpy())
    o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
    assert np.array_equal(a[2][2].cpu().numpy().reshape(2, -1), o["codes"]) and np.array_equal(a[0].cpu().numpy(), o["zq"])
    assert np.array_equal(a[3].cpu().numpy(), o_sel["indices"])
    hf_g = hf.clone().requires_grad_(True)
    q, loss, _, _, _ = encode_dual(r_ent, vq, hf_g, hc, entropy=ent)          # autograd -> differentiable path
    (q.sum() + loss).backward()
    assert hf_g.grad is not None and float(hf_g.grad.abs().sum()) > 0


def test_two_host_threads_two_streams(dev, oracle_mod):
    """include/dvq.h promises re-entrancy (no mutable process-wide state; the last-error string is thread-local): two Python
    threads drive `dvq_vq_assign_routed_dual_f32` concurrently, each on its own HIP stream with its own inputs, SHARING one
    quantizer prep object (image built once, workspaces per stream).  Every call of both threads must equal the oracle."""
    import threading
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
    B, K, D, ROUNDS = 8, 1024, 256, 12
    E = synth.codebook_trained(K, D)
    t = _t(dev)
    tE = t(E)
    prep = _CodebookPrep()
    prep.get(tE)                                  # build the image once on the caller's stream
    torch.cuda.synchronize()
    inputs, expect = [], []
    for i in range(2):
        hf, hco = synth.z_tokens(E, B, 32, 32, 7100 + i), synth.z_tokens(E, B, 16, 16, 7110 + i)
        ent = synth.entropy_map(7120 + i, B, 16, 16)
        og = oracle_mod.entropy_gate(ent, THR)
        o_sel = oracle_mod.route_select_dual(og, hco, hf)
        o = oracle_mod.vq_assign_nchw(o_sel["h_dual"], E, o_sel["codebook_mask"])
        inputs.append((t(hco), t(hf), t(ent)))
        expect.append((o_sel, o))
    results = [[], []]
    errors = []
    start = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(dev)
            start.wait()
            with torch.cuda.stream(st):
                for _ in range(ROUNDS):
                    thc, thf, tent = inputs[i]
                    results[i].append(vq_assign_routed_dual(thc, thf, tE, prep, entropy=tent, threshold=THR))
            st.synchronize()
        except Exception as e:                     # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for i in range(2):
        assert len(results[i]) == ROUNDS
        for r in results[i]:
            _check_dual(r, expect[i][0], expect[i][1], B, oracle_mod=oracle_mod)
    # an invalid call on one thread leaves its message on THAT thread only
    from dynamicvectorquantization_amd import _lib
    msgs = {}

    def bad(i):
        if i == 0:
            _lib.lib.dvq_vq_assign_nchw_f32(0, 0, 0, 0, 1, 256, 1, 1, 0.25, 0, 0, 0, 0, 0, 0, 0)
        msgs[i] = _lib.lib.dvq_last_error_string()

    th = [threading.Thread(target=bad, args=(i,)) for i in range(2)]
    th[0].start(); th[0].join(); th[1].start(); th[1].join()
    assert b"null" in msgs[0] and b"null" not in msgs[1]
