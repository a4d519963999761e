"""GPU (-m gpu): the 1x1 quant_conv FOLDED into the codebook (csrc/vq_fold.hip; dvq_vq_assign_*fold*) --
the opt-in, loss-free form of the model order (stage 2's tokenisation: models/stage2_dynamic/dqtransformer_uncond_entropy.py:166-171
around models/stage1_dynamic/dqvae_dual_entropy.py:124-134).  Pass 1 scores the conv's INPUT against E W and computes no conv.
Contract under test:
  * codes == dvq_qconv_f32 followed by the bit-exact assign, for EVERY token (decided ones by the bound's theorem, undecided
    ones because resolver and exact-list kernel compute that very h);
  * versus the conv-then-quantize order evaluated in float64: equal except at near-ties inside the conv tolerance;
  * z_q within 1e-6 relative of codebook[code]; by-products of the routed forms bit-exact;
  * the bound itself, audited in float64 on the folded scores (tools/bound_audit.py --fold)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
THR = 1.6777750253677368


def _conv(dev, D, seed, bias=True, scale=1.0, ortho=False):
    from dynamicvectorquantization_amd import synth
    conv = torch.nn.Conv2d(D, D, 1, bias=bias)
    w = synth.normal(seed, (D, D), 0.0, scale / np.sqrt(D))
    if ortho:
        w = np.linalg.qr(synth.normal(seed, (D, D)).astype(np.float64))[0].astype(np.float32) * np.float32(scale)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(np.ascontiguousarray(w).reshape(D, D, 1, 1)))
        if bias:
            conv.bias.copy_(torch.from_numpy(synth.normal(seed + 1, (D,), 0.0, 0.1)))
    return conv.to(dev).eval()


def _preimage(conv, h):
    """x with conv(x) ~ h (float64 solve): puts the quantizer's input on the usual token distribution"""
    w = conv.weight.detach().double().cpu().numpy()[:, :, 0, 0]
    b = conv.bias.detach().double().cpu().numpy() if conv.bias is not None else np.zeros(w.shape[0])
    B, D = h.shape[:2]
    hh = h.astype(np.float64).reshape(B, D, -1) - b[None, :, None]
    x = np.einsum("ok,bkn->bon", np.linalg.inv(w), hh)
    return np.ascontiguousarray(x.reshape(h.shape).astype(np.float32))


def _zq_ok(zq, E, codes):
    e = E[codes.reshape(codes.shape[0], -1)]                       # [B, HW, D]
    ref = np.moveaxis(e, 2, 1).reshape(zq.shape)
    return np.all(np.abs(zq - ref) <= 1e-6 * np.maximum(1.0, np.abs(ref)))


@pytest.mark.parametrize("D,K,B,H,W,ortho", [(256, 1024, 3, 16, 16, True), (256, 1024, 2, 32, 32, False), (256, 100, 2, 7, 9, False),
                                             (256, 2048, 1, 16, 16, True), (128, 512, 2, 8, 8, False), (64, 333, 2, 5, 13, True)])
def test_fold_dense_equals_qconv_then_assign(dev, oracle_mod, D, K, B, H, W, ortho):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.qconv import quant_conv
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D, seed=600 + D)
    conv = _conv(dev, D, 610 + D + K, ortho=ortho)
    x = _preimage(conv, synth.z_tokens(E, B, H, W, 620 + K))
    prep = _CodebookPrep()
    zq, codes, loss = vq_assign(t(x), t(E), prep, None, want_loss=False, conv=conv, fold=True)
    torch.cuda.synchronize()
    queued, listed = prep.fallback_count()
    h = quant_conv(conv, t(x))
    zq0, codes0, _ = vq_assign(h, t(E), _CodebookPrep(), None)
    assert loss is None
    assert torch.equal(codes, codes0), int((codes != codes0).sum())
    assert _zq_ok(zq.cpu().numpy(), E, codes.cpu().numpy())
    assert np.all(np.abs(zq.cpu().numpy() - zq0.cpu().numpy()) <= 1e-6 * np.maximum(1.0, np.abs(zq0.cpu().numpy())))
    o = oracle_mod.vq_assign_nchw(h.cpu().numpy(), E, None)
    assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"])
    assert queued < 0.5 * B * H * W, (queued, listed)             # the filter decides most tokens without any conv
    # codes only
    _, codes1, _ = vq_assign(t(x), t(E), prep, None, want_zq=False, want_loss=False, conv=conv, fold=True)
    assert torch.equal(codes1, codes)
    # versus the float64 conv: near-ties only
    w64 = conv.weight.detach().double().cpu().numpy()[:, :, 0, 0]
    b64 = conv.bias.detach().double().cpu().numpy()
    h64 = np.einsum("ok,bk...->bo...", w64, x.astype(np.float64)) + b64.reshape((1, -1) + (1,) * (x.ndim - 2))
    o64 = oracle_mod.vq_assign_nchw(h64.astype(np.float32), E, None)
    got, ref = codes.cpu().numpy().reshape(-1), o64["codes"].reshape(-1)
    rate = float((got == ref).mean())
    assert rate > 0.995, rate
    bad = np.nonzero(got != ref)[0]
    if bad.size:
        hb = np.moveaxis(h64.reshape(B, D, -1), 1, 2).reshape(-1, D)[bad]
        d_g = ((hb - E[got[bad]].astype(np.float64)) ** 2).sum(1)
        d_r = ((hb - E[ref[bad]].astype(np.float64)) ** 2).sum(1)
        assert np.all(np.abs(d_g - d_r) <= 1e-4 * np.maximum(d_r, 1.0)), (d_g - d_r)


def test_fold_routed_dual_and_triple_equal_the_fused_conv_op(dev, oracle_mod):
    """32-wide grids (LDS-staged select) and ragged grids (per-lane select); entropy gate, f32 logits; triple"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual, vq_assign_routed_triple
    D, K = 256, 1024
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 700, ortho=True)
    for B, hc, wc in ((5, 16, 16), (3, 5, 7)):
        hf = _preimage(conv, synth.z_tokens(E, B, 2 * hc, 2 * wc, 701 + hc))
        hco = _preimage(conv, synth.z_tokens(E, B, hc, wc, 702 + hc))
        ent = synth.entropy_map(703 + hc, B, hc, wc)
        ent[0, 0, 0] = np.float32(THR)
        prep = _CodebookPrep()
        r = vq_assign_routed_dual(t(hco), t(hf), t(E), prep, entropy=t(ent), threshold=THR, want_loss=False, conv=conv, fold=True)
        r0 = vq_assign_routed_dual(t(hco), t(hf), t(E), _CodebookPrep(), entropy=t(ent), threshold=THR, conv=conv)
        for k in ("codes", "indices", "codebook_mask", "gate"):
            assert torch.equal(r[k], r0[k]), k
        assert r["loss"] is None and _zq_ok(r["zq"].cpu().numpy(), E, r["codes"].cpu().numpy())
        og = oracle_mod.entropy_gate(ent, THR)
        osel = oracle_mod.route_select_dual(og, hco, hf)
        assert np.array_equal(r["indices"].cpu().numpy(), osel["indices"]) and np.array_equal(r["gate"].cpu().numpy(), og)
        assert np.array_equal(r["codebook_mask"].cpu().numpy(), osel["codebook_mask"])
        lg = synth.normal(704 + hc, (B, hc, wc, 2))
        r = vq_assign_routed_dual(t(hco), t(hf), t(E), prep, gate=t(lg), want_zq=False, want_loss=False, conv=conv, fold=True)
        r0 = vq_assign_routed_dual(t(hco), t(hf), t(E), _CodebookPrep(), gate=t(lg), conv=conv)
        assert torch.equal(r["codes"], r0["codes"]) and torch.equal(r["indices"], r0["indices"]) and r["zq"] is None
    for B, hc, wc in ((4, 8, 8), (2, 3, 5)):
        hf = _preimage(conv, synth.z_tokens(E, B, 4 * hc, 4 * wc, 711 + hc))
        hm = _preimage(conv, synth.z_tokens(E, B, 2 * hc, 2 * wc, 712 + hc))
        hco = _preimage(conv, synth.z_tokens(E, B, hc, wc, 713 + hc))
        lg = synth.grain_logits_triple(714 + hc, B, hc, wc)
        r = vq_assign_routed_triple(t(hco), t(hm), t(hf), t(E), _CodebookPrep(), t(lg), want_loss=False, conv=conv, fold=True)
        r0 = vq_assign_routed_triple(t(hco), t(hm), t(hf), t(E), _CodebookPrep(), t(lg), conv=conv)
        for k in ("codes", "indices", "codebook_mask"):
            assert torch.equal(r[k], r0[k]), k
        assert _zq_ok(r["zq"].cpu().numpy(), E, r["codes"].cpu().numpy())


def test_fold_special_values_and_degenerate_codebooks(dev, oracle_mod):
    """tokens pass 1 cannot score (NaN / Inf / huge inputs) and tokens the resolver cannot resolve (a codebook of 32 vectors x 32
    copies: every token a 32-way tie, 1024 candidate pairs per resolver group, shard overflow) go to the exact-list kernel, which
    computes their conv output itself: same codes as dvq_qconv_f32 + assign"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.qconv import quant_conv
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
    D, K, B, H, W = 256, 1024, 2, 16, 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 800)
    x = _preimage(conv, synth.z_tokens(E, B, H, W, 801))
    x[0, 3, 0, 0] = np.nan
    x[0, :, 1, 1] = np.inf
    x[1, 7, 2, 2] = -np.inf
    x[1, :, 3, 3] = np.float32(7e4)
    x[1, :, 4, 4] = 0.0
    x[0, 11, 5, 5] = np.float32(3.3e38)
    prep = _CodebookPrep()
    zq, codes, _ = vq_assign(t(x), t(E), prep, None, want_loss=False, conv=conv, fold=True)
    torch.cuda.synchronize()
    queued, listed = prep.fallback_count()
    assert listed >= 5, (queued, listed)
    h = quant_conv(conv, t(x))
    zq0, codes0, _ = vq_assign(h, t(E), _CodebookPrep(), None)
    assert torch.equal(codes, codes0)
    fin = torch.isfinite(zq0)
    assert torch.equal(torch.isfinite(zq), fin)
    assert torch.all((zq - zq0).abs()[fin] <= 1e-6 * torch.clamp(zq0.abs()[fin], min=1.0))
    # degenerate codebook: dense and routed
    Ed = np.ascontiguousarray(np.tile(synth.codebook_trained(32, D, seed=811), (32, 1)))
    xd = _preimage(conv, synth.z_tokens(Ed, 4, 16, 16, 812))
    prep = _CodebookPrep()
    _, cd, _ = vq_assign(t(xd), t(Ed), prep, None, want_zq=False, want_loss=False, conv=conv, fold=True)
    torch.cuda.synchronize()
    queued, listed = prep.fallback_count()
    assert listed > 0, (queued, listed)
    _, cd0, _ = vq_assign(quant_conv(conv, t(xd)), t(Ed), _CodebookPrep(), None, want_zq=False, want_loss=False)
    assert torch.equal(cd, cd0) and int(cd.max()) < 32
    hf, hco = _preimage(conv, synth.z_tokens(Ed, 3, 32, 32, 813)), _preimage(conv, synth.z_tokens(Ed, 3, 16, 16, 814))
    ent = synth.entropy_map(815, 3, 16, 16)
    r = vq_assign_routed_dual(t(hco), t(hf), t(Ed), _CodebookPrep(), entropy=t(ent), threshold=THR, want_loss=False, conv=conv, fold=True)
    r0 = vq_assign_routed_dual(t(hco), t(hf), t(Ed), _CodebookPrep(), entropy=t(ent), threshold=THR, conv=conv)
    assert torch.equal(r["codes"], r0["codes"])
    # non-finite conv weight: every token by the exact list (the fold meta says so), still the same answer as the unfused order
    convn = _conv(dev, D, 820)
    with torch.no_grad():
        convn.weight[5, 6, 0, 0] = float("inf")
    xs = synth.normal(821, (1, D, 4, 8))
    _, cn, _ = vq_assign(t(xs), t(E), _CodebookPrep(), None, want_zq=False, want_loss=False, conv=convn, fold=True)
    _, cn0, _ = vq_assign(quant_conv(convn, t(xs)), t(E), _CodebookPrep(), None, want_zq=False, want_loss=False)
    assert torch.equal(cn, cn0)


def test_fold_rebuilds_when_codebook_or_conv_changes(dev):
    from dynamicvectorquantization_amd import synth, qconv
    from dynamicvectorquantization_amd.qconv import quant_conv
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    D, K = 256, 512
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = t(synth.codebook_trained(K, D, seed=900))
    conv = _conv(dev, D, 901)
    x = t(synth.normal(902, (2, D, 8, 8)))
    prep = _CodebookPrep()
    ref = lambda: vq_assign(quant_conv(conv, x), E, _CodebookPrep(), None, want_zq=False, want_loss=False)[1]
    c1 = vq_assign(x, E, prep, None, want_zq=False, want_loss=False, conv=conv, fold=True)[1]
    assert torch.equal(c1, ref())
    with torch.no_grad():
        conv.weight.mul_(-1.0)                      # in-place: bumps the version counter
    c2 = vq_assign(x, E, prep, None, want_zq=False, want_loss=False, conv=conv, fold=True)[1]
    assert torch.equal(c2, ref()) and not torch.equal(c1, c2)
    with torch.no_grad():
        E.mul_(0.5)
    c3 = vq_assign(x, E, prep, None, want_zq=False, want_loss=False, conv=conv, fold=True)[1]
    assert torch.equal(c3, ref())
    conv.weight.data.mul_(2.0)                      # through .data: the caller invalidates
    qconv.invalidate(conv)
    prep.invalidate()
    c4 = vq_assign(x, E, prep, None, want_zq=False, want_loss=False, conv=conv, fold=True)[1]
    assert torch.equal(c4, ref())
    with pytest.raises(ValueError):
        vq_assign(x, E, prep, None, conv=conv, fold=True)           # a loss cannot be had from the fold


def test_fold_bound_holds_with_margin(dev):
    """|G'_j - truth_j(h)| <= W' in float64 for every (token, code) and for h = fp32(conv64), h = dvq_qconv_f32(x) and two h at
    the edge of the conv tolerance ball; no decided token disagrees with the reference argmin on any of those h"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bound_audit", os.path.join(root, "tools", "bound_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run_fold(48, verbose=False)
    assert len(res) >= 6
    for r in res:
        assert r["max_err_over_W"] <= 1.0, r
        assert r["decided_but_wrong"] == 0, r
    assert sum(r["decided"] for r in res) > 0
    print("fold: max |G' - truth| / W' over all cases: %.4f" % max(r["max_err_over_W"] for r in res))


def test_encode_to_tokens_fold_matches_the_fused_conv_path(dev):
    """stage 2's tokenisation behind the stage-1 quant_conv: the fold form gives the same token streams"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_to_tokens
    from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    D, K, B = 256, 1024, 6
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 950, ortho=True)
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    router = DualGrainFixedEntropyRouter(os.path.join(gd, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    perm = DualGrainSeperatePermuter()
    hf = t(_preimage(conv, synth.z_tokens(E, B, 32, 32, 951)))
    hc = t(_preimage(conv, synth.z_tokens(E, B, 16, 16, 952)))
    ent = t(synth.entropy_map(953, B, 16, 16))
    with torch.no_grad():
        s0, g0, c0 = encode_to_tokens(router, vq, perm, hf, hc, entropy=ent, quant_conv=conv)
        s1, g1, c1 = encode_to_tokens(router, vq, perm, hf, hc, entropy=ent, quant_conv=conv, fold=True)
    assert torch.equal(c0, c1) and torch.equal(g0, g1)
    for k in s0:
        if torch.is_tensor(s0[k]):
            assert torch.equal(s0[k], s1[k]), k
