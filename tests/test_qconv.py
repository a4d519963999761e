"""GPU (-m gpu): the 1x1 quant_conv kernels (second half): `dvq_qconv_f32` and the
select-fused `dvq_qconv_select_f32`.  Tolerance contract (a GEMM in a different summation order than the
reference's MIOpen / oneDNN conv): |h - h_fp64| <= 1e-5 * sum_i |w_oi| |x_i| per element (in practice ~1e-7);
the select's by-products bit-exact vs the oracle; the assign downstream bit-exact GIVEN the kernel's h; code
match rate against the conv-then-quantize order evaluated in float64 reported and > 99.5 %."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
THR = 1.6777750253677368


def _conv(dev, D, seed, bias=True, scale=1.0):
    from dynamicvectorquantization_amd import synth
    conv = torch.nn.Conv2d(D, D, 1, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.normal(seed, (D, D, 1, 1), 0.0, scale / np.sqrt(D))))
        if bias:
            conv.bias.copy_(torch.from_numpy(synth.normal(seed + 1, (D,), 0.0, 0.1)))
    return conv.to(dev).eval()


def _ref64(conv, x):
    """fp64 conv and the per-element magnitude sum_i |w||x| (+|b|) the tolerance is relative to"""
    w = conv.weight.detach().double().cpu().numpy()[:, :, 0, 0]
    b = conv.bias.detach().double().cpu().numpy() if conv.bias is not None else np.zeros(w.shape[0])
    x64 = x.astype(np.float64)
    h = np.einsum("oi,bi...->bo...", w, x64) + b.reshape((1, -1) + (1,) * (x.ndim - 2))
    mag = np.einsum("oi,bi...->bo...", np.abs(w), np.abs(x64)) + np.abs(b).reshape((1, -1) + (1,) * (x.ndim - 2))
    return h, mag


@pytest.mark.parametrize("D,B,H,W", [(256, 3, 7, 9), (256, 2, 32, 32), (128, 2, 5, 4), (64, 1, 1, 33)])
def test_qconv_dense_vs_fp64(dev, D, B, H, W):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.qconv import quant_conv
    conv = _conv(dev, D, 300 + D)
    x = synth.normal(310 + D, (B, D, H, W), 0.0, 1.5)
    x[0, :, 0, 0] *= np.float32(1e4)                     # per-token scaling: large ...
    x[-1, :, -1, -1] *= np.float32(1e-6)                 # ... and tiny tokens keep their relative accuracy
    x[0, :, 0, W - 1] = 0.0
    h = quant_conv(conv, torch.from_numpy(x).to(dev)).cpu().numpy()
    ref, mag = _ref64(conv, x)
    err = np.abs(h - ref) / mag
    assert err.max() < 1e-5, err.max()
    assert err.max() < 2e-6                               # what the split-fp16 scheme actually delivers
    with torch.no_grad():
        t = conv(torch.from_numpy(x).to(dev)).cpu().numpy()   # the vendor conv, same tolerance
    assert (np.abs(h - t) / mag).max() < 1e-5


def test_qconv_select_dual_and_triple(dev, oracle_mod):
    """select fused in: by-products bit-exact, h = conv(oracle select) within the tolerance, both gate kinds"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.qconv import quant_conv_select
    D, B = 256, 5
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    conv = _conv(dev, D, 400)
    hf, hc = synth.features(401, B, D, 32, 32), synth.features(402, B, D, 16, 16)
    ent = synth.entropy_map(403, B, 16, 16)
    ent[0, 0, 0] = np.float32(THR); ent[1, 2, 3] = np.nan
    r = quant_conv_select(conv, t(hc), t(hf), entropy=t(ent), threshold=THR)
    og = oracle_mod.entropy_gate(ent, THR)
    o = oracle_mod.route_select_dual(og, hc, hf)
    assert np.array_equal(r["indices"].cpu().numpy(), o["indices"]) and np.array_equal(r["gate"].cpu().numpy(), og)
    assert np.array_equal(r["codebook_mask"].cpu().numpy(), o["codebook_mask"])
    ref, mag = _ref64(conv, o["h_dual"])
    assert (np.abs(r["h"].cpu().numpy() - ref) / mag).max() < 2e-6
    lg2 = synth.normal(404, (B, 16, 16, 2))
    r = quant_conv_select(conv, t(hc), t(hf), gate=t(lg2))
    o = oracle_mod.route_select_dual(lg2, hc, hf)
    ref, mag = _ref64(conv, o["h_dual"])
    assert np.array_equal(r["indices"].cpu().numpy(), o["indices"]) and (np.abs(r["h"].cpu().numpy() - ref) / mag).max() < 2e-6
    # triple, ragged grid, no bias
    conv3 = _conv(dev, D, 410, bias=False)
    hf3, hm3, hc3 = synth.features(411, 3, D, 12, 20), synth.features(412, 3, D, 6, 10), synth.features(413, 3, D, 3, 5)
    lg = synth.grain_logits_triple(414, 3, 3, 5)
    r = quant_conv_select(conv3, t(hc3), t(hf3), h_median=t(hm3), gate=t(lg))
    o = oracle_mod.route_select_triple(lg, hc3, hm3, hf3)
    assert np.array_equal(r["indices"].cpu().numpy(), o["indices"]) and np.array_equal(r["codebook_mask"].cpu().numpy(), o["codebook_mask"])
    ref, mag = _ref64(conv3, o["h_triple"])
    assert (np.abs(r["h"].cpu().numpy() - ref) / mag).max() < 2e-6


def _nan_equal(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("B,H,W", [(3, 7, 9), (2, 32, 32), (9, 16, 16)])
def test_fused_conv_assign_dense(dev, oracle_mod, B, H, W):
    """dvq_vq_assign_qconv_f32: the conv as pass 1's prologue.  With h_buf the op also writes the conv output it scored:
    that h is within the conv's tolerance of the fp64 conv, and codes / z_q / loss are bit-exact GIVEN it (oracle); without
    h_buf (production: h never written) the same bits come out; ragged token counts, masks, scale jumps between k-steps,
    NaN / Inf / huge inputs (exact-list path, reads the spilled h rows)."""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    K, D = 1024, 256
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 600 + B)
    x = synth.z_tokens(E, B, H, W, 610 + B)
    x[0, :, 0, 0] *= np.float32(1e-3)
    x[1, 5, 3, 3] = 40.0                                   # one large channel late in a token: the running scale moves
    x[1, :16, 1, 1] = 1e-6
    x[0, 200:, 2, 2] *= np.float32(1e4)
    mask = np.where(synth.bernoulli(620 + B, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
    hb = torch.empty((B, D, H, W), device=dev)
    zq, codes, loss = vq_assign(t(x), t(E), _CodebookPrep(), t(mask), conv=conv, h_buf=hb)
    h = hb.cpu().numpy()
    ref, mag = _ref64(conv, x)
    assert (np.abs(h - ref) / mag).max() < 2e-6
    o = oracle_mod.vq_assign_nchw(h, E, mask)
    assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(zq.cpu().numpy(), o["zq"])
    ol = float(oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    assert abs(float(loss[1]) - ol) <= 1e-5 * abs(ol)
    zq2, codes2, loss2 = vq_assign(t(x), t(E), _CodebookPrep(), t(mask), conv=conv)          # production form
    assert torch.equal(codes, codes2) and torch.equal(zq, zq2) and torch.equal(loss, loss2)
    zq3, codes3, _ = vq_assign(t(x), t(E), _CodebookPrep(), t(mask), conv=conv, want_zq=False, want_loss=False)
    assert zq3 is None and torch.equal(codes, codes3)
    # special values
    xs = x.copy()
    xs[0, 7, 2, 2] = np.nan; xs[1, :, 5, 5] = np.inf; xs[2 % B, :, 6, 6] *= np.float32(1e30); xs[0, :, 1, 0] = 0.0
    hb2 = torch.empty_like(hb)
    zq4, codes4, _ = vq_assign(t(xs), t(E), _CodebookPrep(), t(mask), conv=conv, h_buf=hb2)
    o4 = oracle_mod.vq_assign_nchw(hb2.cpu().numpy(), E, mask)
    assert np.array_equal(codes4.cpu().numpy().reshape(B, -1), o4["codes"]) and _nan_equal(zq4.cpu().numpy(), o4["zq"])
    zq5, codes5, _ = vq_assign(t(xs), t(E), _CodebookPrep(), t(mask), conv=conv)             # h rows spilled for the exact list
    assert torch.equal(codes4, codes5) and _nan_equal(zq4.cpu().numpy(), zq5.cpu().numpy())


def test_fused_conv_assign_routed_dual_and_triple(dev, oracle_mod):
    """dvq_vq_assign_routed_qconv_{dual,triple}_f32: select -> conv -> assign as one op: the select's by-products bit-exact
    vs the oracle, h = conv(oracle select) within the tolerance, codes / z_q / loss exact given h"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual, vq_assign_routed_triple
    K, D, B = 1024, 256, 6
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 700)
    hf, hc = synth.z_tokens(E, B, 32, 32, 701), synth.z_tokens(E, B, 16, 16, 702)
    ent = synth.entropy_map(703, B, 16, 16)
    ent[0, 0, 0] = np.float32(THR); ent[1, 2, 3] = np.nan
    for gate_kw, og in ((dict(entropy=t(ent), threshold=THR), oracle_mod.entropy_gate(ent, THR)),
                        (dict(gate=t(synth.normal(704, (B, 16, 16, 2)))), synth.normal(704, (B, 16, 16, 2)))):
        hb = torch.empty((B, D, 32, 32), device=dev)
        r = vq_assign_routed_dual(t(hc), t(hf), t(E), _CodebookPrep(), conv=conv, h_buf=hb, **gate_kw)
        osel = oracle_mod.route_select_dual(og, hc, hf)
        assert np.array_equal(r["indices"].cpu().numpy(), osel["indices"])
        assert np.array_equal(r["codebook_mask"].cpu().numpy(), osel["codebook_mask"])
        if "entropy" in gate_kw:
            assert np.array_equal(r["gate"].cpu().numpy(), og)
        h = hb.cpu().numpy()
        ref, mag = _ref64(conv, osel["h_dual"])
        assert (np.abs(h - ref) / mag).max() < 2e-6
        o = oracle_mod.vq_assign_nchw(h, E, osel["codebook_mask"])
        assert np.array_equal(r["codes"].cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(r["zq"].cpu().numpy(), o["zq"])
        ol = float(oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
        assert abs(float(r["loss"][1]) - ol) <= 1e-5 * abs(ol)
        r2 = vq_assign_routed_dual(t(hc), t(hf), t(E), _CodebookPrep(), conv=conv, **gate_kw)
        assert torch.equal(r["codes"], r2["codes"]) and torch.equal(r["zq"], r2["zq"]) and torch.equal(r["loss"], r2["loss"])
    # triple, ragged 12 x 20 grid (odd coarse width), no bias
    conv3 = _conv(dev, D, 710, bias=False)
    hf3, hm3, hc3 = synth.z_tokens(E, 3, 12, 20, 711), synth.z_tokens(E, 3, 6, 10, 712), synth.z_tokens(E, 3, 3, 5, 713)
    lg = synth.grain_logits_triple(714, 3, 3, 5)
    hb = torch.empty((3, D, 12, 20), device=dev)
    r = vq_assign_routed_triple(t(hc3), t(hm3), t(hf3), t(E), _CodebookPrep(), t(lg), conv=conv3, h_buf=hb)
    osel = oracle_mod.route_select_triple(lg, hc3, hm3, hf3)
    assert np.array_equal(r["indices"].cpu().numpy(), osel["indices"]) and np.array_equal(r["codebook_mask"].cpu().numpy(), osel["codebook_mask"])
    h = hb.cpu().numpy()
    ref, mag = _ref64(conv3, osel["h_triple"])
    assert (np.abs(h - ref) / mag).max() < 2e-6
    o = oracle_mod.vq_assign_nchw(h, E, osel["codebook_mask"])
    assert np.array_equal(r["codes"].cpu().numpy().reshape(3, -1), o["codes"]) and np.array_equal(r["zq"].cpu().numpy(), o["zq"])


def test_fused_conv_refuses_what_it_cannot_do(dev):
    from dynamicvectorquantization_amd import _lib, synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    E = torch.from_numpy(synth.codebook_trained(64, 128)).to(dev)
    x = torch.zeros((1, 128, 4, 4), device=dev)
    with pytest.raises(_lib.DvqError):                     # 128 channels: the two-kernel path is the one to use
        vq_assign(x, E, _CodebookPrep(), None, conv=_conv(dev, 128, 800))
    E2 = torch.from_numpy(synth.codebook_trained(64, 256)).to(dev)
    with pytest.raises(_lib.DvqError):                     # exact mode has no conv prologue
        vq_assign(torch.zeros((1, 256, 4, 4), device=dev), E2, _CodebookPrep(), None, conv=_conv(dev, 256, 801), mode=_lib.MODE_EXACT)


def test_encode_with_quant_conv_end_to_end(dev, oracle_mod, golden_dir):
    """the real encode order select -> quant_conv -> quantize (dqvae_dual_feat.py:59-68) through encode_dual, which runs it
    as ONE routed op with the conv as the assign's prologue: equal to that op called directly (whose codes / z_q are exact
    given the h it reports, previous tests); code match rate vs the fp64 conv-then-quantize order; a conv the kernels cannot
    take (not a plain nn.Conv2d) falls back to torch's conv and agrees up to near-ties; 128 channels take the two-kernel path"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual, encode_fixed
    from dynamicvectorquantization_amd.qconv import quant_conv, quant_conv_select
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, _CodebookPrep, vq_assign, vq_assign_routed_dual
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    B, K, D = 8, 1024, 256
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 500, scale=1.0)
    hf, hc = synth.z_tokens(E, B, 32, 32, 501), synth.z_tokens(E, B, 16, 16, 502)
    ent = synth.entropy_map(503, B, 16, 16)
    router = DualGrainFixedEntropyRouter(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    hb = torch.empty((B, D, 32, 32), device=dev)
    with torch.no_grad():
        quant, loss, info, grain, gate = encode_dual(router, vq, t(hf), t(hc), entropy=t(ent), quant_conv=conv)
        r = vq_assign_routed_dual(t(hc), t(hf), t(E), _CodebookPrep(), entropy=t(ent), threshold=router.fine_grain_threshold,
                                  conv=conv, h_buf=hb)
    assert torch.equal(info[2], r["codes"]) and torch.equal(quant, r["zq"]) and float(loss) == float(r["loss"][1])
    h = hb.cpu().numpy()
    o = oracle_mod.vq_assign_nchw(h, E, r["codebook_mask"].cpu().numpy())
    codes = info[2].cpu().numpy().reshape(B, -1)
    assert np.array_equal(codes, o["codes"]) and np.array_equal(quant.cpu().numpy(), o["zq"])
    assert abs(float(loss) - float(oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))) <= 1e-5 * float(loss)
    og = oracle_mod.entropy_gate(ent, router.fine_grain_threshold)
    osel = oracle_mod.route_select_dual(og, hc, hf)
    assert np.array_equal(grain.cpu().numpy(), osel["indices"]) and tuple(gate.shape) == (B, 2, 16, 16)
    # match rate against conv-then-quantize with the conv in float64 (rounded to f32)
    ref, _ = _ref64(conv, osel["h_dual"])
    o64 = oracle_mod.vq_assign_nchw(ref.astype(np.float32), E, osel["codebook_mask"])
    rate = float((codes == o64["codes"]).mean())
    print("code match rate vs fp64 conv-then-quantize: %.5f" % rate)
    assert rate > 0.995
    # the two-kernel path (select + conv kernel, dense assign) agrees up to near-ties of its own h
    with torch.no_grad():
        sel = quant_conv_select(conv, t(hc), t(hf), entropy=t(ent), threshold=router.fine_grain_threshold)
        _, c2, _ = vq_assign(sel["h"], t(E), _CodebookPrep(), sel["codebook_mask"])
    assert float((c2 == info[2]).float().mean()) > 0.995
    # same answer as torch's conv path up to near-ties (a conv the kernels cannot take falls back)
    with torch.no_grad():
        q2, _, info2, _, _ = encode_dual(router, vq, t(hf), t(hc), entropy=t(ent),
                                         quant_conv=torch.nn.Sequential(conv))        # not an nn.Conv2d -> torch path
    assert float((info2[2] == info[2]).float().mean()) > 0.995
    # fixed-granularity model (VQModel.encode): one op as well; equal to the dense fused op, exact given its h
    hb2 = torch.empty_like(hb)
    with torch.no_grad():
        qf, lf, inf_ = encode_fixed(vq, t(hf), quant_conv=conv)
        zqd, cd, ld = vq_assign(t(hf), t(E), _CodebookPrep(), None, conv=conv, h_buf=hb2)
    assert torch.equal(inf_[2], cd) and torch.equal(qf, zqd)
    of = oracle_mod.vq_assign_nchw(hb2.cpu().numpy(), E, None)
    assert np.array_equal(cd.cpu().numpy().reshape(B, -1), of["codes"])
    # ... and with the taming-style quantizer VQModel really has (BASELINE configs[0]: B = 4, 16 x 16; quantize_vqgan.py:271-312):
    # one op too (round 6), flat or [B, H, W] indices, legacy or not; remap / grad-carrying calls keep the two-step path
    from dynamicvectorquantization_amd.quantize import VectorQuantizer2
    z4 = t(synth.z_tokens(E, 4, 16, 16, 505))
    for legacy, sane in ((False, False), (True, True)):
        vqg = VectorQuantizer2(K, D, beta=0.25, legacy=legacy, sane_index_shape=sane).to(dev).eval()
        vqg.embedding.weight.data.copy_(t(E))
        hb4 = torch.empty((4, D, 16, 16), device=dev)
        with torch.no_grad():
            qg, lg, ig = encode_fixed(vqg, z4, quant_conv=conv)
            zq4, c4, l4 = vq_assign(z4, t(E), _CodebookPrep(), None, conv=conv, h_buf=hb4)
            q2s, l2s, i2s = vqg(quant_conv(conv, z4))                 # the two-step order on the same conv kernel's output
        assert tuple(ig[2].shape) == ((4, 16, 16) if sane else (4 * 256,))
        assert torch.equal(ig[2].reshape(-1), c4.reshape(-1)) and torch.equal(qg, zq4) and float(lg) == float(l4[1])
        assert torch.equal(ig[2].reshape(-1), i2s[2].reshape(-1)) and torch.equal(qg, q2s) and abs(float(lg) - float(l2s)) <= 1e-6 * float(l2s)
        o4 = oracle_mod.vq_assign_nchw(hb4.cpu().numpy(), E, None)
        assert np.array_equal(c4.cpu().numpy().reshape(4, -1), o4["codes"]) and np.array_equal(qg.cpu().numpy(), o4["zq"])
        assert abs(float(lg) - float(oracle_mod.vq_loss(o4["sqerr"], o4["numel"], 0.25))) <= 1e-5 * float(lg)
    # 128 channels: select + conv kernel, then the dense assign (exact given that kernel's h)
    E1 = synth.codebook_trained(256, 128)
    conv1 = _conv(dev, 128, 520)
    vq1 = VectorQuantize2(256, 128).to(dev).eval()
    vq1.codebook.weight.data[:-1].copy_(t(E1))
    hf1, hc1 = synth.z_tokens(E1, 2, 16, 16, 521), synth.z_tokens(E1, 2, 8, 8, 522)
    ent1 = synth.entropy_map(523, 2, 8, 8)
    with torch.no_grad():
        q1, _, info1, _, _ = encode_dual(router, vq1, t(hf1), t(hc1), entropy=t(ent1), quant_conv=conv1)
        sel1 = quant_conv_select(conv1, t(hc1), t(hf1), entropy=t(ent1), threshold=router.fine_grain_threshold)
    o1 = oracle_mod.vq_assign_nchw(sel1["h"].cpu().numpy(), E1, sel1["codebook_mask"].cpu().numpy())
    assert np.array_equal(info1[2].cpu().numpy().reshape(2, -1), o1["codes"]) and np.array_equal(q1.cpu().numpy(), o1["zq"])


def test_fused_conv_running_scale_adversarial(dev, oracle_mod):
    """the conv prologue's per-token scale follows the RUNNING maximum over the k-steps (an exact power-of-two rescale of the
    accumulators when a later k-step outgrows it): tokens whose 16-channel groups differ by up to 2^+-30 in either order, groups
    of zeros first / last / everywhere, fp32 subnormals, values near the fp32 maximum (the conv overflows: exact-list path) --
    the conv output stays within 1e-5 * sum |w||x| of the fp64 conv and codes / z_q stay exact given it"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    K, D, B, H, W = 1024, 256, 4, 16, 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 900)
    rng = np.random.default_rng(901)
    x = synth.z_tokens(E, B, H, W, 902)
    n = B * H * W
    ex = rng.integers(-30, 31, size=(n, 16))                       # per token and 16-channel group: 2^ex
    ex[: n // 4] = np.sort(ex[: n // 4], axis=1)                   # monotonically growing (a rescale at almost every k-step)
    ex[n // 4: n // 2] = -np.sort(-ex[n // 4: n // 2], axis=1)     # monotonically shrinking (never a rescale after the first)
    scale = np.repeat(np.ldexp(1.0, ex).astype(np.float32), 16, axis=1).reshape(B, H, W, D).transpose(0, 3, 1, 2)
    x = (x * scale).astype(np.float32)
    x[0, :16, 0, 0] = 0.0                                          # first group zero
    x[0, 240:, 0, 1] = 0.0                                         # last group zero
    x[0, :, 0, 2] = 0.0                                            # an all-zero token: h = bias
    x[0, :, 0, 3] = np.float32(1e-40)                              # subnormal inputs
    x[0, :, 0, 4] = 0.0; x[0, 255, 0, 4] = 3.0                     # everything in the last channel
    x[0, :, 0, 5] = 0.0; x[0, 0, 0, 5] = -7.0                      # everything in the first channel
    hb = torch.empty((B, D, H, W), device=dev)
    zq, codes, loss = vq_assign(t(x), t(E), _CodebookPrep(), None, conv=conv, h_buf=hb)
    h = hb.cpu().numpy()
    ref, mag = _ref64(conv, x)
    assert np.isfinite(h).all()
    err = np.abs(h - ref) / mag
    assert err.max() < 1e-5, err.max()
    assert np.array_equal(h[0, :, 0, 2], conv.bias.detach().cpu().numpy())
    o = oracle_mod.vq_assign_nchw(h, E, None)
    assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(zq.cpu().numpy(), o["zq"])
    # overflow: |x| near the fp32 maximum makes the conv non-finite; the token takes the exact-list path with the spilled h
    xo = x.copy()
    xo[1, :, 3, 3] = np.float32(3e38)
    xo[2, 17, 5, 5] = np.float32(-3.3e38)
    hb2 = torch.empty_like(hb)
    zq2, codes2, _ = vq_assign(t(xo), t(E), _CodebookPrep(), None, conv=conv, h_buf=hb2)
    h2 = hb2.cpu().numpy()
    o2 = oracle_mod.vq_assign_nchw(h2, E, None)
    assert np.array_equal(codes2.cpu().numpy().reshape(B, -1), o2["codes"]) and _nan_equal(zq2.cpu().numpy(), o2["zq"])
    zq3, codes3, _ = vq_assign(t(xo), t(E), _CodebookPrep(), None, conv=conv)
    assert torch.equal(codes2, codes3) and _nan_equal(zq2.cpu().numpy(), zq3.cpu().numpy())


@pytest.mark.parametrize("K", [100, 2048])
def test_fused_conv_other_codebook_sizes(dev, oracle_mod, K):
    """the conv prologue with codebooks that are not 1024 codes: fewer tiles than the ring's prefetch depth needs no special case,
    and a large codebook takes the same (two-workgroups-per-CU) kernel -- exact given h in both cases"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    D, B, H, W = 256, 2, 16, 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = synth.codebook_trained(K, D)
    conv = _conv(dev, D, 950 + K)
    x = synth.z_tokens(E, B, H, W, 951 + K)
    hb = torch.empty((B, D, H, W), device=dev)
    zq, codes, loss = vq_assign(t(x), t(E), _CodebookPrep(), None, conv=conv, h_buf=hb)
    o = oracle_mod.vq_assign_nchw(hb.cpu().numpy(), E, None)
    assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(zq.cpu().numpy(), o["zq"])
    ref, mag = _ref64(conv, x)
    assert (np.abs(hb.cpu().numpy() - ref) / mag).max() < 2e-6


def test_model_order_against_the_reference_models_own_encode(dev, oracle_mod, golden_dir):
    """golden from the IMPORTED reference: `DualGrainVQModel.encode` itself (dqvae_dual_entropy.py:124-134; encoder, fixed-entropy
    router and quantizer instantiated from the reference's YAML, seeded 1x1 quant_conv, trained-like codebook) run on CPU on one
    synthetic image, the two encoder branch outputs captured by hooks (oracle/gen_golden_encode.py).  The fused op on those
    inputs: grain map and gate equal; codes equal except at near-ties of the two convs' roundings (rate reported, > 99.5 %, and
    every differing token's two candidates are within 1e-5 relative in distance given OUR h); loss within 1e-4; z_q within fp16
    of the stored copy where the codes agree"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, _CodebookPrep, vq_assign_routed_dual
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    import zlib
    g = np.load(os.path.join(golden_dir, "encode_dual_entropy_model_B1.npz"))
    crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    cw, cb = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0), synth.normal(9502, (D,), 0.0, 0.1)
    assert crc(E) == g["cb_crc"] and crc(cw) == g["conv_w_crc"] and crc(cb) == g["conv_b_crc"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(t(cw)); conv.bias.copy_(t(cb))
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    router = DualGrainFixedEntropyRouter(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    hf, hc, ent = t(g["h_fine"]), t(g["h_coarse"]), t(g["x_entropy"])
    hb = torch.empty_like(hf)
    with torch.no_grad():
        quant, loss, info, grain, gate = encode_dual(router, vq, hf, hc, entropy=ent, quant_conv=conv)
        r = vq_assign_routed_dual(hc, hf, t(E), _CodebookPrep(), entropy=ent, threshold=router.fine_grain_threshold, conv=conv, h_buf=hb)
    assert torch.equal(info[2], r["codes"])
    assert np.array_equal(grain.cpu().numpy(), g["grain"].astype(np.int64))
    assert np.array_equal(gate.cpu().numpy(), g["gate"].astype(np.int64))
    codes, ref = info[2].cpu().numpy().reshape(-1), g["codes"].astype(np.int64).reshape(-1)
    rate = float((codes == ref).mean())
    print("codes equal to the reference model's encode: %.5f (%d of %d differ)" % (rate, int((codes != ref).sum()), codes.size))
    assert rate > 0.995
    h = hb.cpu().numpy().reshape(D, -1).T.astype(np.float64)              # [tokens, D], the h our op scored
    bad = np.nonzero(codes != ref)[0]
    if bad.size:
        d_ours = ((h[bad] - E[codes[bad]].astype(np.float64)) ** 2).sum(1)
        d_ref = ((h[bad] - E[ref[bad]].astype(np.float64)) ** 2).sum(1)
        assert np.all(np.abs(d_ref - d_ours) <= 1e-5 * np.maximum(d_ours, 1.0)), (d_ref - d_ours)
    assert abs(float(loss) - float(g["emb_loss"])) <= 1e-4 * abs(float(g["emb_loss"]))
    same = (codes == ref).reshape(1, 1, 32, 32).repeat(D, 1)
    q, q16 = quant.cpu().numpy(), g["quant_f16"].astype(np.float32)
    assert np.all(np.abs(q - q16)[same] <= 2e-3 * np.maximum(1.0, np.abs(q16))[same])


@pytest.mark.parametrize("kind", ["dual", "triple"])
def test_feature_router_models_against_the_reference_models_own_encode(dev, golden_dir, kind):
    """the same for the feature-router models: goldens from the reference's `DualGrainVQModel.encode` (dqvae_dual_feat.py:59-68) and
    `TripleGrainVQModel.encode` (dqvae_triple_feat.py:68-77) run on CPU with encoder / router / quantizer from the reference's YAMLs and
    a seeded router MLP: our drop-in router (fused gate kernel) + the fused select -> conv -> assign op give the gate logits within 1e-4,
    the same grain map (all argmax margins of the golden exceed 2e-3), codes equal up to near-ties (> 99.5 %, each differing token's two
    candidates within 1e-5 in distance given our h), loss within 1e-4"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual, encode_triple
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, _CodebookPrep, vq_assign_routed_dual, vq_assign_routed_triple
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    import zlib
    g = np.load(os.path.join(golden_dir, "encode_%s_feature_model_B1.npz" % kind))
    crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    G = 2 if kind == "dual" else 3
    K, D = 1024, 256
    F = G * D
    E = synth.codebook_trained(K, D)
    cw, cb = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0), synth.normal(9502, (D,), 0.0, 0.1)
    w1, b1 = synth.normal(9600 + G, (F, F), 0.0, 1.0 / np.sqrt(F)), synth.normal(9610 + G, (F,), 0.0, 0.1)
    w2, b2 = synth.normal(9620 + G, (G, F), 0.0, 1.0 / np.sqrt(F)), synth.normal(9630 + G, (G,), 0.0, 0.1)
    for a, k in ((E, "cb"), (cw, "conv_w"), (cb, "conv_b"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        assert crc(a) == g[k + "_crc"], k
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    router = (DualGrainFeatureRouter if G == 2 else TripleGrainFeatureRouter)(D, "group-32", "2layer-fc-SiLu").to(dev).eval()
    with torch.no_grad():
        router.gate[0].weight.copy_(t(w1)); router.gate[0].bias.copy_(t(b1))
        router.gate[2].weight.copy_(t(w2)); router.gate[2].bias.copy_(t(b2))
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(t(cw)); conv.bias.copy_(t(cb))
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    hf, hc = t(g["h_fine"]), t(g["h_coarse"])
    hm = t(g["h_median"]) if G == 3 else None
    hb = torch.empty_like(hf)
    with torch.no_grad():
        if G == 2:
            quant, loss, info, grain, gate = encode_dual(router, vq, hf, hc, quant_conv=conv)
            r = vq_assign_routed_dual(hc, hf, t(E), _CodebookPrep(), gate=router(h_fine=hf, h_coarse=hc), conv=conv, h_buf=hb)
        else:
            quant, loss, info, grain, gate = encode_triple(router, vq, hf, hm, hc, quant_conv=conv)
            r = vq_assign_routed_triple(hc, hm, hf, t(E), _CodebookPrep(), router(h_fine=hf, h_median=hm, h_coarse=hc), conv=conv, h_buf=hb)
    assert torch.equal(info[2], r["codes"])
    assert float(g["gate_margin_min"]) > 2e-3
    assert np.abs(gate.cpu().numpy() - g["gate"]).max() < 1e-4                     # [B, G, h, w] logits
    assert np.array_equal(grain.cpu().numpy(), g["grain"].astype(np.int64))
    codes, ref = info[2].cpu().numpy().reshape(-1), g["codes"].astype(np.int64).reshape(-1)
    rate = float((codes == ref).mean())
    print("%s feature model: codes equal to the reference model's encode: %.5f (%d differ)" % (kind, rate, int((codes != ref).sum())))
    assert rate > 0.995
    h = hb.cpu().numpy().reshape(D, -1).T.astype(np.float64)
    bad = np.nonzero(codes != ref)[0]
    if bad.size:
        d_ours = ((h[bad] - E[codes[bad]].astype(np.float64)) ** 2).sum(1)
        d_ref = ((h[bad] - E[ref[bad]].astype(np.float64)) ** 2).sum(1)
        assert np.all(np.abs(d_ref - d_ours) <= 1e-5 * np.maximum(d_ours, 1.0))
    assert abs(float(loss) - float(g["emb_loss"])) <= 1e-4 * abs(float(g["emb_loss"]))


def test_conv_weight_images_are_ordered_across_streams(dev):
    """the conv's weight images are built on the stream of the first call; a second stream that uses them right away (no host
    synchronisation in between, as encode.StreamSlots drives a model) waits for the build's event"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    K, D, B = 1024, 256, 8
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = t(synth.codebook_trained(K, D))
    x = t(synth.z_tokens(synth.codebook_trained(K, D), B, 32, 32, 990))
    conv = _conv(dev, D, 991)
    prep = _CodebookPrep()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        zq1, c1, l1 = vq_assign(x, E, prep, None, conv=conv)
    with torch.cuda.stream(s2):
        zq2, c2, l2 = vq_assign(x, E, prep, None, conv=conv)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2) and torch.equal(zq1, zq2) and torch.equal(l1, l2)


@pytest.mark.parametrize("routed", [False, True])
def test_fused_conv_resolver_overflow_hands_h_to_the_exact_list(dev, oracle_mod, routed):
    """with the conv fused in, the tokens the RESOLVER sends to the exact list (more than RES_CAND = 512
    candidate pairs per 32 queued tokens: a codebook with many duplicate codes) must be evaluated on their conv output.  Since
    round 6 the exact-list kernel computes that output itself from the conv's input (no scratch tensor exists any more; with an
    h_buf it writes the h it scored over pass 1's row of the token): without an h_buf the op has to give the same bits as with
    one, the h_buf has to be the h every token was scored with (oracle GIVEN it), and -- when the conv kernels agree bit for
    bit -- the same as qconv followed by the assign."""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.qconv import quant_conv
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
    D, K, B = 256, 1024, 4
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    E = np.ascontiguousarray(np.tile(synth.codebook_trained(32, D, seed=961), (32, 1)))   # 32 distinct vectors x 32 copies:
    conv = _conv(dev, D, 962)                                  # every token's best is a 32-way exact tie -> queued, 1024
                                                               # candidate pairs per resolver group -> the exact list
    if routed:
        hf, hc = synth.features(963, B, D, 32, 32), synth.features(964, B, D, 16, 16)
        ent = synth.entropy_map(965, B, 16, 16)
        prep = _CodebookPrep()
        hb = torch.empty((B, D, 32, 32), device=dev)
        r_all = vq_assign_routed_dual(t(hc), t(hf), t(E), prep, entropy=t(ent), threshold=THR, conv=conv, h_buf=hb)
        prep2 = _CodebookPrep()
        r = vq_assign_routed_dual(t(hc), t(hf), t(E), prep2, entropy=t(ent), threshold=THR, conv=conv)
        torch.cuda.synchronize()
        queued, listed = prep2.fallback_count()
        assert listed > 0, (queued, listed)
        assert torch.equal(r["codes"], r_all["codes"]) and torch.equal(r["zq"], r_all["zq"])
        assert abs(float(r["loss"][1]) - float(r_all["loss"][1])) <= 1e-6 * abs(float(r_all["loss"][1]))
        o = oracle_mod.vq_assign_nchw(hb.cpu().numpy(), E, r["codebook_mask"].cpu().numpy().reshape(B, -1))
        assert np.array_equal(r["codes"].cpu().numpy().reshape(B, -1), o["codes"])
        assert np.array_equal(r["zq"].cpu().numpy(), o["zq"])
    else:
        x = synth.features(966, B, D, 16, 16)
        h = quant_conv(conv, t(x))
        zq0, codes0, loss0 = vq_assign(h, t(E), _CodebookPrep(), None)
        prep = _CodebookPrep()
        zq, codes, loss = vq_assign(t(x), t(E), prep, None, conv=conv)
        torch.cuda.synchronize()
        queued, listed = prep.fallback_count()
        assert listed > 0, (queued, listed)
        hb = torch.empty((B, D, 16, 16), device=dev)
        zq1, codes1, loss1 = vq_assign(t(x), t(E), _CodebookPrep(), None, conv=conv, h_buf=hb)
        assert torch.equal(codes, codes1) and torch.equal(zq, zq1)
        o = oracle_mod.vq_assign_nchw(hb.cpu().numpy(), E, None)
        assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"]) and np.array_equal(zq.cpu().numpy(), o["zq"])
        if torch.equal(hb, h):                                 # the two conv kernels agree bit for bit on this data: so must the ops
            assert torch.equal(codes, codes0) and torch.equal(zq, zq0)


def test_pixels_to_codes_against_the_reference_models_encode_B4(dev, golden_dir):
    """the WHOLE chain of the entropy-router model from PIXELS -- images regenerated from the stored seed ->
    `Entropy` kernel (dvq_entropy_map_f32) -> DualGrainFixedEntropyRouter -> routing tail + 1x1 quant_conv + VectorQuantize2 as one
    op -- against the golden of the reference's own `DualGrainVQModel.encode` at B = 4 with mixed grains
    (tests/golden/encode_dual_entropy_model_B4.npz, oracle/gen_golden_encode.py): entropy map within 1e-5, grain map and gate
    equal, codes > 99.5 % (near-ties of the two convs' roundings only), loss 1e-4; the fold form gives the same codes as the
    fused-conv form"""
    import json
    import zlib
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual
    from dynamicvectorquantization_amd.entropy import Entropy
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, _CodebookPrep, vq_assign_routed_dual
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    g = np.load(os.path.join(golden_dir, "encode_dual_entropy_model_B4.npz"))
    meta = json.loads(str(g["meta"]))
    crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    K, D, B = 1024, 256, 4
    E = synth.codebook_trained(K, D)
    cw, cb = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0), synth.normal(9502, (D,), 0.0, 0.1)
    assert crc(E) == g["cb_crc"] and crc(cw) == g["conv_w_crc"] and crc(cb) == g["conv_b_crc"]
    img, _ = synth.images_flat_noise(meta["seeds"]["image"], B)
    assert int(crc(img)) == meta["image_crc"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(t(cw)); conv.bias.copy_(t(cb))
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    router = DualGrainFixedEntropyRouter(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    hf, hc = t(g["h_fine"]), t(g["h_coarse"])
    with torch.no_grad():
        ent = Entropy(16, 256, 256)(t(img))                                  # from the pixels, on the fused kernel
        assert np.abs(ent.cpu().numpy() - g["x_entropy"]).max() < 1e-5
        assert np.abs(g["x_entropy"] - router.fine_grain_threshold).min() > 1e-3   # no patch sits at the threshold
        quant, loss, info, grain, gate = encode_dual(router, vq, hf, hc, entropy=ent, quant_conv=conv)
        qf, lf, infof, grainf, gatef = encode_dual(router, vq, hf, hc, entropy=ent, quant_conv=conv, fold=True)
        hb = torch.empty_like(hf)
        vq_assign_routed_dual(hc, hf, t(E), _CodebookPrep(), entropy=ent, threshold=router.fine_grain_threshold, conv=conv, h_buf=hb)
    assert np.array_equal(grain.cpu().numpy(), g["grain"].astype(np.int64)) and 0.2 < float(g["fine_ratio"]) < 0.8
    assert np.array_equal(gate.cpu().numpy(), g["gate"].astype(np.int64))
    assert lf is None and torch.equal(infof[2], info[2]) and torch.equal(grainf, grain) and torch.equal(gatef, gate)
    assert torch.all((qf - quant).abs() <= 1e-6 * torch.clamp(quant.abs(), min=1.0))
    codes, ref = info[2].cpu().numpy().reshape(-1), g["codes"].astype(np.int64).reshape(-1)
    rate = float((codes == ref).mean())
    print("B = 4 from pixels: codes equal to the reference model's encode: %.5f (%d of %d differ)" % (rate, int((codes != ref).sum()), codes.size))
    assert rate > 0.995
    h = np.moveaxis(hb.cpu().numpy().reshape(B, D, -1), 1, 2).reshape(-1, D).astype(np.float64)
    bad = np.nonzero(codes != ref)[0]
    if bad.size:
        d_ours = ((h[bad] - E[codes[bad]].astype(np.float64)) ** 2).sum(1)
        d_ref = ((h[bad] - E[ref[bad]].astype(np.float64)) ** 2).sum(1)
        assert np.all(np.abs(d_ref - d_ours) <= 1e-5 * np.maximum(d_ours, 1.0)), (d_ref - d_ours)
    assert abs(float(loss) - float(g["emb_loss"])) <= 1e-4 * abs(float(g["emb_loss"]))


@pytest.mark.parametrize("kind", ["dual", "triple"])
def test_feature_router_models_B2_against_the_reference_models_own_encode(dev, golden_dir, kind):
    """the feature-router models at B = 2 (tests/golden/encode_{dual,triple}_feature_model_B2.npz): gate logits within 1e-4 of the
    reference's, the grain map equal in every cell whose argmax margin exceeds 1e-3 (the fixtures hold cells down to 3e-4: those may
    flip inside the logits' tolerance and are excluded, with their tokens), codes > 99.5 % on the compared tokens"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual, encode_triple
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    import zlib
    g = np.load(os.path.join(golden_dir, "encode_%s_feature_model_B2.npz" % kind))
    crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    G = 2 if kind == "dual" else 3
    K, D = 1024, 256
    F = G * D
    E = synth.codebook_trained(K, D)
    cw, cb = synth.normal(9501, (D, D, 1, 1), 0.0, 1.0 / 16.0), synth.normal(9502, (D,), 0.0, 0.1)
    w1, b1 = synth.normal(9600 + G, (F, F), 0.0, 1.0 / np.sqrt(F)), synth.normal(9610 + G, (F,), 0.0, 0.1)
    w2, b2 = synth.normal(9620 + G, (G, F), 0.0, 1.0 / np.sqrt(F)), synth.normal(9630 + G, (G,), 0.0, 0.1)
    for a, k in ((E, "cb"), (cw, "conv_w"), (cb, "conv_b"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        assert crc(a) == g[k + "_crc"], k
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    router = (DualGrainFeatureRouter if G == 2 else TripleGrainFeatureRouter)(D, "group-32", "2layer-fc-SiLu").to(dev).eval()
    with torch.no_grad():
        router.gate[0].weight.copy_(t(w1)); router.gate[0].bias.copy_(t(b1))
        router.gate[2].weight.copy_(t(w2)); router.gate[2].bias.copy_(t(b2))
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(t(cw)); conv.bias.copy_(t(cb))
    vq = VectorQuantize2(K, D).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(t(E))
    hf, hc = t(g["h_fine"]), t(g["h_coarse"])
    with torch.no_grad():
        if G == 2:
            quant, loss, info, grain, gate = encode_dual(router, vq, hf, hc, quant_conv=conv)
        else:
            quant, loss, info, grain, gate = encode_triple(router, vq, hf, t(g["h_median"]), hc, quant_conv=conv)
    assert np.abs(gate.cpu().numpy() - g["gate"]).max() < 1e-4
    top2 = np.sort(g["gate"], axis=1)[:, -2:]
    sure = (top2[:, 1] - top2[:, 0]) > 1e-3                                  # [B, hc, wc]
    assert sure.mean() > 0.98
    assert np.array_equal(grain.cpu().numpy()[sure], g["grain"].astype(np.int64)[sure])
    S = 32 // sure.shape[1]
    tok_ok = np.repeat(np.repeat(sure & (grain.cpu().numpy() == g["grain"]), S, 1), S, 2).reshape(-1)
    codes, ref = info[2].cpu().numpy().reshape(-1), g["codes"].astype(np.int64).reshape(-1)
    rate = float((codes[tok_ok] == ref[tok_ok]).mean())
    print("%s feature model B = 2: codes equal to the reference's on %d compared tokens: %.5f" % (kind, int(tok_ok.sum()), rate))
    assert rate > 0.995
