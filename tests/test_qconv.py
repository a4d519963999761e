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


def test_encode_with_quant_conv_end_to_end(dev, oracle_mod, golden_dir):
    """the real encode order select -> quant_conv -> quantize (dqvae_dual_feat.py:59-68) through encode_dual: codes
    / z_q bit-exact GIVEN the kernel's conv output; code match rate vs the fp64 conv-then-quantize order"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.encode import encode_dual, encode_fixed, encode_triple
    from dynamicvectorquantization_amd.qconv import quant_conv_select
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
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
    with torch.no_grad():
        quant, loss, info, grain, gate = encode_dual(router, vq, t(hf), t(hc), entropy=t(ent), quant_conv=conv)
        sel = quant_conv_select(conv, t(hc), t(hf), entropy=t(ent), threshold=router.fine_grain_threshold)
    h = sel["h"].cpu().numpy()
    o = oracle_mod.vq_assign_nchw(h, E, sel["codebook_mask"].cpu().numpy())
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
    # same answer as torch's conv path up to near-ties (the select path with a non-1x1-able conv falls back)
    with torch.no_grad():
        q2, _, info2, _, _ = encode_dual(router, vq, t(hf), t(hc), entropy=t(ent),
                                         quant_conv=torch.nn.Sequential(conv))        # not an nn.Conv2d -> torch path
    assert float((info2[2] == info[2]).float().mean()) > 0.995
    # fixed-granularity model (VQModel.encode): dense conv then VectorQuantizer2-style call
    with torch.no_grad():
        qf, lf, inf_ = encode_fixed(vq, t(hf), quant_conv=conv)
    of = oracle_mod.vq_assign_nchw(__import__("dynamicvectorquantization_amd.qconv", fromlist=["x"]).quant_conv(conv, t(hf)).cpu().numpy(), E, None)
    assert np.array_equal(inf_[2].cpu().numpy().reshape(B, -1), of["codes"])
