"""Autograd and training-mode paths of the drop-in modules against vectors captured from the imported
reference (oracle/gen_golden_grad.py): train-mode VectorQuantize2 forward + EMA update + dead-code restart
+ backward, VectorQuantizer2 codebook gradients (legacy True / False), the differentiable select,
get_soft_codes, remap.  Gradients are fp32 elementwise formulas of bit-exact forward quantities:
1e-6 relative; EMA sums 1e-5 (float atomics)."""
import os
import tempfile

import numpy as np
import pytest
import torch

from tests import _cases as C


def _close(got, ref, rel):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() <= rel * max(1e-30, np.abs(ref).max())


@pytest.mark.gpu
def test_vq2_train_mode_forward_ema_restart_backward(dev, monkeypatch):
    """the EMA update writes the codebook in place between forward and backward; the
    gradient must still be the forward-time one and nothing may raise"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    g = C.load("vq2_train_grad")
    K, D, B, H, W = (int(g[k]) for k in ("K", "D", "B", "H", "W"))
    E = synth.codebook_trained(K, D, seed=7101)
    z = synth.z_tokens(E, B, H, W, 7102)
    mask = np.where(synth.bernoulli(7103, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
    gw = synth.normal(7104, z.shape)
    assert C.crc(z) == g["z_crc"] and C.crc(E) == g["cb_crc"] and C.crc(mask) == g["mask_crc"] and C.crc(gw) == g["gw_crc"]
    m = VectorQuantize2(K, D, restart_unused_codes=True).to(dev)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.codebook.embed_ema.copy_(torch.from_numpy(E))
    m.train()
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.arange(n - 1, -1, -1, device=device))
    zt = torch.from_numpy(z).to(dev).requires_grad_(True)
    xq, loss, (_, _, codes) = m(zt, codebook_mask=torch.from_numpy(mask).to(dev))
    ((xq * torch.from_numpy(gw).to(dev)).sum() + 3.0 * loss).backward()
    assert np.array_equal(codes.cpu().numpy(), g["codes"].astype(np.int64))
    assert C.loss_close(float(loss), g["loss"])
    assert _close(zt.grad.cpu().numpy(), g["z_grad"], 1e-6)
    for name, got in (("cluster_size_ema", m.codebook.cluster_size_ema), ("embed_ema", m.codebook.embed_ema),
                      ("weight_after", m.codebook.weight[:K])):
        assert _close(got.detach().cpu().numpy(), g[name], 1e-5), name
    # the next forward sees the updated codebook (prep cache invalidated by the EMA update)
    m.eval()
    with torch.no_grad():
        _, _, (_, _, c2) = m(zt.detach())
    from oracle import oracle
    o = oracle.vq_assign_nchw(z, m.codebook.weight[:K].detach().cpu().numpy(), None)
    assert np.array_equal(c2.cpu().numpy().reshape(B, -1), o["codes"])


@pytest.mark.gpu
@pytest.mark.parametrize("legacy", [False, True])
def test_vqgan_codebook_and_input_gradients(dev, legacy):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantizer2
    g = C.load("vqgan_grad")
    K, D, B, H, W = (int(g[k]) for k in ("K", "D", "B", "H", "W"))
    E = synth.codebook_default_init(K, D, seed=7205)
    z = synth.z_tokens(synth.codebook_trained(K, D), B, H, W, 7202) * np.float32(0.002)
    gw = synth.normal(7204, z.shape)
    assert C.crc(z) == g["z_crc"] and C.crc(E) == g["cb_crc"] and C.crc(gw) == g["gw_crc"]
    m = VectorQuantizer2(K, D, beta=0.25, legacy=legacy).to(dev)
    m.embedding.weight.data.copy_(torch.from_numpy(E))
    zt = torch.from_numpy(z).to(dev).requires_grad_(True)
    zq, loss, (_, _, idx) = m(zt)
    ((zq * torch.from_numpy(gw).to(dev)).sum() + 5.0 * loss).backward()
    s = "_legacy%d" % int(legacy)
    assert np.array_equal(idx.cpu().numpy(), g["codes" + s].astype(np.int64))
    assert C.loss_close(float(loss), g["loss" + s])
    assert _close(zt.grad.cpu().numpy(), g["z_grad" + s], 1e-6)
    assert _close(m.embedding.weight.grad.cpu().numpy(), g["w_grad" + s], 1e-5)      # index_add_ order
    # an optimizer step through .data in train mode is picked up at the next forward
    m.train()
    with torch.no_grad():
        m.embedding.weight.data.mul_(0.5)
    _, _, (_, _, idx2) = m(zt.detach())
    from oracle import oracle
    o = oracle.vq_assign_nchw(z, E * np.float32(0.5), None)
    assert np.array_equal(idx2.cpu().numpy(), o["codes"].reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dual", "dual_entropy", "triple"])
def test_route_select_is_differentiable(dev, kind):
    """the gradient reaches the encoder branches exactly as through the reference's
    repeat_interleave + torch.where"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.router import route_select_dual, route_select_dual_entropy, route_select_triple
    B, Cc, hc, wc = 2, 8, 4, 6
    t = lambda a: torch.from_numpy(a).to(dev)
    s = 4 if kind == "triple" else 2
    hf = t(synth.features(41, B, Cc, s * hc, s * wc)).requires_grad_(True)
    hco = t(synth.features(42, B, Cc, hc, wc)).requires_grad_(True)
    hm = t(synth.features(43, B, Cc, 2 * hc, 2 * wc)).requires_grad_(True) if kind == "triple" else None
    gup = t(synth.normal(44, (B, Cc, s * hc, s * wc)))
    if kind == "dual":
        out = route_select_dual(t(synth.grain_gate_dual(45, B, hc, wc)), hco, hf)
    elif kind == "dual_entropy":
        out = route_select_dual_entropy(t(synth.entropy_map(46, B, hc, wc)), 1.6777750253677368, hco, hf)
    else:
        out = route_select_triple(t(synth.grain_logits_triple(47, B, hc, wc)), hco, hm, hf)
    h = out["h_triple" if kind == "triple" else "h_dual"]
    assert h.requires_grad and not out["indices"].requires_grad
    (h * gup).sum().backward()
    got = [x.grad.clone() for x in (hco, hm, hf) if x is not None]
    for x in (hco, hm, hf):
        if x is not None:
            x.grad = None
    ind = out["indices"]
    up = ind.repeat_interleave(s, 1).repeat_interleave(s, 2).unsqueeze(1)
    if kind == "triple":
        ref = torch.where(up == 0, hco.repeat_interleave(4, -1).repeat_interleave(4, -2), hf)
        ref = torch.where(up == 1, hm.repeat_interleave(2, -1).repeat_interleave(2, -2), ref)
    else:
        ref = torch.where(up == 0, hco.repeat_interleave(2, -1).repeat_interleave(2, -2), hf)
    assert torch.equal(ref.detach(), h.detach())
    (ref * gup).sum().backward()
    want = [x.grad for x in (hco, hm, hf) if x is not None]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    # no_grad / detached inputs keep the plain kernel path
    with torch.no_grad():
        o2 = route_select_dual(t(synth.grain_gate_dual(45, B, hc, wc)), t(synth.features(42, B, Cc, hc, wc)),
                               t(synth.features(41, B, Cc, 2 * hc, 2 * wc)))
    assert not o2["h_dual"].requires_grad


@pytest.mark.gpu
def test_get_soft_codes_golden(dev):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    g = C.load("vq2_soft_codes")
    K, D = int(g["K"]), int(g["D"])
    E = synth.codebook_trained(K, D, seed=7301)
    assert C.crc(E) == g["cb_crc"]
    m = VectorQuantize2(K, D).to(dev).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    x = torch.from_numpy(g["x"]).to(dev)
    soft, code = m.get_soft_codes(x, temp=float(g["temp"]), stochastic=False)
    assert tuple(soft.shape) == tuple(g["soft"].shape) and code.dtype == torch.int64
    assert np.array_equal(code.cpu().numpy(), g["code"].astype(np.int64))
    assert _close(m.codebook.compute_distances(x).cpu().numpy(), g["dist"], 1e-5)
    assert np.abs(soft.cpu().numpy() - g["soft"]).max() < 1e-4          # exp of distances ~ 1e2
    s2, c2 = m.get_soft_codes(x, temp=0.7, stochastic=True)
    assert tuple(c2.shape) == tuple(code.shape) and int(c2.min()) >= 0 and int(c2.max()) < K
    # the hard code of get_soft_codes equals the kernel's assignment
    assert torch.equal(m.codebook.find_nearest_embedding(x), code)


def test_vqgan_remap_golden():
    """pure index arithmetic: runs on CPU tensors (no kernel involved)"""
    from dynamicvectorquantization_amd.quantize import VectorQuantizer2
    g = C.load("vqgan_remap")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "used.npy")
        np.save(path, g["used"])
        for tag, unk in (("extra", "extra"), ("int", 2)):
            m = VectorQuantizer2(int(g["K"]), int(g["D"]), beta=0.25, remap=path, unknown_index=unk)
            assert m.re_embed == int(g["re_embed_" + tag])
            new = m.remap_to_used(torch.from_numpy(g["inds"].copy()))
            assert np.array_equal(new.numpy(), g["to_used_" + tag])
            assert np.array_equal(m.unmap_to_all(new).numpy(), g["to_all_" + tag])
        m = VectorQuantizer2(int(g["K"]), int(g["D"]), beta=0.25, remap=path, unknown_index="random")
        new = m.remap_to_used(torch.from_numpy(g["inds"].copy()))
        known = np.isin(g["inds"], g["used"])
        assert np.array_equal(new.numpy()[known], g["to_used_int"][known])
        assert int(new.min()) >= 0 and int(new.max()) < m.re_embed


@pytest.mark.gpu
def test_prep_cache_invalidation_and_streams(dev, oracle_mod):
    """ADVICE r1 #4 / VERDICT hygiene: `.data` writes need invalidate_codebook_cache(); load_state_dict and
    .to() invalidate by themselves; two streams driving one quantizer use separate workspaces"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    K, D, B = 256, 256, 4
    E1, E2 = synth.codebook_trained(K, D, seed=11), synth.codebook_trained(K, D, seed=12)
    z = synth.z_tokens(E1, B, 16, 16, 13)
    zt = torch.from_numpy(z).to(dev)
    m = VectorQuantize2(K, D).to(dev).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E1))
    o1, o2 = oracle_mod.vq_assign_nchw(z, E1, None), oracle_mod.vq_assign_nchw(z, E2, None)
    run = lambda: m(zt)[2][2].cpu().numpy().reshape(B, -1)
    with torch.no_grad():
        assert np.array_equal(run(), o1["codes"])
        m.codebook.weight.data[:-1].copy_(torch.from_numpy(E2))          # version counter unchanged
        m.invalidate_codebook_cache()
        assert np.array_equal(run(), o2["codes"])
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["codebook.weight"][:-1] = torch.from_numpy(E1).to(dev)
        m.load_state_dict(sd)
        assert np.array_equal(run(), o1["codes"])
        # two streams, same module, interleaved launches
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        z2 = torch.from_numpy(synth.z_tokens(E1, B, 16, 16, 14)).to(dev)
        torch.cuda.synchronize()
        res = {}
        for rep in range(4):
            for st, inp, key in ((s1, zt, "a"), (s2, z2, "b")):
                with torch.cuda.stream(st):
                    res[key] = m(inp)
        torch.cuda.synchronize()
        assert np.array_equal(res["a"][2][2].cpu().numpy().reshape(B, -1), o1["codes"])
        ob = oracle_mod.vq_assign_nchw(z2.cpu().numpy(), E1, None)
        assert np.array_equal(res["b"][2][2].cpu().numpy().reshape(B, -1), ob["codes"])
        assert np.array_equal(res["b"][0].cpu().numpy(), ob["zq"])
        assert C.loss_close(float(res["a"][1]), oracle_mod.vq_loss(o1["sqerr"], o1["numel"], 0.25))


def _list_items(E, seed):
    from dynamicvectorquantization_amd import synth
    out = []
    for i, shp in enumerate([(5, 7), (33,), (2, 3, 4)]):
        n = int(np.prod(shp))
        z = synth.z_tokens(E, 1, n, 1, seed + i)
        out.append(np.ascontiguousarray(z[0, :, :, 0].T).reshape(shp + (E.shape[1],)))
    return out


@pytest.mark.gpu
def test_list_quantizer_eval_golden(dev):
    """VectorQuantize2List (quantize2_list.py:135-170) in eval mode: one assign over the concatenated rows; codes and
    x_q of every item bit-exact vs the imported reference, loss 1e-5"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2List
    g = C.load("vq2_list_eval")
    K, D = int(g["K"]), int(g["D"])
    E = synth.codebook_trained(K, D, seed=7301)
    xs = _list_items(E, 7310)
    assert C.crc(E) == g["cb_crc"] and [C.crc(x) for x in xs] == list(g["x_crc"])
    m = VectorQuantize2List(K, D).to(dev).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    assert sorted(m.state_dict().keys()) == ["codebook.cluster_size_ema", "codebook.embed_ema", "codebook.weight"]
    with torch.no_grad():
        xq, loss, (_, _, codes) = m([torch.from_numpy(x).to(dev) for x in xs])
    for i, x in enumerate(xs):
        assert codes[i].shape == x.shape[:-1] and xq[i].shape == x.shape
        assert np.array_equal(codes[i].cpu().numpy(), g["codes%d" % i].astype(np.int64)), i
        assert C.crc(xq[i].cpu().numpy()) == g["xq_crc%d" % i], i
    assert C.loss_close(float(loss), g["loss"])


BIG_LIST_SHAPES = [(16, 32, 32), (3001,), (40, 50, 13)]


def _big_list_items(E, seed):
    from dynamicvectorquantization_amd import synth
    out = []
    for i, shp in enumerate(BIG_LIST_SHAPES):
        n = int(np.prod(shp))
        z = synth.z_tokens(E, 1, n, 1, seed + i)
        out.append(np.ascontiguousarray(z[0, :, :, 0].T).reshape(shp + (E.shape[1],)))
    return out


def test_oracle_matches_the_list_quantizer_at_dispatch_size(oracle_mod):
    """CPU: the oracle (row-major tokens are [N, D, 1] to it) against the reference's list quantizer at K = 1024 on 45 385
    tokens (tests/golden/vq2_list_eval_big.npz, made by oracle/gen_golden_list.py from the imported reference,
    quantize2_list.py:153-170): codes of every item, x_q CRCs, the loss (mean over items of the per-item means)"""
    from dynamicvectorquantization_amd import synth
    g = C.load("vq2_list_eval_big")
    K, D = int(g["K"]), int(g["D"])
    E = synth.codebook_trained(K, D)
    xs = _big_list_items(E, 7510)
    assert C.crc(E) == g["cb_crc"] and [C.crc(x) for x in xs] == list(g["x_crc"])
    losses = []
    for i, x in enumerate(xs):
        rows = x.reshape(-1, D)
        o = oracle_mod.vq_assign_nchw(np.ascontiguousarray(rows[:, :, None, None]), E, None)     # [N, D, 1, 1]
        codes = o["codes"].reshape(-1)
        assert C.crc(codes.astype(np.int64)) == g["codes_crc%d" % i], i
        assert np.array_equal(codes[:512], g["codes_head%d" % i].astype(np.int64)), i
        assert C.crc(o["zq"].reshape(x.shape)) == g["xq_crc%d" % i], i
        losses.append(oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))
    assert C.loss_close(float(np.mean(losses)), g["loss"])


@pytest.mark.gpu
def test_list_quantizer_at_dispatch_size_row_major_form(dev):
    """the list quantizer at dispatch size goes through the ROW-MAJOR form of pass 1
    (`dvq_vq_assign_flat_f32`: 16-byte accesses along a token's row): codes of all 45 385 tokens, x_q of every item bit-exact vs
    the imported reference (tests/golden/vq2_list_eval_big.npz), loss 1e-5; the same rows through `VectorQuantize2(channel_last=
    True)` and `VQEmbedding.forward`, and with a row pointer that is NOT 16-byte aligned (falls back to 4-byte accesses)"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, VectorQuantize2List, _CodebookPrep, vq_assign
    g = C.load("vq2_list_eval_big")
    K, D = int(g["K"]), int(g["D"])
    E = synth.codebook_trained(K, D)
    xs = _big_list_items(E, 7510)
    m = VectorQuantize2List(K, D).to(dev).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    with torch.no_grad():
        xq, loss, (_, _, codes) = m([torch.from_numpy(x).to(dev) for x in xs])
    for i, x in enumerate(xs):
        assert codes[i].shape == x.shape[:-1] and xq[i].shape == x.shape
        assert C.crc(codes[i].cpu().numpy().astype(np.int64)) == g["codes_crc%d" % i], i
        assert C.crc(xq[i].cpu().numpy()) == g["xq_crc%d" % i], i
    assert C.loss_close(float(loss), g["loss"])
    # channel_last module and the embedding's own forward on item 0
    x0 = torch.from_numpy(xs[0]).to(dev)                      # [16, 32, 32, D]
    vq = VectorQuantize2(K, D, accept_image_fmap=False, channel_last=True).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    with torch.no_grad():
        q0, _, (_, _, c0) = vq(x0.reshape(16, 1024, D))
        emb, ids = vq.codebook(x0)
    assert torch.equal(c0.reshape(-1), codes[0].reshape(-1)) and torch.equal(q0.reshape(x0.shape), xq[0])
    assert torch.equal(ids.reshape(-1), codes[0].reshape(-1))
    assert np.array_equal(emb.cpu().numpy(), E[ids.cpu().numpy()])
    # misaligned rows: a [N, D] view 4 bytes into a buffer
    buf = torch.empty(3001 * D + 1, dtype=torch.float32, device=dev)
    xv = buf[1:].view(3001, D)
    xv.copy_(torch.from_numpy(xs[1]))
    assert xv.data_ptr() % 16 == 4
    zq, cc, _ = vq_assign(xv, torch.from_numpy(E).to(dev), _CodebookPrep())
    assert torch.equal(cc, codes[1]) and torch.equal(zq, xq[1])


@pytest.mark.gpu
def test_list_quantizer_train_golden(dev, monkeypatch):
    """train mode: the EMA update after item i is what item i + 1 is quantized with (the reference's loop), gradients
    of every item, buffers and codebook after the step vs the imported reference"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2List
    g = C.load("vq2_list_train")
    K, D = int(g["K"]), int(g["D"])
    E = synth.codebook_trained(K, D, seed=7401)
    xs = _list_items(E, 7410)
    gws = [synth.normal(7420 + i, x.shape) for i, x in enumerate(xs)]
    assert C.crc(E) == g["cb_crc"] and [C.crc(x) for x in xs] == list(g["x_crc"])
    m = VectorQuantize2List(K, D, restart_unused_codes=True).to(dev)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.codebook.embed_ema.copy_(torch.from_numpy(E))
    m.train()
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.arange(n - 1, -1, -1, device=device))
    xt = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in xs]
    xq, loss, (_, _, codes) = m(xt)
    (sum((q * torch.from_numpy(gw).to(dev)).sum() for q, gw in zip(xq, gws)) + 2.0 * loss).backward()
    for i in range(len(xs)):
        assert np.array_equal(codes[i].cpu().numpy(), g["codes%d" % i].astype(np.int64)), i
        assert _close(xt[i].grad.cpu().numpy(), g["grad%d" % i], 1e-6), i
    assert C.loss_close(float(loss), g["loss"])
    for name, got in (("cluster_size_ema", m.codebook.cluster_size_ema), ("embed_ema", m.codebook.embed_ema),
                      ("weight_after", m.codebook.weight[:K])):
        assert _close(got.detach().cpu().numpy(), g[name], 1e-5), name



@pytest.mark.gpu
@pytest.mark.parametrize("D,HW", [(256, 32 * 32), (64, 7 * 9), (128, 16)])
def test_fused_backward_kernel_equals_the_torch_expression(dev, D, HW):
    """dvq_vq_backward_nchw_f32 (one streaming pass) against the element-wise torch expression it replaces, bit for bit:
    g_z = g_zq + (g_loss * fl(2 c / numel)) * ((z - e) * m), with and without mask / g_zq, and through autograd with a loss-only
    and a z_q-only objective; the forward-time codebook is used even when the weight is overwritten before backward"""
    from dynamicvectorquantization_amd import _lib, synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    B, K = 3, 256
    E = synth.codebook_trained(K, D, seed=8101)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    z = t(synth.normal(8102, (B, D, HW), 0.0, 1.0))
    codes = torch.from_numpy(np.random.default_rng(8103).integers(0, K, (B, HW))).to(dev)
    mask = t(np.where(synth.bernoulli(8104, (B, HW), 0.5), 1.0, 0.25).astype(np.float32))
    gq = t(synth.normal(8105, (B, D, HW), 0.0, 1.0))
    gl = torch.tensor(3.0, device=dev)
    Et = t(E)
    e = Et[codes].permute(0, 2, 1)
    cs = 0.25 * (2.0 / z.numel())
    for m_, gq_ in ((mask, gq), (None, gq), (mask, None)):
        gz = torch.empty_like(z)
        _lib.check(_lib.lib.dvq_vq_backward_nchw_f32(z.data_ptr(), Et.data_ptr(), codes.data_ptr(), _lib.ptr(m_), _lib.ptr(gq_),
                                                     gl.reshape(1).data_ptr(), float(cs), B, D, HW, K, gz.data_ptr(),
                                                     _lib.stream_ptr(dev)), "bw")
        diff = z - e
        if m_ is not None:
            diff = diff * m_.reshape(B, 1, HW)
        want = (gq_ if gq_ is not None else 0) + gl * cs * diff
        assert torch.equal(gz, want)
    # through the module: the loss alone, z_q alone, and the weight overwritten between forward and backward
    vq = VectorQuantize2(K, D, accept_image_fmap=True).to(dev).eval()
    vq.codebook.weight.data[:-1].copy_(Et)
    x = z.reshape(B, D, HW, 1).clone().requires_grad_(True)
    q, loss, (_, _, c) = vq(x)
    w0 = vq.codebook.weight.data[:-1].clone()
    vq.codebook.weight.data[:-1].mul_(-3.0)                   # what an in-place EMA update would do before backward
    loss.backward()
    e2 = w0[c.reshape(B, HW)].permute(0, 2, 1).reshape(x.shape)
    want = (torch.tensor(1.0, device=dev) * (0.25 * (2.0 / x.numel()))) * (x.detach() - e2)
    assert torch.equal(x.grad, want)
    x.grad = None
    vq.codebook.weight.data[:-1].copy_(w0)
    vq.invalidate_codebook_cache()
    q, loss, _ = vq(x)
    q.backward(gq.reshape(x.shape))
    assert torch.equal(x.grad, gq.reshape(x.shape))


@pytest.mark.gpu
@pytest.mark.parametrize("D,HW,K", [(256, 32 * 32, 1024), (64, 7 * 9, 100), (96, 50, 8192)])
def test_codebook_gradient_kernel_vs_index_add(dev, D, HW, K):
    """dvq_vq_backward_codebook_nchw_f32 (tiles transposed through LDS, equal codes of a tile combined, one row of float atomics per
    distinct code and tile) against the reference's construction -- index_add_ of the [N, D] masked differences -- in float64:
    equal up to fp32 summation order; accumulates into g_weight; skewed code usage and runs of equal codes included"""
    from dynamicvectorquantization_amd import _lib, synth
    B = 4
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(8200 + D)
    E = t(synth.normal(8201, (K, D), 0.0, 0.5))
    z = t(synth.normal(8202, (B, D, HW), 0.0, 1.0))
    codes_np = (rng.random((B, HW)) ** 3 * K).astype(np.int64).clip(0, K - 1)        # skewed: many tokens on the low codes
    codes_np[0, : HW // 2] = np.repeat(codes_np[0, : HW // 2: 4], 4)[: HW // 2]      # runs of four equal codes
    codes = t(codes_np)
    mask = t(np.where(synth.bernoulli(8203, (B, HW), 0.5), 1.0, 0.25).astype(np.float32))
    gl = torch.tensor(1.7, device=dev)
    cs = 1.0 * (2.0 / z.numel())
    for m_ in (mask, None):
        gw = torch.zeros((K + 1, D), device=dev)
        gw[K] = 0.5
        run = lambda: _lib.check(_lib.lib.dvq_vq_backward_codebook_nchw_f32(
            z.data_ptr(), E.data_ptr(), codes.data_ptr(), _lib.ptr(m_), gl.reshape(1).data_ptr(), float(cs), B, D, HW, K,
            gw.data_ptr(), _lib.stream_ptr(dev)), "gw")
        run()
        diff = (z.double() - E.double()[codes].permute(0, 2, 1))
        if m_ is not None:
            diff = diff * m_.double().reshape(B, 1, HW)
        ge = (-(gl.double() * cs) * diff).permute(0, 2, 1).reshape(-1, D)
        want = torch.zeros((K + 1, D), dtype=torch.float64, device=dev).index_add_(0, codes.reshape(-1), ge)
        err = (gw[:K].double() - want[:K]).abs().max() / want.abs().max()
        assert float(err) < 1e-5, float(err)
        assert torch.equal(gw[K], torch.full((D,), 0.5, device=dev))                 # the padding row is never touched
        run()                                                                         # it ACCUMULATES
        assert float((gw[:K].double() - 2 * want[:K]).abs().max() / want.abs().max()) < 2e-5


@pytest.mark.gpu
def test_restart_pick_is_a_prefix_of_a_random_permutation(dev):
    """the dead-code restart takes `torch.randperm(n)[:K]` in the reference (quantize2_mask.py:93-96); `_restart_pick` samples the
    same distribution without the device-wide sort: K distinct indices of range(n), every index equally likely, order random;
    and a replaced torch.randperm (how tests / goldens pin the restart) is honoured"""
    from dynamicvectorquantization_amd import quantize
    n, k = 262144, 1024
    hits = torch.zeros(n, dtype=torch.int64, device=dev)
    first = []
    for _ in range(200):
        p = quantize._restart_pick(n, k, dev)
        assert p.shape == (k,) and p.dtype == torch.int64
        assert int(p.min()) >= 0 and int(p.max()) < n and torch.unique(p).numel() == k
        hits[p] += 1
        first.append(int(p[0]))
    # 204 800 draws over 262 144 cells: the lower / upper half of the range get half the draws each (5 sigma = 0.6 %)
    lo = int(hits[: n // 2].sum())
    assert abs(lo - 102400) < 1200, lo
    assert len(set(first)) > 190                                  # the leading entry moves
    import torch as _t
    real = _t.randperm
    try:
        _t.randperm = lambda m, device=None, **kw: _t.arange(m - 1, -1, -1, device=device)
        assert quantize._restart_pick(n, 4, dev).tolist() == [n - 1, n - 2, n - 3, n - 4]
    finally:
        _t.randperm = real
