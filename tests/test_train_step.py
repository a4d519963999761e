"""Training-mode EMA codebook update: HIP scatter-add statistics vs the
reference's dense one-hot matmul.  FP reductions in a different order: 1e-5 relative, not bit-exact."""
import numpy as np
import pytest
import torch

from tests import _cases as C

pytestmark = pytest.mark.gpu


def test_train_step_matches_reference(dev):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    g = C.load("vq2_train_step")
    K, D, B, H, W = (int(g[k]) for k in ("K", "D", "B", "H", "W"))
    E = synth.codebook_trained(K, D, seed=7001)
    z = synth.z_tokens(E, B, H, W, 7002)
    mask = np.where(synth.bernoulli(7003, (B, 1, H, W), 0.5), 1.0, 0.25).astype(np.float32)
    assert C.crc(z) == g["z_crc"] and C.crc(E) == g["cb_crc"]
    m = VectorQuantize2(K, D, restart_unused_codes=False).to(dev)
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E))
    m.codebook.embed_ema.copy_(torch.from_numpy(E))
    m.train()
    xq, loss, (_, _, codes) = m(torch.from_numpy(z).to(dev), codebook_mask=torch.from_numpy(mask).to(dev))
    assert np.array_equal(codes.cpu().numpy(), g["codes"].astype(np.int64))      # assignment uses the old codebook
    assert C.loss_close(float(loss), g["loss"])
    for name, got in (("cluster_size_ema", m.codebook.cluster_size_ema), ("embed_ema", m.codebook.embed_ema),
                      ("weight_after", m.codebook.weight)):
        ref = g[name][:K]                                   # row K of weight is the (randomly initialised) padding row
        err = np.abs(got.detach().cpu().numpy()[:K] - ref).max() / max(1e-30, np.abs(ref).max())
        assert err < 1e-5, (name, err)


def test_ema_accumulate_vs_float64(dev):
    """cfg-2 sized batch: counts exact, sums within 1e-5 of a float64 scatter-add"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import VectorQuantize2
    K, D, B = 1024, 256, 16
    E = synth.codebook_trained(K, D)
    z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 7102)).to(dev)
    m = VectorQuantize2(K, D).to(dev).eval()
    m.codebook.weight.data[:-1].copy_(torch.from_numpy(E).to(dev))
    with torch.no_grad():
        _, _, (_, _, codes) = m(z)
        cs, vs, _ = m.codebook._cluster_sums(z, codes.reshape(-1), nchw=z)
    tok = z.reshape(B, D, -1).permute(0, 2, 1).reshape(-1, D).double()
    ref_cs = torch.bincount(codes.reshape(-1), minlength=K).double()
    ref_vs = torch.zeros(K, D, dtype=torch.float64, device=dev).index_add_(0, codes.reshape(-1), tok)
    assert torch.equal(cs.double(), ref_cs)
    assert float((vs.double() - ref_vs).abs().max() / ref_vs.abs().max()) < 1e-5
    # restart_unused_codes=True path runs (RNG-dependent: shape/finite checks only)
    m2 = VectorQuantize2(K, D).to(dev).train()
    m2(z)
    assert torch.isfinite(m2.codebook.weight).all() and float(m2.codebook.cluster_size_ema.sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("restart", [0, 1, 2])
def test_ema_update_kernel_against_the_reference_expressions(dev, restart):
    """dvq_ema_update_f32 (round 6: the EMA updates, the dead-code restart and _update_embedding as one kernel) against the
    reference's own expressions (quantize2_mask.py:89-115) as torch ops on the CPU: restart rows given (the data-parallel form:
    rank 0's rows), gathered from the NCHW latents by token index, or no restart; counts chosen so that a third of the codes is dead"""
    from dynamicvectorquantization_amd import _lib
    K, D, B, HW = 96, 256, 3, 64
    g = torch.Generator().manual_seed(11 + restart)
    decay, eps = 0.99, 1e-5
    cs = torch.rand(K, generator=g) * 3.0                     # EMA counts: a third below 1 after the update
    emb = torch.randn(K, D, generator=g)
    cnt = torch.randint(0, 5, (K,), generator=g).float()
    vsum = torch.randn(K, D, generator=g) * cnt[:, None]
    z = torch.randn(B, D, HW, generator=g)
    pick = torch.randperm(B * HW, generator=g)[:K]
    rows = z.permute(0, 2, 1).reshape(-1, D)[pick].contiguous()
    # reference expressions
    cs_r = cs.clone().mul_(decay).add_(cnt, alpha=1 - decay)
    emb_r = emb.clone().mul_(decay).add_(vsum, alpha=1 - decay)
    if restart:
        dead = cs_r < 1
        emb_r = torch.where(dead[:, None], rows, emb_r)
        cs_r = cs_r.masked_fill(dead, 1.0)
        assert 5 < int(dead.sum()) < K - 5
    n = cs_r.sum()
    w_r = emb_r / (n * (cs_r + eps) / (n + K * eps)).reshape(-1, 1)
    t = lambda a: a.to(dev).contiguous()
    cs_d, emb_d, w_d, out = t(cs), t(emb), torch.full((K + 1, D), 7.0, device=dev), torch.empty(K, device=dev)
    vs_d, cn_d, z_d, pk_d, rw_d = t(vsum), t(cnt), t(z), t(pick), t(rows)
    with _lib.on_device(dev):
        _lib.check(_lib.lib.dvq_ema_update_f32(vs_d.data_ptr(), cn_d.data_ptr(), decay, eps, K, D, cs_d.data_ptr(), out.data_ptr(),
                                               emb_d.data_ptr(), w_d.data_ptr(), restart, rw_d.data_ptr() if restart == 1 else None,
                                               z_d.data_ptr() if restart == 2 else None, B, HW, pk_d.data_ptr() if restart == 2 else None,
                                               _lib.stream_ptr(dev)), "dvq_ema_update_f32")
    torch.cuda.synchronize()
    assert torch.equal(cs_d.cpu(), cs)                          # the old counts are read, not written
    assert torch.allclose(out.cpu(), cs_r, rtol=1e-6, atol=1e-7) and torch.allclose(emb_d.cpu(), emb_r, rtol=1e-5, atol=1e-6)
    assert torch.allclose(w_d[:K].cpu(), w_r, rtol=1e-5, atol=1e-6) and bool((w_d[K] == 7.0).all())     # the padding row is not touched
    # argument checks: aliasing the count arrays is refused (every workgroup sums the OLD counts)
    assert _lib.lib.dvq_ema_update_f32(vs_d.data_ptr(), cn_d.data_ptr(), decay, eps, K, D, cs_d.data_ptr(), cs_d.data_ptr(),
                                       emb_d.data_ptr(), w_d.data_ptr(), 0, None, None, 0, 0, None, _lib.stream_ptr(dev)) == -1
