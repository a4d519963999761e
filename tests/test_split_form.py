"""GPU (-m gpu): the small-batch form of pass 1 (several workgroups per token block, each on a slice of the code tiles,
fence-free hand-off to the workgroup that merges) -- every op that takes it for <= 8192 positions must give the bits of DVQ_MODE_EXACT
(itself pinned to the oracle / the reference goldens elsewhere) and of the oracle directly, call after call on one workspace with
fresh data (a stale slice entry would show as a wrong code), at every slice count (K decides how many slices a block gets)."""
import numpy as np
import pytest
import torch

from tests import _cases as C

pytestmark = pytest.mark.gpu


def _eq(a, b):
    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("K", [1024, 333, 96, 40])          # 32 / 11 / 3 / 2 code tiles: 8, 4 (or fewer), 1 slices
def test_dense_nchw_and_row_major_equal_exact_mode_call_after_call(dev, D, K):
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    E = synth.codebook_trained(K, D, seed=900 + K)
    Et = torch.from_numpy(E).to(dev)
    pf, pr, px = _CodebookPrep(), _CodebookPrep(), _CodebookPrep()
    for it, (B, H, W) in enumerate([(4, 16, 16), (1, 16, 16), (3, 9, 7), (8, 32, 32), (2, 5, 3), (4, 16, 16)]):
        z = torch.from_numpy(synth.z_tokens(E, B, H, W, 7000 + 13 * it + K)).to(dev)
        m = torch.from_numpy(np.where(synth.uniform(7100 + it, (B, 1, H, W)) < 0.5, 1.0, 0.25).astype(np.float32)).to(dev)
        zq0, c0, l0 = vq_assign(z, Et, px, m, mode=_lib.MODE_EXACT)
        for rep in range(3):                                  # same workspace, three times
            zq1, c1, l1 = vq_assign(z, Et, pf, m)
            assert torch.equal(c0, c1) and _eq(zq0, zq1)
            assert abs(float(l0[1]) - float(l1[1])) <= 1e-6 * abs(float(l0[1])) + 1e-30
        zr = z.reshape(B, D, -1).permute(0, 2, 1).reshape(-1, D).contiguous()
        zq2, c2, l2 = vq_assign(zr, Et, pr, m.reshape(-1))
        assert torch.equal(c0.reshape(-1), c2) and _eq(zq0, zq2.reshape(B, -1, D).permute(0, 2, 1).reshape(zq0.shape))
        assert abs(float(l0[1]) - float(l2[1])) <= 1e-6 * abs(float(l0[1])) + 1e-30


def test_dense_against_the_oracle(dev, oracle_mod):
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    K, D, B, H = 1024, 256, 4, 16                             # BASELINE configs[0]
    E = synth.codebook_trained(K, D)
    z = synth.z_tokens(E, B, H, H, 2001)
    o = oracle_mod.vq_assign_nchw(z, E, None)
    zq, codes, loss = vq_assign(torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev), _CodebookPrep())
    assert np.array_equal(codes.cpu().numpy().reshape(B, -1), o["codes"])
    assert np.array_equal(zq.cpu().numpy(), o["zq"])
    assert C.loss_close(float(loss[1]), oracle_mod.vq_loss(o["sqerr"], o["numel"], 0.25))


@pytest.mark.parametrize("G", [2, 3])
@pytest.mark.parametrize("with_conv", [False, True])
def test_routed_ops_equal_the_unfused_chain(dev, G, with_conv):
    """routed dual / triple at B = 1 .. 4 (<= 8192 positions: per-lane select in the split form), with and without the fused conv"""
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual, vq_assign_routed_triple
    from dynamicvectorquantization_amd.router import route_select_dual, route_select_triple
    from dynamicvectorquantization_amd.qconv import quant_conv
    K, D = 1024, 256
    E = synth.codebook_trained(K, D)
    Et = torch.from_numpy(E).to(dev)
    conv = None
    if with_conv:
        conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(synth.normal(8101, (D, D, 1, 1), 0.0, 1.0 / 16.0)).to(dev))
            conv.bias.copy_(torch.from_numpy(synth.normal(8102, (D,), 0.0, 0.1)).to(dev))
    prep = _CodebookPrep()
    S = 1 << (G - 1)
    for it, (B, hc, wc) in enumerate([(4, 16 // (G - 1), 16 // (G - 1)), (1, 8, 8), (3, 5, 7), (2, 16 // (G - 1), 16 // (G - 1))]):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        hs = [t(synth.z_tokens(E, B, hc << g, wc << g, 8200 + 17 * it + g)) for g in range(G)]
        gate = t(synth.normal(8300 + it, (B, hc, wc, G)))
        with torch.no_grad():
            if G == 2:
                r = vq_assign_routed_dual(hs[0], hs[1], Et, prep, gate=gate, conv=conv)
                sel = route_select_dual(gate, hs[0], hs[1]); hsel = sel["h_dual"]
            else:
                r = vq_assign_routed_triple(hs[0], hs[1], hs[2], Et, prep, gate, conv=conv)
                sel = route_select_triple(gate, hs[0], hs[1], hs[2]); hsel = sel["h_triple"]
            if with_conv:
                hsel = quant_conv(conv, hsel)
            zq0, c0, l0 = vq_assign(hsel, Et, _CodebookPrep(), sel["codebook_mask"], mode=_lib.MODE_EXACT)
        assert torch.equal(r["indices"], sel["indices"]) and torch.equal(r["codebook_mask"], sel["codebook_mask"])
        assert torch.equal(r["codes"].reshape(c0.shape), c0) and _eq(r["zq"], zq0)
        assert abs(float(l0[1]) - float(r["loss"][1])) <= 2e-6 * abs(float(l0[1])) + 1e-30
