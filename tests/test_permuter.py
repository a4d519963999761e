"""Permuter: the oracle against the reference's own known-answer
self-test and synthetic goldens (CPU), the HIP kernels against both (GPU).  Integer work: bit-exact."""
import numpy as np
import pytest
import torch

from tests import _cases as C

FILES = ["permuter_reference_selftest", "permuter_synthetic"]
KEYS = ["coarse_content", "fine_content", "coarse_position", "fine_position", "coarse_segment", "fine_segment"]


@pytest.mark.parametrize("name", FILES)
@pytest.mark.parametrize("order", ["region-first", "row-first"])
def test_oracle_matches_reference(name, order):
    from oracle import permuter as P
    g = C.load(name)
    tag = order.split("-")[0]
    o = P.forward(g["indices"], g["grain"], order=order)
    for k in KEYS:
        assert np.array_equal(o[k], g["%s_%s" % (tag, k)].astype(np.int64)), k
    back = P.forward_back(o["coarse_content"], o["fine_content"], o["coarse_position"], o["fine_position"])
    assert np.array_equal(back, g["indices"].astype(np.int64))          # the reference's round-trip check
    if name == "permuter_reference_selftest":
        assert o["coarse_content"].shape == (2, 104) and o["fine_content"].shape == (2, 613)


def test_oracle_forward_back_duplicates():
    from oracle import permuter as P
    g = C.load("permuter_synthetic")
    back = P.forward_back(g["dup_cc"], g["dup_fc"], g["dup_cp"], g["dup_fp"])
    assert np.array_equal(back, g["dup_back"].astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
@pytest.mark.parametrize("order", ["region-first", "row-first"])
def test_hip_matches_golden(dev, name, order):
    from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
    g = C.load(name)
    tag = order.split("-")[0]
    perm = DualGrainSeperatePermuter(fine_position_order=order)
    idx = torch.from_numpy(g["indices"].astype(np.int64)).to(dev)
    grain = torch.from_numpy(g["grain"].astype(np.int64)).to(dev)
    out = perm(idx, grain)
    for k in KEYS:
        assert out[k].dtype == torch.int64
        assert np.array_equal(out[k].cpu().numpy(), g["%s_%s" % (tag, k)].astype(np.int64)), k
    back = perm.forward_back(out["coarse_content"], out["fine_content"], out["coarse_position"], out["fine_position"])
    assert torch.equal(back, idx)


@pytest.mark.gpu
def test_hip_forward_back_semantics(dev):
    """later entries win, entries after EOS are ignored, no EOS in the coarse stream -> no upsample"""
    from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
    from oracle import permuter as P
    g = C.load("permuter_synthetic")
    perm = DualGrainSeperatePermuter()
    t = lambda a: torch.from_numpy(np.asarray(a, np.int64)).to(dev)
    back = perm.forward_back(t(g["dup_cc"]), t(g["dup_fc"]), t(g["dup_cp"]), t(g["dup_fp"]))
    assert np.array_equal(back.cpu().numpy(), g["dup_back"].astype(np.int64))
    cc = np.array([[5, 6, 7]]); cp = np.array([[3, 4, 10]])            # coarse stream without EOS
    fc = np.array([[11, 1025]]); fp = np.array([[6, 1025]])
    back = perm.forward_back(t(cc), t(fc), t(cp), t(fp)).cpu().numpy()
    assert np.array_equal(back, P.forward_back(cc, fc, cp, fp)) and back.sum() == 11


@pytest.mark.gpu
def test_hip_round_trip_full_batch(dev):
    """B = 256 images of real encoder output shape: encode -> permute -> un-permute is the identity"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
    from oracle import permuter as P
    B = 256
    grain = synth.bernoulli(4201, (B, 16, 16), 0.5).astype(np.int64)
    fine = synth.randint(4202, (B, 32, 32), 1024)
    coarse = synth.randint(4203, (B, 16, 16), 1024).repeat(2, axis=-1).repeat(2, axis=-2)
    idx = np.where(grain.repeat(2, axis=-1).repeat(2, axis=-2) == 1, fine, coarse)
    for order in ("region-first", "row-first"):
        perm = DualGrainSeperatePermuter(fine_position_order=order)
        out = perm(torch.from_numpy(idx).to(dev), torch.from_numpy(grain).to(dev))
        back = perm.forward_back(out["coarse_content"], out["fine_content"], out["coarse_position"], out["fine_position"])
        assert np.array_equal(back.cpu().numpy(), idx)
        o = P.forward(idx[:8], grain[:8], order=order)                 # oracle on a slice (its own padding)
        for k in ("coarse_content", "fine_position"):
            L = o[k].shape[1]
            got = out[k][:8].cpu().numpy()
            assert np.array_equal(got[:, :L][o[k] != (1024 if "content" in k or "fine" in k else 256)],
                                  o[k][o[k] != (1024 if "content" in k or "fine" in k else 256)])
