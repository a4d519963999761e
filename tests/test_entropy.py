"""Patch-entropy map (row a12): fp32 transcendental math -> 1e-5 tolerance; the grain map the
fixed-entropy router derives from it must equal the reference's (values sit far from the threshold)."""
import numpy as np
import pytest
import torch

from tests import _cases as C


def _check(run):
    from dynamicvectorquantization_amd import synth
    g = C.load("entropy_map_B2")
    img, noisy = synth.images_flat_noise(int(g["seed"]), 2)
    assert C.crc(img) == g["img_crc"] and np.array_equal(noisy, g["noisy"].astype(bool))
    with torch.no_grad():
        ent = run(torch.from_numpy(img)).cpu().numpy()
    assert ent.shape == (2, 16, 16) and ent.dtype == np.float32
    ref = g["entropy"]
    assert np.all(np.abs(ent - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref))), np.abs(ent - ref).max()
    assert np.array_equal(ent > float(g["thr"]), noisy)                # identical grain map
    assert ent[~noisy].min() > 0                                         # 1e-40 epsilon survived (no flush to zero)
    return ent


def test_entropy_torch_restatement_matches_reference():
    """the comparator itself (oracle/entropy_torch.py, CPU) against the captured reference output"""
    from oracle.entropy_torch import entropy_map
    _check(lambda x: entropy_map(x, chunk=1))


@pytest.mark.gpu
def test_entropy_gpu_matches_reference(dev, golden_dir):
    import os
    from dynamicvectorquantization_amd.entropy import Entropy
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    ent = _check(lambda x: Entropy(16, 256, 256)(x.to(dev)))          # the fused HIP kernel
    r = DualGrainFixedEntropyRouter(os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
    gate = r(entropy=torch.from_numpy(ent).to(dev))
    g = C.load("entropy_map_B2")
    assert np.array_equal(gate[..., 1].cpu().numpy().astype(bool), g["noisy"].astype(bool))


def test_fused_entropy_refuses_cpu_tensors():
    from dynamicvectorquantization_amd import _lib
    from dynamicvectorquantization_amd.entropy import Entropy
    with pytest.raises(_lib.DvqError):
        Entropy(16, 256, 256)(torch.zeros(1, 3, 256, 256))
    with pytest.raises(_lib.DvqError):                            # other patch sizes run as GPU tensor ops: still no CPU path
        Entropy(8, 256, 256)(torch.zeros(1, 3, 256, 256))


@pytest.mark.gpu
def test_fused_entropy_full_batch(dev):
    """B = 64: fused kernel vs the reference op sequence as torch ops ON THE CPU (the reference path of record).  Not on
    the GPU: flat patches whose gray value sits ~14.4 sigma below the first bin give a kernel value of one or two fp32
    SUBNORMAL ulps, which the normalisation by (sum + 1e-40) turns into an entropy of ~1e-4; glibc's expf and this
    kernel round such values to the nearest subnormal, torch's device exp flushes them to zero."""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.entropy import Entropy
    from oracle.entropy_torch import entropy_map
    img, noisy = synth.images_flat_noise(5001, 64)
    x = torch.from_numpy(img)
    with torch.no_grad():
        a = Entropy(16, 256, 256)(x.to(dev)).cpu()
        b = entropy_map(x, chunk=8)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5), float((a - b).abs().max())
    assert np.array_equal((a > 1.6777750253677368).numpy(), noisy)


@pytest.mark.gpu
def test_threshold_calibration_matches_reference_procedure(dev):
    """reference scripts/tools/calculate_entropy_thresholds.py:92-110 on synthetic images: np.sort of all patch
    entropies, threshold k = sorted[(size * k) // 100]; a router built from the table at ratio r sends ~r of the
    patches to the fine grain"""
    import json
    import os
    import tempfile
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.entropy import Entropy, calibrate_thresholds
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    imgs = [torch.from_numpy(synth.images_flat_noise(5100 + i, 8, p_noise=0.3 + 0.1 * i)[0]).to(dev) for i in range(3)]
    table = calibrate_thresholds(imgs)
    ent = np.sort(np.concatenate([Entropy(16, 256, 256)(x).reshape(-1).cpu().numpy() for x in imgs]))
    assert list(table.keys()) == [str(k) for k in range(1, 100)]
    for k in (1, 25, 50, 75, 99):
        assert table[str(k)] == float(ent[(ent.size * k) // 100])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "thr.json")
        json.dump(table, open(path, "w"))
        r = DualGrainFixedEntropyRouter(path, 0.25)                 # key "75": the top quarter of the patches is fine
        gate = r(entropy=torch.from_numpy(ent.reshape(1, 1, -1).copy()).to(dev))
        frac = float(gate[..., 1].float().mean())
        assert abs(frac - 0.25) < 0.02, frac


@pytest.mark.gpu
def test_fused_entropy_value_sweep_and_special_pixels(dev):
    """patches built to stress the two-exp factorisation of the fused kernel: flat patches at every offset between two
    bin centres and outside [0, 1] (the reference's [-1, 1] images against bins on [0, 1]), smooth ramps, two-level
    patches, and NaN / Inf pixels -- against the reference's op sequence as torch ops on the CPU"""
    from dynamicvectorquantization_amd.entropy import Entropy
    from oracle.entropy_torch import entropy_map
    rng = np.random.default_rng(7)
    img = np.zeros((4, 3, 256, 256), dtype=np.float32)
    vals = np.concatenate([np.linspace(-0.25, 1.25, 200), rng.uniform(-1, 1, 56)]).astype(np.float32)
    for i in range(256):                                         # image 0: flat patches, value sweep
        py, px = divmod(i, 16)
        img[0, :, py * 16:(py + 1) * 16, px * 16:(px + 1) * 16] = vals[i]
    ramp = np.linspace(0, 1, 256, dtype=np.float32)
    img[1] = ramp[None, None, :] * np.float32(0.9) + ramp[None, :, None] * np.float32(0.1)       # smooth ramps
    two = rng.uniform(0, 1, (16, 16, 2)).astype(np.float32)
    sel = rng.integers(0, 2, (256, 256))
    for py in range(16):
        for px in range(16):
            blk = sel[py * 16:(py + 1) * 16, px * 16:(px + 1) * 16]
            img[2, :, py * 16:(py + 1) * 16, px * 16:(px + 1) * 16] = two[py, px][blk]
    img[3] = rng.uniform(-1, 1, (3, 256, 256)).astype(np.float32)
    img[3, 0, 5, 5] = np.nan                                     # patch (0, 0): NaN -> NaN entropy
    img[3, 1, 40, 200] = np.inf                                  # patch (2, 12): +Inf pixel contributes nothing ...
    img[3, 2, 100, 100] = -np.inf                                # ... but 0.114 * -inf + finite = -inf, same
    x = torch.from_numpy(img)
    with torch.no_grad():
        a = Entropy(16, 256, 256)(x.to(dev)).cpu().numpy()
        b = entropy_map(x, chunk=2).numpy()                        # on the CPU: see test_fused_entropy_full_batch
    nan_b = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan_b) and nan_b[3, 0, 0]
    ok = ~nan_b
    assert np.all(np.abs(a[ok] - b[ok]) <= 1e-5 * np.maximum(1.0, np.abs(b[ok]))), float(np.abs(a[ok] - b[ok]).max())



@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 16, 16), (3, 48, 80), (5, 16, 48), (2, 240, 240), (70, 64, 64)])
def test_fused_entropy_shapes_and_edges_of_the_far_path(dev, shape):
    """round 6 (the kernel became persistent, two patches per wave): a single patch, odd patch counts (the last wave's second half
    has no patch), rows of an odd number of patches (a wave's two patches sit in different rows of patches), more pairs than resident
    waves can take in one go -- and gray values concentrated where the bins at distance 3 and 4 are all a patch has (nearest bin -4,
    -3, 34, 35: the register path of the far bins), mixed with ordinary patches in the same wave"""
    from dynamicvectorquantization_amd.entropy import Entropy
    from oracle.entropy_torch import entropy_map
    B, H, W = shape
    rng = np.random.default_rng(100 + B + H)
    img = rng.uniform(-1, 1, (B, 3, H, W)).astype(np.float32)
    gh, gw = H // 16, W // 16
    k = 0
    for b in range(B):
        for py in range(gh):
            for px in range(gw):
                k += 1
                if k % 3 == 0:       # flat-ish patch just outside the bins: gray in (-0.145, -0.08) or (1.08, 1.145)
                    v = rng.uniform(-0.145, -0.081) if k % 2 else rng.uniform(1.081, 1.145)
                    img[b, :, py * 16:(py + 1) * 16, px * 16:(px + 1) * 16] = np.float32(v) / np.float32(0.9999)
                    img[b, :, py * 16 + 3, px * 16 + 5] = np.float32(-0.9)      # and one pixel far outside
                elif k % 7 == 0:     # far outside altogether: every bin 0 -> the 1e-40 epsilons decide
                    img[b, :, py * 16:(py + 1) * 16, px * 16:(px + 1) * 16] = np.float32(-0.6)
    x = torch.from_numpy(img)
    with torch.no_grad():
        a = Entropy(16, W, H)(x.to(dev)).cpu().numpy()
        b_ = entropy_map(x, chunk=2).numpy()
    assert a.shape == (B, gh, gw)
    assert np.all(np.abs(a - b_) <= 1e-5 * np.maximum(1.0, np.abs(b_))), (float(np.abs(a - b_).max()), np.unravel_index(np.abs(a - b_).argmax(), a.shape))


@pytest.mark.gpu
def test_entropy_other_patch_sizes(dev):
    """patch sizes the fused kernel does not cover (the reference's calculate_entropy_thresholds.py takes --patch_size)
    run as tensor ops on the GPU: same values as the comparator ops, and patch 16 through both routes agrees"""
    from dynamicvectorquantization_amd import synth
    from dynamicvectorquantization_amd.entropy import Entropy, _entropy_ops
    from oracle.entropy_torch import entropy_map
    x, _ = synth.images_flat_noise(5103, 3, size=128, patch=8)
    xt = torch.from_numpy(x).to(dev)
    for p in (8, 32):
        got = Entropy(p, 128, 128)(xt)
        assert tuple(got.shape) == (3, 128 // p, 128 // p)
        assert torch.allclose(got, entropy_map(xt, patch=p), rtol=0, atol=1e-6)
    assert torch.allclose(Entropy(16, 128, 128)(xt), _entropy_ops(xt, 16), rtol=0, atol=2e-5)
