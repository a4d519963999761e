"""The codebook "prep" buffer of dvq_codebook_prepare_f32, section by section, against a numpy restatement of the documented
layout (csrc/dvq_common.h, csrc/vq_assign_filter.hip): f32 tile images in fp32-MFMA operand order, squared norms in the ATen
order of the oracle, the fp16 section's meta words, both fp16 tile images and the accumulator seeds.  Everything here is
bit-exact except etamax, which is a bound (>= the largest rounding-residual norm, within its 0.1 % margin).  Round 6 rebuilt
the prep as two launches (was seven); this pins the format the assign kernels read."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED_PAD = np.float32(-3.0e38)


def _sumsq_aten_order(E):
    """oracle/dvq_oracle.c: dvq_oracle_sumsq -- 32 accumulators over k0, ((a[l]+a[l+8])+a[l+16])+a[l+24], then l = 0..7 in order"""
    K, D = E.shape
    sq = (E * E).astype(np.float32)
    a = np.zeros((K, 32), np.float32)
    for k0 in range(0, D, 32):
        a = (a + sq[:, k0:k0 + 32]).astype(np.float32)
    tl = (((a[:, 0:8] + a[:, 8:16]).astype(np.float32) + a[:, 16:24]).astype(np.float32) + a[:, 24:32]).astype(np.float32)
    s = tl[:, 0].copy()
    for l in range(1, 8):
        s = (s + tl[:, l]).astype(np.float32)
    return s


def _expected(E):
    K, D = E.shape
    T = (K + 31) // 32
    Ep = np.zeros((32 * T, D), np.float32)
    Ep[:K] = E
    en = np.zeros(32 * T, np.float32)
    en[:K] = _sumsq_aten_order(E)
    # f32 tiles: img[kg][c][p] = E[32t + c][8kg + 2(p & 3) + (p >> 2)]
    p = np.arange(8)
    kk = 2 * (p & 3) + (p >> 2)
    tiles = Ep.reshape(T, 32, D // 8, 8)[:, :, :, kk]                # [t][c][kg][p]
    tiles = np.ascontiguousarray(tiles.transpose(0, 2, 1, 3)).reshape(T, 32 * D)
    finite = bool(np.isfinite(E).all() and np.isfinite(en[:K]).all())
    amax = np.float32(np.abs(E[np.isfinite(E)]).max()) if np.isfinite(E).any() else np.float32(0)
    b = 0
    if amax > 0:
        b = 15 - int(np.frexp(amax)[1])
    ok = finite and -100 <= b <= 100
    sb = np.float32(np.ldexp(1.0, b if ok else 0))
    enmax = np.float32(en[:K][np.isfinite(en[:K])].max()) if np.isfinite(en[:K]).any() else np.float32(0)
    emax = np.float32(np.sqrt(enmax, dtype=np.float32) * np.float32(1.00001))
    with np.errstate(over="ignore", invalid="ignore"):
        V = (Ep * sb).astype(np.float32)
        H = V.astype(np.float16)
        R = (V - H.astype(np.float32)).astype(np.float64)
    lane = np.arange(64)
    S16, S32 = D // 16, D // 32
    Hr = H.reshape(T, 32, D)
    img8 = np.empty((T, S16, 64, 8), np.float16)
    for s in range(S16):
        img8[:, s] = np.stack([Hr[:, l & 31, 16 * s + 8 * (l >> 5):16 * s + 8 * (l >> 5) + 8] for l in lane], 1)
    img16 = np.empty((T, 2 * S32, 64, 8), np.float16)
    for c2 in range(2):
        for sp in range(S32):
            img16[:, c2 * S32 + sp] = np.stack([Hr[:, 16 * c2 + (l & 15), 32 * sp + 8 * (l >> 4):32 * sp + 8 * (l >> 4) + 8]
                                                for l in lane], 1)
    with np.errstate(over="ignore", invalid="ignore"):
        seeds = np.maximum((np.float32(-0.5) * sb * en).astype(np.float32), SEED_PAD)
    seeds[K:] = SEED_PAD
    seeds = np.where(np.isnan((np.float32(-0.5) * sb * en)), SEED_PAD, seeds)      # fmaxf(NaN, pad) = pad
    seeds[K:] = SEED_PAD
    eta_true = float(np.sqrt(np.nanmax((R[:K] ** 2).sum(1)))) if ok else None
    return dict(T=T, tiles=tiles, en=en, ok=int(ok), b=b, sb=sb, emax=emax, enmax=enmax, img8=img8, img16=img16, seeds=seeds,
                eta_true=eta_true)


def _build(E, dev):
    from dynamicvectorquantization_amd import _lib
    K, D = E.shape
    Et = torch.from_numpy(E).to(dev)
    nb = _lib.lib.dvq_codebook_prep_bytes(K, D)
    buf = torch.full((nb,), 0xA5, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib.dvq_codebook_prepare_f32(Et.data_ptr(), K, D, buf.data_ptr(), nb, _lib.stream_ptr(dev)), "prepare")
    torch.cuda.synchronize()
    return buf.cpu().numpy()


def _cases():
    from dynamicvectorquantization_amd import synth
    rng = np.random.default_rng(17)
    yield "trained 1024x256", synth.codebook_trained(1024, 256)
    yield "default init 1000x256 (K % 32 != 0)", synth.codebook_default_init(1000, 256)
    yield "33x64", (rng.standard_normal((33, 64)) * 1e-3).astype(np.float32)
    yield "520x128", (rng.standard_normal((520, 128)) * 40.0).astype(np.float32)
    yield "8200x64 (one workgroup per tile)", rng.standard_normal((8200, 64)).astype(np.float32)
    e = rng.standard_normal((100, 256)).astype(np.float32); e[57, 13] = np.nan
    yield "a NaN", e
    e = rng.standard_normal((100, 256)).astype(np.float32); e[99, 255] = np.inf
    yield "an inf", e
    yield "zeros", np.zeros((64, 128), np.float32)


@pytest.mark.parametrize("name,E", list(_cases()), ids=[n for n, _ in _cases()])
def test_prep_sections_against_the_documented_layout(dev, name, E):
    E = np.ascontiguousarray(E, dtype=np.float32)
    K, D = E.shape
    x = _expected(E)
    raw = _build(E, dev)
    T, tf = x["T"], 32 * D + 64
    tiles = raw[:T * tf * 4].view(np.float32).reshape(T, tf)
    assert np.array_equal(tiles[:, :32 * D].view(np.uint32), x["tiles"].view(np.uint32)), "f32 tile images"
    assert np.array_equal(tiles[:, 32 * D:32 * D + 32].reshape(-1), x["en"], equal_nan=True), "norms inside the tiles"
    en_off = T * tf * 4
    assert np.array_equal(raw[en_off:en_off + T * 128].view(np.float32), x["en"], equal_nan=True), "norm array"
    f16 = (en_off + T * 128 + 255) // 256 * 256
    meta_i, meta_f = raw[f16:f16 + 24].view(np.int32), raw[f16:f16 + 24].view(np.float32)
    assert meta_i[0] == x["ok"]
    if not x["ok"]:
        return                                                  # every token goes to the exact list; the images are not used
    assert meta_i[1] == x["b"] and meta_f[2] == x["sb"]
    assert meta_f[3] == x["emax"] and meta_f[4] == x["enmax"]
    eta = float(meta_f[5])
    assert eta >= x["eta_true"] and eta <= x["eta_true"] * 1.002 + 1e-30, (eta, x["eta_true"])
    tb = D * 64 + 256
    o16 = (T * tb + 255) // 256 * 256
    for nm, off, want in (("32x32x16 image", f16 + 256, x["img8"]), ("16x16x32 image", f16 + 256 + o16, x["img16"])):
        img = raw[off:off + T * tb].reshape(T, tb)
        assert np.array_equal(img[:, :D * 64].copy().view(np.uint16).reshape(want.shape), want.view(np.uint16)), nm
        tail = img[:, D * 64:].copy().view(np.float32).reshape(T, 64)
        assert np.array_equal(tail[:, :32].reshape(-1).view(np.uint32), x["seeds"].view(np.uint32)), nm + " seeds"
        assert not tail[:, 32:].any(), nm + " tail padding"
