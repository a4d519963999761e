"""Deterministic synthetic inputs for the VQ hot path.

A counter-based generator (splitmix64 finaliser over seed/index) built from
integer ops and exact IEEE double add/multiply only -- no libm calls -- so the
same (seed, shape) gives bit-identical float32 arrays in the build container
and on the GPU box.  `torch.manual_seed` streams are deliberately not used:
they differ across devices and versions.

Approximate normals are a 4-term Irwin-Hall sum (four 16-bit uniforms from one
64-bit draw), variance-normalised; tails stop at +-3.46 sigma, which is fine for
synthetic feature maps.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_SQRT3 = 1.7320508075688772


def _mix(x):
    x = x.copy()
    x ^= x >> np.uint64(30)
    x *= _M1
    x ^= x >> np.uint64(27)
    x *= _M2
    x ^= x >> np.uint64(31)
    return x


def bits(seed, n, offset=0):
    """n 64-bit words for counters offset .. offset+n-1 of stream `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        base = _mix(np.array([np.uint64(seed) * _GOLDEN + np.uint64(0x632BE59BD9B4E019)],
                             dtype=np.uint64))[0]
        return _mix(idx * _GOLDEN + base)


def uniform(seed, shape, lo=0.0, hi=1.0, offset=0):
    n = int(np.prod(shape))
    u = (bits(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, shape, mean=0.0, std=1.0, offset=0, chunk=1 << 24):
    n = int(np.prod(shape))
    out = np.empty(n, np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        b = bits(seed, m, offset + s)
        acc = np.zeros(m, np.float64)
        for sh in (0, 16, 32, 48):
            acc += ((b >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.float64)
        # sum of four U{0..65535}: mean 2*65535, var 4*(65536^2-1)/12
        x = (acc - 131070.0) * (_SQRT3 / 65536.0)
        out[s:s + m] = (mean + std * x).astype(np.float32)
    return out.reshape(shape)


def randint(seed, shape, high, offset=0):
    n = int(np.prod(shape))
    return (bits(seed, n, offset) % np.uint64(high)).astype(np.int64).reshape(shape)


def bernoulli(seed, shape, p, offset=0):
    return uniform(seed, shape, offset=offset) < np.float32(p)


# ---- the SURVEY section-8d recipes ---------------------------------------------------------

def codebook_trained(K, D=256, seed=1001):
    """'trained-like' codebook, N(0, 0.5^2)."""
    return normal(seed, (K, D), 0.0, 0.5)


def codebook_default_init(K, D=256, seed=1002):
    """reference default init U(-1/K, 1/K) (quantize2_mask.py:155) -- the tie-stress case."""
    return uniform(seed, (K, D), -1.0 / K, 1.0 / K)


def z_tokens(codebook, B, H, W, seed, image_offset=0):
    """z-level VQ input [B, D, H, W]: per token 50 % clustered E[j] + 0.3 N(0,1), 50 % N(0,1).

    image_offset shifts the counters so rank r of an image-parallel job can
    generate images [off, off+B) of the global batch without the others."""
    K, D = codebook.shape
    HW = H * W
    n0 = image_offset * HW
    N = B * HW
    noise = normal(seed, (N, D), offset=n0 * D)
    clustered = bernoulli(seed + 7919, (N,), 0.5, offset=n0)
    j = randint(seed + 15838, (N,), K, offset=n0)
    tok = np.where(clustered[:, None], codebook[j] + np.float32(0.3) * noise, noise)
    return np.ascontiguousarray(tok.reshape(B, HW, D).transpose(0, 2, 1)).reshape(B, D, H, W)


def features(seed, B, C, h, w, image_offset=0):
    return normal(seed, (B, C, h, w), offset=image_offset * C * h * w)


def grain_gate_dual(seed, B, hc, wc, p_fine=0.5, image_offset=0):
    """int64 gate [B, hc, wc, 2] as DualGrainFixedEntropyRouter emits it."""
    fine = bernoulli(seed, (B, hc, wc), p_fine, offset=image_offset * hc * wc)
    return np.stack([~fine, fine], axis=-1).astype(np.int64)


def grain_logits_triple(seed, B, hc, wc, probs=(0.4, 0.3, 0.3), image_offset=0):
    """f32 gate logits [B, hc, wc, 3] whose argmax is categorical(probs)."""
    u = uniform(seed, (B, hc, wc), offset=image_offset * hc * wc)
    g = (u >= probs[0]).astype(np.int64) + (u >= probs[0] + probs[1]).astype(np.int64)
    logits = normal(seed + 1, (B, hc, wc, 3), 0.0, 0.1, offset=image_offset * hc * wc * 3)
    np.put_along_axis(logits, g[..., None], np.float32(2.0) + np.take_along_axis(
        logits, g[..., None], axis=-1), axis=-1)
    return logits


def entropy_map(seed, B, hc, wc, p_noise=0.5, image_offset=0):
    """patch-entropy map [B, hc, wc] shaped like Entropy() on the flat/noise patch mixture
    of section 8d: flat patches H in [0, 0.69], noise patches H in [2.96, 3.27]."""
    off = image_offset * hc * wc
    noisy = bernoulli(seed, (B, hc, wc), p_noise, offset=off)
    lo = uniform(seed + 1, (B, hc, wc), 0.0, 0.69, offset=off)
    hi = uniform(seed + 2, (B, hc, wc), 2.96, 3.27, offset=off)
    return np.where(noisy, hi, lo).astype(np.float32)


def seeded_param(seed, i, key, shape):
    """i-th state_dict entry of a router module, seeded (used for feature-router parity):
    GroupNorm weights ~ 1 + 0.1 N, Linear weights ~ N(0, 1/fan_in), biases ~ 0.1 N."""
    if key.endswith("weight") and len(shape) == 1:
        a = 1.0 + 0.1 * normal(seed + i, shape)
    elif len(shape) == 2:
        a = normal(seed + i, shape, 0.0, 1.0 / np.sqrt(shape[1]))
    else:
        a = 0.1 * normal(seed + i, shape)
    return a.astype(np.float32)


def images_flat_noise(seed, B, size=256, patch=16, p_noise=0.5, image_offset=0):
    """images [B, 3, size, size] in [-1, 1] (section 8d): every patch x patch block is either one
    flat colour (patch entropy ~0..0.7 -> coarse) or U(-1, 1) noise (entropy ~3 -> fine)."""
    g = size // patch
    off = image_offset
    noisy = bernoulli(seed, (B, g, g), p_noise, offset=off * g * g)
    flat = uniform(seed + 1, (B, 3, g, g), -1.0, 1.0, offset=off * 3 * g * g)
    noise = uniform(seed + 2, (B, 3, size, size), -1.0, 1.0, offset=off * 3 * size * size)
    flat_up = flat.repeat(patch, axis=-1).repeat(patch, axis=-2)
    mask = noisy.repeat(patch, axis=-1).repeat(patch, axis=-2)[:, None]
    return np.where(mask, noise, flat_up).astype(np.float32), noisy
