"""ctypes/numpy front-end of the CPU parity oracle (oracle/dvq_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Every function takes and returns numpy arrays (C-contiguous, f32 / int64) and
restates one reference call; the reference file:line each follows is in
dvq_oracle.c's header.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdvq_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "dvq_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libdvq_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.dvq_oracle_sumsq.restype = ctypes.c_float
        _lib.dvq_oracle_sumsq.argtypes = [_f32p, ctypes.c_int, ctypes.c_long]
        _lib.dvq_oracle_vq_assign_nchw.restype = ctypes.c_int
        _lib.dvq_oracle_embed_gather.restype = ctypes.c_int
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def sumsq_rows(v):
    """ATen-order sum of squares of each row of v [N, D] -> [N] f32."""
    v, _ = _f32(v)
    n, d = v.shape
    out = np.empty(n, np.float32)
    L = lib()
    for i in range(n):
        out[i] = L.dvq_oracle_sumsq(v[i].ctypes.data_as(_f32p), d, 1)
    return out


def codebook_norms(E):
    E, pE = _f32(E)
    K, D = E.shape
    en = np.empty(K, np.float32)
    lib().dvq_oracle_codebook_norms(pE, ctypes.c_int(K), ctypes.c_int(D),
                                    en.ctypes.data_as(_f32p))
    return en


def token_distances(z_token, E):
    """Reference-arithmetic distances of ONE token [D] to every code -> [K] f32."""
    z, pz = _f32(z_token)
    E, pE = _f32(E)
    K, D = E.shape
    en = codebook_norms(E)
    d = np.empty(K, np.float32)
    lib().dvq_oracle_token_distances(pz, ctypes.c_long(1), pE, en.ctypes.data_as(_f32p),
                                     ctypes.c_int(D), ctypes.c_int(K), d.ctypes.data_as(_f32p))
    return d


def vq_assign_nchw(z, E, mask=None, want_zq=True, want_dmin=False, out=None):
    """VectorQuantize2 / VectorQuantizer2 forward core on z [B, D, H, W] (or [B, D, HW]).

    Returns dict(codes [B, HW] i64, zq [B, D, ...] f32 or None,
                 sqerr float (sum of (e-z)^2*mask in double), numel int, dmin or None).
    """
    z, pz = _f32(z)
    E, pE = _f32(E)
    B, D = z.shape[0], z.shape[1]
    HW = int(np.prod(z.shape[2:])) if z.ndim > 2 else 1
    K = E.shape[0]
    assert E.shape[1] == D
    pm = None
    if mask is not None:
        mask, pm = _f32(np.asarray(mask).reshape(B, HW))
    if out is not None:                       # (zq, codes) to reuse: fresh 100-MB arrays page-fault every call
        zq, codes = out
        assert zq.shape == z.shape and zq.dtype == np.float32 and codes.shape == (B, HW) and codes.dtype == np.int64
    else:
        zq = np.empty_like(z) if want_zq else None
        codes = np.empty((B, HW), np.int64)
    dmin = np.empty((B, HW), np.float32) if want_dmin else None
    sq = ctypes.c_double(0.0)
    rc = lib().dvq_oracle_vq_assign_nchw(
        pz, pE, pm, ctypes.c_int(B), ctypes.c_int(D), ctypes.c_int(HW), ctypes.c_int(K),
        zq.ctypes.data_as(_f32p) if want_zq else None,
        codes.ctypes.data_as(_i64p), ctypes.byref(sq),
        dmin.ctypes.data_as(_f32p) if want_dmin else None)
    if rc != 0:
        raise RuntimeError("dvq_oracle_vq_assign_nchw failed rc=%d" % rc)
    return dict(codes=codes, zq=zq, sqerr=sq.value, numel=B * HW * D, dmin=dmin)


def vq_loss(sqerr, numel, beta, legacy=False):
    """loss = beta*mean + mean  (quantize2_mask.py:175-179; quantize_vqgan.py:290-295).

    Both means are the same number in the forward pass, so legacy only swaps
    which addend carries beta; evaluated in f32 like the reference."""
    m = np.float32(sqerr / numel)
    b = np.float32(beta)
    if legacy:
        return np.float32(m + np.float32(b * m))
    return np.float32(np.float32(b * m) + m)


def embed_gather(E, idx):
    E, pE = _f32(E)
    idx, pi = _i64(idx)
    K, D = E.shape
    out = np.empty(idx.shape + (D,), np.float32)
    rc = lib().dvq_oracle_embed_gather(pE, ctypes.c_int(K), ctypes.c_int(D), pi,
                                       ctypes.c_long(idx.size), out.ctypes.data_as(_f32p))
    if rc != 0:
        raise IndexError("code index out of range")
    return out


def entropy_gate(entropy, thr):
    ent, pe = _f32(entropy)
    gate = np.empty(ent.shape + (2,), np.int64)
    lib().dvq_oracle_entropy_gate(pe, ctypes.c_long(ent.size), ctypes.c_double(thr),
                                  gate.ctypes.data_as(_i64p))
    return gate


def _gate_ptr(gate):
    if np.issubdtype(gate.dtype, np.integer):
        g = np.ascontiguousarray(gate, dtype=np.int64)
        return g, g.ctypes.data_as(ctypes.c_void_p), 1
    g = np.ascontiguousarray(gate, dtype=np.float32)
    return g, g.ctypes.data_as(ctypes.c_void_p), 0


def route_select_dual(gate, h_coarse, h_fine, out=None):
    """EncoderDual.py:134-149.  gate [B, hc, wc, 2] (f32 logits or int64).  out = (h_dual, indices, mask) to reuse."""
    gate, pg, is_i64 = _gate_ptr(np.asarray(gate))
    hcz, pc = _f32(h_coarse)
    hf, pf = _f32(h_fine)
    B, C, hc, wc = hcz.shape
    assert hf.shape == (B, C, 2 * hc, 2 * wc) and gate.shape == (B, hc, wc, 2)
    if out is not None:
        out, ind, cm = out
        assert out.shape == hf.shape and ind.shape == (B, hc, wc) and cm.shape == (B, 1, 2 * hc, 2 * wc)
    else:
        out = np.empty_like(hf)
        ind = np.empty((B, hc, wc), np.int64)
        cm = np.empty((B, 1, 2 * hc, 2 * wc), np.float32)
    lib().dvq_oracle_route_select_dual(pg, ctypes.c_int(is_i64), pc, pf, ctypes.c_int(B),
                                       ctypes.c_int(C), ctypes.c_int(hc), ctypes.c_int(wc),
                                       out.ctypes.data_as(_f32p), ind.ctypes.data_as(_i64p),
                                       cm.ctypes.data_as(_f32p))
    return dict(h_dual=out, indices=ind, codebook_mask=cm)


def route_select_triple(gate, h_coarse, h_median, h_fine):
    """EncoderTriple.py:148-176.  gate [B, hc, wc, 3]."""
    gate, pg, is_i64 = _gate_ptr(np.asarray(gate))
    hcz, pc = _f32(h_coarse)
    hm, pm = _f32(h_median)
    hf, pf = _f32(h_fine)
    B, C, hc, wc = hcz.shape
    assert hm.shape == (B, C, 2 * hc, 2 * wc) and hf.shape == (B, C, 4 * hc, 4 * wc)
    assert gate.shape == (B, hc, wc, 3)
    out = np.empty_like(hf)
    ind = np.empty((B, hc, wc), np.int64)
    cm = np.empty((B, 1, 4 * hc, 4 * wc), np.float32)
    lib().dvq_oracle_route_select_triple(pg, ctypes.c_int(is_i64), pc, pm, pf, ctypes.c_int(B),
                                         ctypes.c_int(C), ctypes.c_int(hc), ctypes.c_int(wc),
                                         out.ctypes.data_as(_f32p), ind.ctypes.data_as(_i64p),
                                         cm.ctypes.data_as(_f32p))
    return dict(h_triple=out, indices=ind, codebook_mask=cm)
