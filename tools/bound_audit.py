"""Audit of the filter path's error bound W (derivation in vq_assign_filter.hip).

For sampled tokens the pass-1 scores G_j (fp16 MFMA, seeded accumulator, packed index bits), the per-token
threshold 2W and xn come back from the GPU through dvq_debug_filter_scores_f32; the REFERENCE-arithmetic
distances d_j = fl(fl(xn + en_j) - 2 dot_j) come from the CPU oracle; in float64
      truth_j = -2^(b-1) (d_j - xn)
and the claim under audit is |G_j - truth_j| <= W for every (token, code) -- which is what makes
"best - second > 2W  =>  the reference's argmin is pass 1's best" a theorem rather than a heuristic.
Prints the largest observed ratio |G - truth| / W (1.0 would be the edge of the bound).

Usage (GPU box): python tools/bound_audit.py [n_tokens_per_case]
       DVQ_LIBRARY=<...>/libdvq_tuning.so python tools/bound_audit.py [n] --production
--production audits the PRODUCTION pass-1 kernel instead of the restatement kernel: the tuning build stores, per token,
the best score, the runner-up, 2W and the provisional code exactly as vq_assign_filter_kernel computed them (its seeds,
its fragment layout, its top-2 merge); checked are |best - truth(code)| <= W and |second - max_{j != code} truth_j| <= W,
and that no provably-decided token disagrees with the reference argmin.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, synth                       # noqa: E402
from dynamicvectorquantization_amd.quantize import _CodebookPrep            # noqa: E402


def scores(tokens, E, dev):
    """tokens [n, D], E [K, D] numpy -> (G [n, K] f32, W [n] f64, xn [n] f32, scale 2^b)"""
    n, D = tokens.shape
    K = E.shape[0]
    Kpad = (K + 31) // 32 * 32
    tt, Et = torch.from_numpy(np.ascontiguousarray(tokens)).to(dev), torch.from_numpy(E).to(dev)
    prep = _CodebookPrep()
    pbuf = prep.get(Et)
    G = torch.empty((n, Kpad), dtype=torch.float32, device=dev)
    thr = torch.empty(n, dtype=torch.float32, device=dev)
    xn = torch.empty(n, dtype=torch.float32, device=dev)
    sc = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.dvq_debug_filter_scores_f32(tt.data_ptr(), n, pbuf.data_ptr(), D, K, G.data_ptr(), thr.data_ptr(),
                                                    xn.data_ptr(), sc.data_ptr(), _lib.stream_ptr(dev)),
               "dvq_debug_filter_scores_f32")
    torch.cuda.synchronize()
    return G.cpu().numpy()[:, :K], thr.cpu().numpy().astype(np.float64) / 2.0, xn.cpu().numpy(), float(sc.item())


def audit_case(name, tokens, E, dev):
    from oracle import oracle
    G, W, xn, sB = scores(tokens, E, dev)
    n, K = G.shape
    worst, decided, wrong, skipped = 0.0, 0, 0, 0
    for i in range(n):
        if not np.isfinite(W[i]):
            skipped += 1                                  # the kernel sends such tokens to the exact list
            continue
        d = oracle.token_distances(tokens[i], E).astype(np.float64)       # reference fp32 arithmetic
        truth = -0.5 * sB * (d - np.float64(xn[i]))
        err = np.abs(G[i].astype(np.float64) - truth)
        worst = max(worst, float(err.max() / W[i]))
        order = np.argsort(-G[i], kind="stable")
        if G[i][order[0]] - G[i][order[1]] > 2 * W[i]:
            decided += 1
            wrong += int(order[0] != int(np.argmin(d)))
    return {"case": name, "tokens": n, "codes": K, "scale_b": sB, "max_err_over_W": worst, "decided": decided,
            "decided_but_wrong": wrong, "skipped_unscorable": skipped}


def audit_case_production(name, tokens, E, dev):
    """the production kernel's own (best, second, 2W, code) per token, through the tuning build's debug store"""
    from oracle import oracle
    from dynamicvectorquantization_amd.quantize import vq_assign
    n, D = tokens.shape
    K = E.shape[0]
    z = torch.from_numpy(np.ascontiguousarray(tokens.T[None])).to(dev)            # [1, D, n]: NCHW with HW = n
    Et = torch.from_numpy(E).to(dev)
    dbg = torch.full((n, 4), float("nan"), dtype=torch.float32, device=dev)
    prep = _CodebookPrep()
    assert _lib.lib.dvq_tuning_buffers(0, dbg.data_ptr()) == 0
    try:
        vq_assign(z, Et, prep, None, want_zq=False, want_loss=False, mode=_lib.MODE_FILTER_PASS1)
        torch.cuda.synchronize()
    finally:
        _lib.lib.dvq_tuning_buffers(0, 0)
    # scale 2^b and xn as the kernel has them: from the restatement ABI (same prep arithmetic)
    _, W, xn, sB = scores(tokens, E, dev)
    d4 = dbg.cpu().numpy().astype(np.float64)
    worst, decided, wrong, skipped = 0.0, 0, 0, 0
    for i in range(n):
        best, second, thr2W, code = d4[i]
        if not np.isfinite(thr2W) or not np.isfinite(W[i]):
            skipped += 1
            continue
        assert abs(thr2W / 2.0 - W[i]) <= 1e-6 * W[i], "production 2W differs from the restatement's"
        code = int(code)
        d = oracle.token_distances(tokens[i], E).astype(np.float64)
        truth = -0.5 * sB * (d - np.float64(xn[i]))
        others = np.delete(truth, code)
        worst = max(worst, abs(best - truth[code]) / W[i], abs(second - others.max()) / W[i] if K > 1 else 0.0)
        if best - second > 2 * W[i]:
            decided += 1
            wrong += int(code != int(np.argmin(d)))
    return {"case": name, "kernel": "production pass 1", "tokens": n, "codes": K, "scale_b": sB, "max_err_over_W": worst,
            "decided": decided, "decided_but_wrong": wrong, "skipped_unscorable": skipped}


def fold_scores(x, E, Wc, bc, dev):
    """conv inputs x [n, D], codebook E [K, D], conv weight Wc [D, D] / bias bc [D] numpy -> (G' [n, K], W' [n], ||x||^2 [n], 2^b')
    through dvq_fold_prepare_f32 + dvq_debug_fold_scores_f32 (the folded image pass 1 scores against)"""
    n, D = x.shape
    K = E.shape[0]
    Kpad = (K + 31) // 32 * 32
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Et = t(E)
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(t(Wc.reshape(D, D, 1, 1)))
        conv.bias.copy_(t(bc))
    prep = _CodebookPrep()
    _, fbuf = prep.fold(Et, conv)
    G = torch.empty((n, Kpad), dtype=torch.float32, device=dev)
    thr = torch.empty(n, dtype=torch.float32, device=dev)
    xn = torch.empty(n, dtype=torch.float32, device=dev)
    sc = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.dvq_debug_fold_scores_f32(t(x).data_ptr(), n, fbuf.data_ptr(), D, K, G.data_ptr(), thr.data_ptr(),
                                                  xn.data_ptr(), sc.data_ptr(), _lib.stream_ptr(dev)), "dvq_debug_fold_scores_f32")
    torch.cuda.synchronize()
    meta = fbuf[:64].view(torch.float32).cpu().numpy()
    return G.cpu().numpy()[:, :K], thr.cpu().numpy().astype(np.float64) / 2.0, xn.cpu().numpy(), float(sc.item()), meta, conv


def audit_case_fold(name, x, E, Wc, bc, dev):
    """the FOLD bound (dvq_filter.h: dvq_fold_threshold): |G'_j - truth_j(h)| <= W' for EVERY h inside the conv tolerance of the
    real-number conv, truth_j(h) = -2^(b'-1) (d_j(h) - xn(h)) in the reference's fp32 arithmetic on h.  Audited on h = fp32 of the
    float64 conv, on h = dvq_qconv_f32(x) (what the resolver computes) and on two h at the EDGE of the tolerance ball
    (+- 1e-5 (sum |w||x| + |b|) per channel, random signs)."""
    from oracle import oracle
    from dynamicvectorquantization_amd.qconv import quant_conv
    G, W, xnx, sB, meta, conv = fold_scores(x, E, Wc, bc, dev)
    n, K = G.shape
    D = x.shape[1]
    x64, w64, b64 = x.astype(np.float64), Wc.astype(np.float64), bc.astype(np.float64)
    h64 = x64 @ w64.T + b64
    mag = np.abs(x64) @ np.abs(w64).T + np.abs(b64)
    hq = quant_conv(conv, torch.from_numpy(np.ascontiguousarray(x.T[None])).to(dev).reshape(1, D, n, 1)).cpu().numpy()[0, :, :, 0].T
    assert np.all(np.abs(hq - h64) <= 1e-5 * mag + 1e-30), "dvq_qconv_f32 outside its own tolerance"
    rng = np.random.RandomState(12345)
    hs = [h64.astype(np.float32), hq,
          (h64 + 1e-5 * mag * rng.choice([-1.0, 1.0], size=h64.shape)).astype(np.float32),
          (h64 - 1e-5 * mag * np.sign(h64)).astype(np.float32)]
    worst, decided, wrong, skipped = 0.0, 0, 0, 0
    for i in range(n):
        if not np.isfinite(W[i]):
            skipped += 1
            continue
        order = np.argsort(-G[i], kind="stable")
        dec = G[i][order[0]] - G[i][order[1]] > 2 * W[i]
        decided += int(dec)
        for h in hs:
            d = oracle.token_distances(h[i], E).astype(np.float64)
            xn_h = np.float64(oracle.sumsq_rows(h[i][None])[0])      # the reference's own fp32 norm of h
            truth = -0.5 * sB * (d - xn_h)
            worst = max(worst, float(np.abs(G[i].astype(np.float64) - truth).max() / W[i]))
            if dec:
                wrong += int(order[0] != int(np.argmin(d)))
    return {"case": name, "kernel": "fold scores", "tokens": n, "codes": K, "scale_b": sB, "max_err_over_W": worst,
            "decided": decided, "decided_but_wrong": wrong, "skipped_unscorable": skipped,
            "sigma": float(meta[8]), "qmax": float(meta[7]), "emax_fold": float(meta[3])}


def fold_cases(n):
    """(name, conv inputs [n, D], codebook, conv weight, conv bias)"""
    out = []
    D = 256
    E = synth.codebook_trained(1024, D)
    Wg = synth.normal(31, (D, D), 0.0, 1.0 / 16.0)
    bg = synth.normal(32, (D,), 0.0, 0.1)
    q, _ = np.linalg.qr(synth.normal(33, (D, D)).astype(np.float64))
    Wo = np.ascontiguousarray(q.astype(np.float32))
    pre = lambda Wc, bc, seed: np.ascontiguousarray(                                   # inputs whose conv output sits near the codebook
        ((synth.z_tokens(E, 1, 1, n, seed)[0, :, 0, :].T.astype(np.float64) - bc) @ np.linalg.inv(Wc.astype(np.float64)).T).astype(np.float32))
    out.append(("orthogonal conv + bias, h ~ trained-like tokens", pre(Wo, bg, 41), E, Wo, bg))
    out.append(("gaussian conv N(0,1/16) + bias, x ~ N(0,1)", synth.normal(42, (n, D)), E, Wg, bg))
    out.append(("gaussian conv, h ~ trained-like tokens", pre(Wg, bg, 43), E, Wg, bg))
    out.append(("conv x 30, x ~ N(0, 0.03)", synth.normal(44, (n, D), 0.0, 0.03), E, Wg * np.float32(30), bg))
    out.append(("conv x 1e-3, big x", synth.normal(45, (n, D), 0.0, 300.0), E, Wg * np.float32(1e-3), bg * 0))
    Ed = synth.codebook_default_init(1024, D)
    out.append(("default-init codebook, orthogonal conv", synth.normal(46, (n, D), 0.0, 1e-3), Ed, Wo, bg * np.float32(1e-3)))
    E64 = synth.codebook_trained(512, 64, seed=77)
    W64 = synth.normal(47, (64, 64), 0.0, 1.0 / 8.0)
    out.append(("D=64", synth.normal(48, (n, 64)), E64, W64, synth.normal(49, (64,), 0.0, 0.1)))
    return out


def cases(n):
    """(name, tokens [n, D], codebook) -- trained-like data, the tie-stress default init, large / tiny magnitudes,
    fp16-subnormal territory, near-duplicate codes, K = 16384, D = 64"""
    out = []
    E = synth.codebook_trained(1024, 256)
    tok = lambda E_, seed, m=n: synth.z_tokens(E_, 1, 1, m, seed)[0, :, 0, :].T.copy()
    out.append(("trained K=1024", tok(E, 11), E))
    Ed = synth.codebook_default_init(1024, 256)
    out.append(("default-init U(-1/K,1/K), tokens ~ N(0,1)", tok(Ed, 12), Ed))
    out.append(("default-init, tokens at codebook scale", tok(Ed, 13) * np.float32(1e-3), Ed))
    out.append(("tokens x 1e3", tok(E, 14) * np.float32(1e3), E))
    out.append(("tokens x 1e-5 (fp16 subnormals)", tok(E, 15) * np.float32(1e-5), E))
    Em = E * np.exp2(np.arange(1024) % 11 - 5).astype(np.float32)[:, None]
    out.append(("codebook with norms spread over 2^10", tok(Em, 16), Em))
    En = E.copy()
    En[1::2] = En[0::2] * np.float32(1 + 2 ** -12)
    out.append(("near-duplicate code pairs", tok(En, 17), En))
    E16 = synth.codebook_trained(16384, 256)
    out.append(("trained K=16384", tok(E16, 18, max(32, n // 4)), E16))
    E64 = synth.codebook_trained(512, 64, seed=77)
    out.append(("D=64", tok(E64, 19), E64))
    return out


def run_fold(n=64, verbose=True):
    dev = torch.device("cuda:0")
    res = [audit_case_fold(name, x, E, Wc, bc, dev) for name, x, E, Wc, bc in fold_cases(n)]
    if verbose:
        for r in res:
            print(json.dumps(r))
    return res


def run(n=96, verbose=True, production=False):
    dev = torch.device("cuda:0")
    if production:
        assert hasattr(_lib.lib, "dvq_tuning_buffers"), "--production needs DVQ_LIBRARY=<...>/libdvq_tuning.so"
    fn = audit_case_production if production else audit_case
    res = [fn(name, t, E, dev) for name, t, E in cases(n)]
    if verbose:
        for r in res:
            print(json.dumps(r))
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--fold" in sys.argv:
        r = run_fold(int(args[0]) if args else 128)
        print(json.dumps({"max_err_over_W": max(x["max_err_over_W"] for x in r),
                          "decided_but_wrong": sum(x["decided_but_wrong"] for x in r)}))
        sys.exit(0)
    r = run(int(args[0]) if args else 256, production="--production" in sys.argv)
    print(json.dumps({"max_err_over_W": max(x["max_err_over_W"] for x in r),
                      "decided_but_wrong": sum(x["decided_but_wrong"] for x in r)}))
