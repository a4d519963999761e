"""fuzz of the row-complete de-duplicated routed form (DVQ_ROUTED_DEDUP=2) against the fused form (0), bit for bit:
random batch sizes, codebook sizes, gate patterns (per-row all-coarse / all-fine / mixed), NaN / huge tokens, both MFMA
loop shapes, with / without z_q and loss"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual, vq_assign_routed_triple
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t = lambda a: torch.from_numpy(a).to(dev)
bad = 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for case in range(ncase):
    triple = bool(rng.integers(0, 2))
    B = int(rng.integers(1, 41))
    K = int(rng.choice([64, 333, 1024, 2048]))
    E = synth.codebook_trained(K, 256, seed=1000 + case)
    hc_ = 8 if triple else 16
    feats = [synth.z_tokens(E, B, hc_ << i, hc_ << i, 5000 + 10 * case + i) for i in range(3 if triple else 2)]
    G = 3 if triple else 2
    lg = rng.standard_normal((B, hc_, hc_, G)).astype(np.float32)
    for b in range(B):                                   # structured rows: all of one grain, or random
        for y in range(hc_):
            m = rng.integers(0, 4)
            if m < G:
                lg[b, y, :, :] = -1.0
                lg[b, y, :, m] = 1.0
    if rng.integers(0, 3) == 0:
        for f in feats:
            f[rng.integers(0, B), :, rng.integers(0, f.shape[2]), rng.integers(0, f.shape[3])] = np.nan
            f[rng.integers(0, B), :, rng.integers(0, f.shape[2]), rng.integers(0, f.shape[3])] *= np.float32(1e6)
    os.environ["DVQ_MFMA16"] = str(int(rng.integers(0, 2)))
    want_zq = bool(rng.integers(0, 4))
    res = []
    for form in ("0", "2"):
        os.environ["DVQ_ROUTED_DEDUP"] = form
        prep = _CodebookPrep()
        if triple:
            r = vq_assign_routed_triple(t(feats[0]), t(feats[1]), t(feats[2]), t(E), prep, t(lg), want_zq=want_zq)
        else:
            r = vq_assign_routed_dual(t(feats[0]), t(feats[1]), t(E), prep, gate=t(lg), want_zq=want_zq)
        torch.cuda.synchronize()
        res.append(r)
    a, b2 = res
    same = torch.equal(a["codes"], b2["codes"]) and torch.equal(a["indices"], b2["indices"]) and torch.equal(a["codebook_mask"], b2["codebook_mask"])
    if want_zq:
        same = same and torch.equal(torch.nan_to_num(a["zq"], nan=7.0), torch.nan_to_num(b2["zq"], nan=7.0))
    la, lb = float(a["loss"][1]), float(b2["loss"][1])
    same = same and (abs(la - lb) <= 1e-5 * abs(la) or (la != la and lb != lb))
    if not same:
        bad += 1
        print("MISMATCH case", case, "triple" if triple else "dual", B, K, want_zq, la, lb)
for k in ("DVQ_ROUTED_DEDUP", "DVQ_MFMA16"):
    os.environ.pop(k, None)
print("cases", ncase, "mismatches", bad)
