"""Randomised parity sweep of the ROUTED ops: DVQ_MODE_FILTER (select fused into pass 1, per-lane and LDS-staged forms, coarse-cell
de-dup in the queue, resolver, list) must reproduce DVQ_MODE_EXACT bit for bit -- codes, z_q, grain indices, codebook_mask, loss to
1e-6 of its scale -- and both must equal the unfused chain route_select_* -> vq_assign on the same inputs.  With conv (every other case at
D = 256): the fused-conv form against dvq_qconv_f32 -> vq_assign on its output (bit-exact GIVEN that h).
The exact mode is pinned to the oracle by tests/; this sweep hunts for inputs on which a fused form disagrees with it.
(tools/fuzz_routed.py is the sweep with the folded conv; this one adds the routed op's own EXACT mode and tiny / huge / default-init
codebooks up to K = 4096.)   usage: python tools/fuzz_routed_modes.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual, vq_assign_routed_triple
from dynamicvectorquantization_amd.router import route_select_dual, route_select_triple


def eq(a, b):
    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


def run(ncases=100, seed=2024, verbose=True):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    bad = 0
    nq = []
    for case in range(ncases):
        G = int(rng.choice([2, 2, 3]))
        D = int(rng.choice([64, 128, 256, 256]))
        K = int(rng.choice([5, 33, 100, 333, 1024, 1024, 2048, 4096]))
        B = int(rng.choice([1, 2, 3, 5, 9]))
        if G == 2:
            hc, wc = int(rng.choice([1, 2, 3, 5, 8, 16])), int(rng.choice([1, 2, 4, 7, 8, 16]))
        else:
            hc, wc = int(rng.choice([1, 2, 3, 4, 8])), int(rng.choice([1, 2, 3, 4, 8]))
        S = 1 << (G - 1)
        kind = rng.choice(["trained", "default", "mixed", "dups", "tiny", "huge"])
        E = synth.codebook_trained(K, D, seed=int(rng.integers(1 << 30)))
        if kind == "default":
            E = synth.codebook_default_init(K, D, seed=int(rng.integers(1 << 30)))
        elif kind == "mixed":
            E = E * np.exp2(rng.integers(-6, 6, size=(K, 1))).astype(np.float32)
        elif kind == "dups":
            idx = rng.integers(0, K, size=K // 2 + 1); E[idx] = E[(idx + 1) % K] * np.float32(1 + 2.0 ** -rng.integers(10, 24))
        elif kind == "tiny":
            E = E * np.float32(1e-12)
        elif kind == "huge":
            E = E * np.float32(3e9)
        E = np.ascontiguousarray(E)
        zs = np.float32(np.exp2(rng.integers(-10, 10))) if rng.random() < 0.3 else np.float32(1.0)
        hs = [synth.z_tokens(E, B, hc << g, wc << g, int(rng.integers(1 << 30))) * zs for g in range(G)]   # coarse -> fine
        if rng.random() < 0.1:
            hs[-1].reshape(-1)[rng.integers(0, hs[-1].size, size=2)] = [np.nan, np.inf]
        # gate: int64 one-hot, f32 logits (ties, NaN), or (dual) the entropy router
        gk = rng.choice(["onehot", "logits", "entropy"]) if G == 2 else rng.choice(["onehot", "logits"])
        thr = 1.6777750253677368
        if gk == "onehot":
            g = rng.integers(0, G, size=(B, hc, wc))
            gate = np.eye(G, dtype=np.int64)[g]
        elif gk == "logits":
            gate = synth.normal(int(rng.integers(1 << 30)), (B, hc, wc, G))
            gate[0, 0, 0] = 0.25
            if rng.random() < 0.3:
                gate[-1, -1, -1, G - 1] = np.nan
        else:
            gate = synth.entropy_map(int(rng.integers(1 << 30)), B, hc, wc)
            gate[0, 0, 0] = np.float32(thr)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        Et, ht = t(E), [t(h) for h in hs]
        gt = t(gate)
        use_conv = (D == 256 and case % 2 == 1)
        conv = None
        if use_conv:
            conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
            with torch.no_grad():
                conv.weight.mul_(float(np.exp2(rng.integers(-3, 3))))
        res = {}
        try:
            for name, mode in (("exact", _lib.MODE_EXACT), ("filter", _lib.MODE_FILTER)):
                if use_conv and mode == _lib.MODE_EXACT:
                    continue                                   # the fused conv is a filter-path form; its reference is the unfused chain below
                prep = _CodebookPrep()
                kw = dict(mode=mode, conv=conv) if use_conv else dict(mode=mode)
                if G == 2:
                    if gk == "entropy":
                        r = vq_assign_routed_dual(ht[0], ht[1], Et, prep, entropy=gt, threshold=thr, **kw)
                    else:
                        r = vq_assign_routed_dual(ht[0], ht[1], Et, prep, gate=gt, **kw)
                else:
                    r = vq_assign_routed_triple(ht[0], ht[1], ht[2], Et, prep, gt, **kw)
                res[name] = r
                if mode == _lib.MODE_FILTER:
                    q = prep.fallback_count()
                    nq.append((q[0] + q[1]) / float(B * hc * wc * S * S))
            # the unfused chain: select -> [conv] -> dense assign (exact mode)
            with torch.no_grad():
                if G == 2:
                    g_sel = res["filter"]["gate"] if gk == "entropy" else gt
                    sel = route_select_dual(g_sel, ht[0], ht[1])
                    hsel = sel["h_dual"]
                else:
                    sel = route_select_triple(gt, ht[0], ht[1], ht[2])
                    hsel = sel["h_triple"]
                if use_conv:
                    from dynamicvectorquantization_amd.qconv import quant_conv
                    hsel = quant_conv(conv, hsel)
                zq0, c0, l0 = vq_assign(hsel, Et, _CodebookPrep(), sel["codebook_mask"], mode=_lib.MODE_EXACT)
            torch.cuda.synchronize()
            ok = True
            why = []
            for name, r in res.items():
                if not torch.equal(r["indices"], sel["indices"]): ok = False; why.append(name + ":indices")
                if not torch.equal(r["codebook_mask"], sel["codebook_mask"]): ok = False; why.append(name + ":mask")
                if not torch.equal(r["codes"].reshape(c0.shape), c0): ok = False; why.append(name + ":codes")
                if not eq(r["zq"], zq0): ok = False; why.append(name + ":zq")
                a, b = float(l0[1]), float(r["loss"][1])
                with np.errstate(all="ignore"):
                    scale = abs(a) + 1.25 * float(np.nanmean(np.square(np.where(np.isfinite(hs[-1]), hs[-1], 0), dtype=np.float64))
                                                  + np.mean(np.square(E, dtype=np.float64)))
                if not ((np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= 1e-6 * scale): ok = False; why.append(name + ":loss %g %g" % (a, b))
        except Exception as e:                                  # an op refusing a case is a finding too
            ok, why = False, ["exception " + repr(e)[:200]]
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(G=G, D=D, K=K, B=B, hc=hc, wc=wc, kind=str(kind), gate=str(gk), conv=use_conv, zs=float(zs)), why, flush=True)
    if verbose:
        print("cases", ncases, "mismatches", bad, "mean queued fraction %.3f" % (float(np.mean(nq)) if nq else 0.0))
    return bad, ncases


if __name__ == "__main__":
    bad, n = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    sys.exit(1 if bad else 0)
