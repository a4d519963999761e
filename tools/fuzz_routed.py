"""Randomised parity sweep of the ROUTED ops (select fused into pass 1; per-lane and LDS-staged select; the cached / non-temporal
kernel pair; buffer addressing on ragged grids): for random grid sizes (odd widths, single cells, 32-wide), dual / triple, gate as
logits / int64 / entropy + threshold, random scales, special values:
  (1) fused routed op (filter mode)  ==  route_select kernel -> dense assign in EXACT mode: codes, z_q, indices, codebook_mask bit
      for bit, loss to 1e-6 of its scale;
  (2) with a 1x1 conv (256 channels): fused conv op == dvq_qconv_select -> dense assign (filter) bit for bit, and the FOLDED op has
      the same codes with z_q = E[code].
The pieces on the right are pinned to the oracle / reference goldens by tests/.   usage: python tools/fuzz_routed.py [cases]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual, vq_assign_routed_triple
from dynamicvectorquantization_amd.router import route_select_dual, route_select_dual_entropy, route_select_triple
from dynamicvectorquantization_amd.qconv import quant_conv

THR = 1.6777750253677368


def run(ncases=200, seed=2468):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bad, done, conv_cases = 0, 0, 0
    for case in range(ncases):
        G = int(rng.choice([2, 2, 3]))
        D = int(rng.choice([64, 128, 256, 256, 256]))
        K = int(rng.choice([5, 32, 100, 512, 1024, 1024, 2048]))
        B = int(rng.choice([1, 2, 3, 5]))
        hc, wc = (int(rng.choice([1, 2, 3, 5, 8, 16])), int(rng.choice([1, 3, 4, 7, 8, 16]))) if rng.random() < 0.7 else \
                 ((16, 16) if G == 2 else (8, 8))                      # 32-wide output grid: the LDS-staged select
        S = 2 if G == 2 else 4
        E = synth.codebook_trained(K, D, seed=int(rng.integers(1 << 30)))
        kind = rng.choice(["trained", "dups", "mixed"])
        if kind == "dups" and K > 4:
            idx = rng.integers(0, K, size=K // 3 + 1); E[idx] = E[(idx + 1) % K]
        elif kind == "mixed":
            E = E * np.exp2(rng.integers(-4, 4, size=(K, 1))).astype(np.float32)
        zs = np.float32(np.exp2(rng.integers(-6, 6))) if rng.random() < 0.3 else np.float32(1.0)
        hf = synth.z_tokens(E, B, S * hc, S * wc, int(rng.integers(1 << 30))) * zs
        hco = synth.z_tokens(E, B, hc, wc, int(rng.integers(1 << 30))) * zs
        hm = synth.z_tokens(E, B, 2 * hc, 2 * wc, int(rng.integers(1 << 30))) * zs if G == 3 else None
        if rng.random() < 0.1:
            hf.reshape(-1)[rng.integers(0, hf.size, size=2)] = [np.nan, np.inf]
        Et = t(E)
        use_conv = D == 256 and rng.random() < 0.35
        conv = None
        if use_conv:
            conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
            with torch.no_grad():
                conv.weight.copy_(t(synth.normal(int(rng.integers(1 << 30)), (D, D, 1, 1), 0.0, 1.0 / 16.0)))
                conv.bias.copy_(t(synth.normal(int(rng.integers(1 << 30)), (D,), 0.0, 0.1)))
        # ---- the gate
        gate_kind = rng.choice(["logits", "i64", "entropy"]) if G == 2 else rng.choice(["logits", "i64"])
        kw = {}
        if gate_kind == "entropy":
            ent = synth.entropy_map(int(rng.integers(1 << 30)), B, hc, wc)
            ent.reshape(-1)[0] = np.float32(THR)
            kw = dict(entropy=t(ent), threshold=THR)
        else:
            lg = synth.normal(int(rng.integers(1 << 30)), (B, hc, wc, G))
            if rng.random() < 0.3:
                lg.reshape(-1, G)[rng.integers(0, B * hc * wc, size=2)] = 0.25      # ties -> first index
            if gate_kind == "i64":
                oh = np.zeros((B, hc, wc, G), np.int64)
                np.put_along_axis(oh, lg.argmax(-1)[..., None], 1, -1)
                kw = dict(gate=t(oh))
            else:
                kw = dict(gate=t(lg))
        with torch.no_grad():
            if G == 2:
                fused = vq_assign_routed_dual(t(hco), t(hf), Et, _CodebookPrep(), conv=conv, **kw)
                sel = route_select_dual_entropy(kw["entropy"], THR, t(hco), t(hf)) if gate_kind == "entropy" else \
                    route_select_dual(kw["gate"], t(hco), t(hf))
                hsel = sel["h_dual"]
            else:
                fused = vq_assign_routed_triple(t(hco), t(hm), t(hf), Et, _CodebookPrep(), kw["gate"], conv=conv)
                sel = route_select_triple(kw["gate"], t(hco), t(hm), t(hf))
                hsel = sel["h_triple"]
            if use_conv:
                conv_cases += 1
                h = quant_conv(conv, hsel)
                zq0, c0, l0 = vq_assign(h, Et, _CodebookPrep(), sel["codebook_mask"], mode=_lib.MODE_FILTER)
                if G == 2:
                    fold = vq_assign_routed_dual(t(hco), t(hf), Et, _CodebookPrep(), conv=conv, fold=True, want_loss=False, **kw)
                else:
                    fold = vq_assign_routed_triple(t(hco), t(hm), t(hf), Et, _CodebookPrep(), kw["gate"], conv=conv, fold=True, want_loss=False)
            else:
                zq0, c0, l0 = vq_assign(hsel, Et, _CodebookPrep(), sel["codebook_mask"], mode=_lib.MODE_EXACT)
        torch.cuda.synchronize()
        eqn = lambda a, b: bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
        flags = dict(codes=torch.equal(fused["codes"], c0), zq=eqn(fused["zq"], zq0), indices=torch.equal(fused["indices"], sel["indices"]),
                     cmask=torch.equal(fused["codebook_mask"], sel["codebook_mask"]))
        a, b = float(l0[1]), float(fused["loss"][1])
        flags["loss"] = bool((np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= 2e-6 * (abs(a) + 1e-30))
        ok = all(flags.values())
        if use_conv:
            finite = torch.isfinite(zq0).reshape(B, D, -1).all(1).reshape(c0.shape)
            flags["fold_codes"] = torch.equal(fold["codes"], c0)
            flags["fold_ncodes_diff"] = int((fold["codes"] != c0).sum())
            flags["fused_ncodes_diff"] = int((fused["codes"] != c0).sum())
            zqe = Et[fold["codes"]].permute(0, 3, 1, 2)
            # z_q of the fold: e[code] for the tokens pass 1 decides, fl(h + fl(e - h)) for the resolver's: |z_q - e| <= ulp-level of
            # max(|h|, |e|) per element (north_star's bar is 1e-5)
            hmag = torch.maximum(h.abs(), zqe.abs())
            flags["fold_zq"] = bool((torch.where(finite[:, None], (fold["zq"] - zqe).abs() - 2.5e-7 * hmag - 1e-30,
                                                 -torch.ones_like(zqe)) <= 0).all())
            ok = ok and flags["fold_codes"] and flags["fold_zq"]
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(G=G, D=D, K=K, B=B, hc=hc, wc=wc, kind=str(kind), gate=str(gate_kind), conv=bool(use_conv), zs=float(zs)), a, b, flags)
    print(json.dumps({"cases": done, "with_conv_and_fold": conv_cases, "mismatches": bad, "seed": seed}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 2468) else 0)
