"""The filter path on a codebook it cannot filter: the reference's default initialisation U(-1/K, 1/K) (quantize2_mask.py:155 --
every distance is ||z||^2 to within the bound, so every token is undecided).  Time per op of DVQ_MODE_FILTER / DVQ_MODE_EXACT and
the fallback counts.  usage: python tools/tie_stress_probe.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, quantize, synth


def t_op(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000.0


dev = torch.device("cuda:0")
K, D = 1024, 256
res = {}
for name, E in (("default_init", synth.codebook_default_init(K, D)), ("trained", synth.codebook_trained(K, D))):
    cb = torch.from_numpy(E).to(dev)
    for B in (16, 256):
        z = torch.from_numpy(synth.z_tokens(synth.codebook_trained(K, D), min(B, 32), 32, 32, 77)).to(dev)
        z = torch.cat([z] * (B // z.shape[0]), 0).contiguous()
        row = {}
        outs = {}
        for mname, mode in (("filter", _lib.MODE_FILTER), ("exact", _lib.MODE_EXACT)):
            prep = quantize._CodebookPrep()
            out = quantize.vq_assign(z, cb, prep, mode=mode)
            row[mname + "_us"] = round(t_op(lambda: quantize.vq_assign(z, cb, prep, mode=mode, out=out)), 1)
            outs[mname] = out
            if mode == _lib.MODE_FILTER:
                row["queued_listed"] = list(prep.fallback_count())
        row["same"] = bool(torch.equal(outs["filter"][1], outs["exact"][1]) and torch.equal(outs["filter"][0], outs["exact"][0]))
        res["%s_B%d" % (name, B)] = row
        print(name, B, row, flush=True)
print(json.dumps(res))
