"""Small-batch latency of the dense assign op (BASELINE configs[0] is B = 4, 16 x 16, K = 1024, D = 256): HIP-event time per op of
DVQ_MODE_FILTER and DVQ_MODE_EXACT at several token counts.  usage: python tools/small_probe.py [out.json]"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, quantize, synth


def t_op(fn, n=400):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1000.0)
    return best


def main():
    dev = torch.device("cuda:0")
    K, D = 1024, 256
    En = synth.codebook_trained(K, D)
    cb = torch.from_numpy(En).to(dev)
    res = {}
    for (B, H) in ((1, 16), (4, 16), (16, 16), (4, 32), (8, 32), (16, 32), (64, 32)):
        z = torch.from_numpy(synth.z_tokens(En, B, H, H, 500 + B)).to(dev)
        prep = quantize._CodebookPrep()
        row = {}
        outs = {}
        has_switch = hasattr(_lib.lib, "dvq_tuning_set")      # tuning build (DVQ_LIBRARY=.../libdvq_tuning.so): split form on / off
        forms = (("filter_unsplit", _lib.MODE_FILTER, 0), ("filter", _lib.MODE_FILTER, 1)) if has_switch else (("filter", _lib.MODE_FILTER, 1),)
        for name, mode, sp in forms + (("exact", _lib.MODE_EXACT, 1),):
            if has_switch:
                _lib.lib.dvq_tuning_set(b"split", sp)
            out = quantize.vq_assign(z, cb, prep, mode=mode)
            outs[name] = out
            row[name + "_us"] = round(t_op(lambda: quantize.vq_assign(z, cb, prep, mode=mode, out=out), 400 if mode else 100), 2)
            off = _lib.lib.dvq_vq_assign_fallback_count_offset(D, H * H, K, B * H * H) if hasattr(_lib.lib, "dvq_vq_assign_fallback_count_offset") else None
        row["same"] = all(torch.equal(outs[k][0], outs["exact"][0]) and torch.equal(outs[k][1], outs["exact"][1]) and
                          torch.equal(outs[k][2], outs["exact"][2]) for k in outs)
        res["B%d_%dx%d_N%d" % (B, H, H, B * H * H)] = row
        print(B, H, row, flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
