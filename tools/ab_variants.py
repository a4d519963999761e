"""Same-box A/B of the pass-1 kernel variants on the BASELINE configs[2] step (B = 256, K = 1024):
  select path  (route select kernel + dense assign): dense pass-1 variant -1 (legacy), 0..3 (low-register)
  routed path  (one op on the unique tokens):          routed variant 0..3
Inputs are generated once; variants are switched at run time with dvq_set_pass1_variant; every variant's
codes / z_q are compared with the first one's.  Prints one JSON object.  Usage: python tools/ab_variants.py [B] [K]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.router import route_select_dual_entropy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(os.environ.get("AB_REPS", "3"))
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
THR = 1.6777750253677368
En = synth.codebook_trained(K, 256)
b0 = min(B, 64)
def tile(x):
    return torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)
h_dual = torch.empty_like(hf); grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev)
cmask = torch.empty((B, 1, 32, 32), device=dev); zq = torch.empty_like(hf)
codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)

def timeit(fn, n=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us

out = {"B": B, "K": K}
ref = None
for _ in range(40):   # clocks up
    route_select_dual_entropy(ent, THR, hc, hf, out=(h_dual, grain, cmask, gate))
out["select_kernel_us"] = timeit(lambda: route_select_dual_entropy(ent, THR, hc, hf, out=(h_dual, grain, cmask, gate)))
for rep in range(reps):
    for v in (-1, 0, 1, 2, 3):
        _lib.lib.dvq_set_pass1_variant(v, -2)
        prep = _CodebookPrep()
        p1 = timeit(lambda: vq_assign(h_dual, E, prep, cmask, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
        op = timeit(lambda: vq_assign(h_dual, E, prep, cmask, mode=_lib.MODE_FILTER, out=(zq, codes, loss)))
        torch.cuda.synchronize()
        if ref is None:
            ref = (zq.clone(), codes.clone(), float(loss[1]))
        same = bool(torch.equal(zq, ref[0]) and torch.equal(codes, ref[1]))
        out.setdefault("dense_v%d" % v, []).append({"pass1_us": round(p1, 1), "op_us": round(op, 1), "same": same, "queue": prep.fallback_count()})
    for v in (0, 1, 2, 3):
        _lib.lib.dvq_set_pass1_variant(-2, v)
        prep = _CodebookPrep()
        o = (zq, codes, None, grain, cmask, gate)
        p1 = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1, out=o))
        o = (zq, codes, loss, grain, cmask, gate)
        op = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER, out=o))
        torch.cuda.synchronize()
        same = bool(torch.equal(zq, ref[0]) and torch.equal(codes, ref[1]))
        out.setdefault("routed_v%d" % v, []).append({"pass1_us": round(p1, 1), "op_us": round(op, 1), "same": same,
                                                      "loss_rel": abs(float(loss[1]) - ref[2]) / ref[2], "queue": prep.fallback_count()})
_lib.lib.dvq_set_pass1_variant(-1, 0)
print(json.dumps(out))
