"""Where does the routed pass 1 spend its time?  Event timings of the routed op (pass-1-only mode) at B = 256,
K = 1024 with pieces of the epilogue switched off through the API (no z_q, codes only) and with all-coarse /
all-fine / mixed gates, next to the dense low-register kernel on the equivalent tensors.  Prints JSON."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual

B, K = 256, 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf, hc, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), t(En)
gmix = tile(t(synth.grain_gate_dual(77, b0, 16, 16)))
gfine = torch.zeros_like(gmix); gfine[..., 1] = 1
gcoarse = torch.zeros_like(gmix); gcoarse[..., 0] = 1
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)

def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n * 1e3, 1)

out = {}
variants = [int(v) for v in os.environ.get("PROBE_VARIANTS", "0,1").split(",")]
for v in variants:
    _lib.lib.dvq_set_pass1_variant(v, v)
    prep = _CodebookPrep()
    for gname, g in (("mix", gmix), ("fine", gfine), ("coarse", gcoarse)):
        for oname, o in (("full", (zq, codes, None, grain, cmask, None)), ("nozq", (None, codes, None, grain, cmask, None))):
            out["routed_v%d_%s_%s" % (v, gname, oname)] = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, gate=g, mode=_lib.MODE_FILTER_PASS1, out=o))
    # mode PASS1 with loss=None and zq=None -> codes only (records still written for queued tokens)
    for oname, o in (("full", (zq, codes, None)), ("nozq", (None, codes, None))):
        out["dense_v%d_%s" % (v, oname)] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=o))
        out["dense_v%d_coarse_tensor_%s" % (v, oname)] = timeit(lambda: vq_assign(hc, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(None if o[0] is None else zq[:, :, :16, :16].contiguous(), codes[:, :16, :16].contiguous(), None)))
_lib.lib.dvq_set_pass1_variant(-1, 0)
prep = _CodebookPrep()
out["legacy_full"] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
out["legacy_nozq"] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(None, codes, None)))
print(json.dumps(out, indent=0))
