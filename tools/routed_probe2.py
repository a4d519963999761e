"""Timing-only experiments on the routed pass 1 (mix gate, B = 256, K = 1024): DVQ_P1_DEBUG drops some z_q stores
(results are then wrong on purpose) to see which stores cost what."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
B, K = 256, 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf, hc, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), t(En)
gmix = tile(t(synth.grain_gate_dual(77, b0, 16, 16)))
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
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
prep = _CodebookPrep()
for v in (1, 0):
    _lib.lib.dvq_set_pass1_variant(-2, v)
    for dbg in (0, 1, 2, 4, 6):
        os.environ["DVQ_P1_DEBUG"] = str(dbg)
        out["v%d_debug%d" % (v, dbg)] = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, gate=gmix, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None, grain, cmask, None)))
os.environ["DVQ_P1_DEBUG"] = "0"
print(json.dumps(out))
