"""First-generation stagger sweep (DVQ_STAGGER_US) for the low-register pass-1 variants, dense tensor, B = 256, K = 1024."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
B, K = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = 32
hf = torch.cat([torch.roll(t(synth.z_tokens(En, b0, 32, 32, 2903)), 5 * k, -1) for k in range(B // b0)], 0).contiguous()
E = t(En)
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
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
for v in (0, 1, 2):
    _lib.lib.dvq_set_pass1_variant(v, -2)
    for us in (0, 8, 14, 20, 26, 32, 40):
        os.environ["DVQ_STAGGER_US"] = str(us)
        out["v%d_stagger%d" % (v, us)] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
os.environ["DVQ_STAGGER_US"] = "0"
_lib.lib.dvq_set_pass1_variant(-1, -2)
out["legacy"] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
print(json.dumps(out))
