"""Is pass 1 held back by the power-limited matrix clock?  Same kernel, same launch, on random latents vs on
all-zero latents (zero operands toggle no multiplier bits: the part then holds ~2.4 GHz, MI355X guide 'DVFS
give-back' item 1) and on zero latents AND a zero codebook image.  Timing only."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
B, K = 256, 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = 32
z = torch.cat([torch.roll(t(synth.z_tokens(En, b0, 32, 32, 2903)), 5 * k, -1) for k in range(B // b0)], 0).contiguous()
z0 = torch.zeros_like(z)
E, E0 = t(En), torch.zeros(K, 256, device=dev)
zq = torch.empty_like(z); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n * 1e3, 1)
out = {}
for name, zz, EE in (("random", z, E), ("zero_latents", z0, E), ("zero_both", z0, E0), ("random_again", z, E)):
    prep = _CodebookPrep()
    for v in (-1, 0):
        _lib.lib.dvq_set_pass1_variant(v, -2)
        out["%s_v%d_full" % (name, v)] = timeit(lambda: vq_assign(zz, EE, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
        out["%s_v%d_codes_only" % (name, v)] = timeit(lambda: vq_assign(zz, EE, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(None, codes, None)))
_lib.lib.dvq_set_pass1_variant(-1, -2)
print(json.dumps(out))
