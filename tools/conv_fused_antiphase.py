import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K, D, B = 1024, 256, 256
En = synth.codebook_trained(K, D); E = t(En)
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
with torch.no_grad():
    conv.weight.copy_(t(synth.normal(6012, (D, D, 1, 1), 0.0, 1.0 / 16.0))); conv.bias.copy_(t(synth.normal(6013, (D,), 0.0, 0.1)))
b0 = 32
tile = lambda a: torch.cat([torch.roll(a, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))); hc = tile(t(synth.z_tokens(En, b0, 16, 16, 2913))); ent = tile(t(synth.entropy_map(5903, b0, 16, 16)))
prep = _CodebookPrep(); thr = 1.6777750253677368
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)
def timeit(fn, n=50, warm=200):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fr = lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=thr, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None, grain, cmask, gate), conv=conv)
fd = lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None), conv=conv)
for anti in (0, 1, 0, 1):
    _lib.lib.dvq_tuning_set(b"antiphase", anti)
    print(json.dumps({"antiphase": anti, "fused_routed_pass1_us": round(timeit(fr), 1), "fused_dense_pass1_us": round(timeit(fd), 1)}))
