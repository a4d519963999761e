import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K, D, B = 1024, 256, 256
En = synth.codebook_trained(K, D); E = t(En)
b0 = 32
tile = lambda a: torch.cat([torch.roll(a, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf = tile(t(synth.z_tokens(En, b0, 32, 32, 2903)))
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
def timeit(fn, n=50, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
p1, p2 = _CodebookPrep(), _CodebookPrep()
for rep in range(2):
    a = timeit(lambda: vq_assign(hf, E, p1, None, mode=_lib.MODE_FILTER, out=(zq, codes, loss)))
    c1 = codes.clone()
    b = timeit(lambda: vq_assign(hf, E, p2, None, mode=_lib.MODE_FILTER_WIDE, out=(zq, codes, loss)))
    print(json.dumps({"normal_whole_op_us": round(a, 1), "wide_whole_op_us": round(b, 1), "same_codes": bool(torch.equal(c1, codes))}))
    a = timeit(lambda: vq_assign(hf, E, p1, None, mode=_lib.MODE_FILTER, out=(None, codes, None), want_zq=False, want_loss=False))
    b = timeit(lambda: vq_assign(hf, E, p2, None, mode=_lib.MODE_FILTER_WIDE, out=(None, codes, None), want_zq=False, want_loss=False))
    print(json.dumps({"codes_only_normal_us": round(a, 1), "codes_only_wide_us": round(b, 1)}))
