"""Do a matrix-core kernel and an HBM-streaming kernel from two streams share the CUs?
D=128 filter pass (152 VGPRs: leaves register room) on stream A, route select (34 VGPRs) on stream B."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
from dynamicvectorquantization_amd.router import entropy_gate, route_select_dual
dev = torch.device("cuda:0")
D, K, B = 128, 4096, 256
E = synth.codebook_trained(K, D)
Et = torch.from_numpy(E).to(dev)
z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 11)).to(dev)
hc = torch.from_numpy(synth.z_tokens(E, B, 16, 16, 12)).to(dev)
hf2 = z.clone()
gate = entropy_gate(torch.from_numpy(synth.entropy_map(13, B, 16, 16)).to(dev), 1.6777750253677368)
p = _CodebookPrep()
zq = torch.empty_like(z); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
hd = torch.empty_like(z); gr = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cm = torch.empty((B, 1, 32, 32), device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def A():
    vq_assign(z, Et, p, None, mode=_lib.MODE_FILTER_PASS1, want_loss=False, out=(zq, codes, None))
def Bk():
    route_select_dual(gate, hc, hf2, out=(hd, gr, cm))
def timeit(fa, fb, n=20):
    for _ in range(3):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb(); fb()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb(); fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("A alone (filter pass D=128 K=4096) ms", timeit(A, None))
print("B alone (2x select C=128)          ms", timeit(None, Bk))
print("A || B                             ms", timeit(A, Bk))
