"""Routed dual step (feature router gate + fused select/assign) over batch sizes: where does the plain-load policy stop paying?
usage: [DVQ_LIBRARY=...] python tools/cache_policy_sweep.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.quantize import VectorQuantize2
from dynamicvectorquantization_amd.router import DualGrainFeatureRouter
from dynamicvectorquantization_amd.encode import encode_dual
dev = torch.device("cuda:0")
E = synth.codebook_trained(1024, 256)
vq = VectorQuantize2(1024, 256).to(dev).eval(); vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E).to(dev))
r = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
res = {}
g = torch.Generator().manual_seed(5)
for B in (64, 96, 128, 144, 152, 160, 192, 256):
    hf = torch.randn(B, 256, 32, 32, generator=g).to(dev); hc = torch.randn(B, 256, 16, 16, generator=g).to(dev)
    with torch.no_grad():
        for _ in range(20): encode_dual(r, vq, hf, hc)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(100): encode_dual(r, vq, hf, hc)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 100 * 1e3)
    res["B%d" % B] = round(sorted(ts)[2] / B, 4)          # us per image
    del hf, hc
print(json.dumps(res))
