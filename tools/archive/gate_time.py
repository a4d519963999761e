"""Wall time (HIP events) of the feature-router gate op at the sizes VERDICT r3 item 2 names.  usage: python tools/gate_time.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
res = {}
for nb, B in ((2, 64), (3, 128), (2, 256), (3, 1024)):
    if nb == 2:
        r = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
        hs = dict(h_fine=torch.randn(B, 256, 32, 32, generator=g).to(dev), h_coarse=torch.randn(B, 256, 16, 16, generator=g).to(dev))
    else:
        r = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
        hs = dict(h_fine=torch.randn(B, 256, 32, 32, generator=g).to(dev), h_median=torch.randn(B, 256, 16, 16, generator=g).to(dev),
                  h_coarse=torch.randn(B, 256, 8, 8, generator=g).to(dev))
    with torch.no_grad():
        for _ in range(30):
            r(**hs)
        torch.cuda.synchronize()
        n = 200 if B <= 256 else 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(n):
                r(**hs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    res["%s_B%d" % ("dual" if nb == 2 else "triple", B)] = {"us_per_op_median": sorted(ts)[2], "min": min(ts), "max": max(ts)}
    del hs, r
    torch.cuda.empty_cache()
print(json.dumps(res))
