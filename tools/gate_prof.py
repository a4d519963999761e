"""workload for rocprofv3 --kernel-trace --stats: the feature-router gate alone.  argv: nb (2 dual | 3 triple), B"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
nb, B = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
if nb == 2:
    r = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
    hs = dict(h_fine=torch.randn(B, 256, 32, 32, generator=g).to(dev), h_coarse=torch.randn(B, 256, 16, 16, generator=g).to(dev))
else:
    r = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
    hs = dict(h_fine=torch.randn(B, 256, 32, 32, generator=g).to(dev), h_median=torch.randn(B, 256, 16, 16, generator=g).to(dev),
              h_coarse=torch.randn(B, 256, 8, 8, generator=g).to(dev))
with torch.no_grad():
    for _ in range(150):
        out = r(**hs)
torch.cuda.synchronize()
