"""feature-router gate only (dual B = 64, triple B = 128), for rocprofv3 --kernel-trace --stats"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
r2 = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
r3 = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
hf, hc = t(synth.features(1, 64, 256, 32, 32)), t(synth.features(2, 64, 256, 16, 16))
f3, m3, c3 = t(synth.features(3, 128, 256, 32, 32)), t(synth.features(4, 128, 256, 16, 16)), t(synth.features(5, 128, 256, 8, 8))
with torch.no_grad():
    for _ in range(30):
        r2(h_fine=hf, h_coarse=hc)
    torch.cuda.synchronize()
    for _ in range(30):
        r3(h_fine=f3, h_median=m3, h_coarse=c3)
torch.cuda.synchronize()
