"""workload of tools/entropy_pmc.sh: the patch-entropy kernel at B = 256 on the section-8d image mixture, 6 launches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.entropy import Entropy
dev = torch.device("cuda:0")
base = torch.from_numpy(synth.images_flat_noise(5000, 32)[0]).to(dev)
img = torch.cat([torch.roll(base, 16 * k, -1) for k in range(8)], 0).contiguous()
ent = Entropy(16, 256, 256).to(dev)
for _ in range(6):
    ent(img)
torch.cuda.synchronize()
