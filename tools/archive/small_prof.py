"""rocprofv3 target: the dense assign op at one small size, 300 back-to-back ops.  usage: python3 tools/small_prof.py B H"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, quantize, synth

B, H = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
K, D = 1024, 256
En = synth.codebook_trained(K, D)
cb = torch.from_numpy(En).to(dev)
z = torch.from_numpy(synth.z_tokens(En, B, H, H, 500 + B)).to(dev)
prep = quantize._CodebookPrep()
out = quantize.vq_assign(z, cb, prep)
for _ in range(300):
    quantize.vq_assign(z, cb, prep, out=out)
torch.cuda.synchronize()
