"""rocprofv3 target: the dense assign op at the headline size, 100 ops.  usage: python3 tools/dense_prof.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, quantize, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
K, D = 1024, 256
En = synth.codebook_trained(K, D)
cb = torch.from_numpy(En).to(dev)
z = torch.from_numpy(synth.z_tokens(En, min(B, 32), 32, 32, 77)).to(dev)
z = torch.cat([z] * max(1, B // z.shape[0]), 0).contiguous()
prep = quantize._CodebookPrep()
out = quantize.vq_assign(z, cb, prep)
for _ in range(100):
    quantize.vq_assign(z, cb, prep, out=out)
torch.cuda.synchronize()
print(prep.fallback_count())
