"""one-config workload for rocprofv3 --kernel-trace: full filter op at K=16384, B=64"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = synth.codebook_trained(K, 256)
Et = torch.from_numpy(E).to(dev)
z = torch.from_numpy(synth.z_tokens(E, 64, 32, 32, 2005)).to(dev)
p = _CodebookPrep()
for _ in range(6):
    vq_assign(z, Et, p, None, mode=_lib.MODE_FILTER)
torch.cuda.synchronize()
print(p.fallback_count())
