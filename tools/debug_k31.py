import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device('cuda:0')
for K in (31, 32, 20, 40):
    E = synth.codebook_trained(K, 256, seed=800 + K)
    z = torch.from_numpy(synth.z_tokens(E, 1, 3, 3, 810 + K)).to(dev)
    Et = torch.from_numpy(E).to(dev)
    pe, pf = _CodebookPrep(), _CodebookPrep()
    zq0, c0, l0 = vq_assign(z, Et, pe, None, mode=0)
    zq1, c1, l1 = vq_assign(z, Et, pf, None, mode=1)
    torch.cuda.synchronize()
    print(K, c0.flatten().tolist(), c1.flatten().tolist(), pf.fallback_count())
