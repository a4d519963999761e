"""Diagnostic: exact kernel with K=32 (one code tile) = cost of its dword-granular z load / z_q store."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device('cuda:0')
B = 256
E1024 = synth.codebook_trained(1024, 256)
z = torch.from_numpy(synth.z_tokens(E1024, B, 32, 32, 2003)).to(dev)
for name, K, wz, wl in [("K32 full", 32, True, True), ("K32 codes-only", 32, False, False), ("K64 full", 64, True, True)]:
    Et = torch.from_numpy(E1024[:K].copy()).to(dev)
    p = _CodebookPrep()
    for _ in range(5):
        vq_assign(z, Et, p, None, want_zq=wz, want_loss=wl, mode=_lib.MODE_EXACT)
    torch.cuda.synchronize()
