"""Diagnostic: per-kernel durations of the filter path under a few ablations (run under
rocprofv3 --kernel-trace; tools/parse_trace.py groups the dispatches by config)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device('cuda:0')
B = 256
E1024 = synth.codebook_trained(1024, 256)
z = torch.from_numpy(synth.z_tokens(E1024, B, 32, 32, 2003)).to(dev)
configs = [("K1024 full", 1024, True, True), ("K1024 codes-only", 1024, False, False),
           ("K32 full", 32, True, True), ("K32 codes-only", 32, False, False), ("K256 codes-only", 256, False, False)]
for name, K, wz, wl in configs:
    Et = torch.from_numpy(E1024[:K].copy()).to(dev)
    p = _CodebookPrep()
    for _ in range(5):
        vq_assign(z, Et, p, None, want_zq=wz, want_loss=wl, mode=_lib.MODE_FILTER)
    torch.cuda.synchronize()
    print(name, "fallback", p.fallback_count())
