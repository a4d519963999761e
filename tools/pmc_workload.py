"""Workload for the PMC (HBM traffic) passes: run under
   rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and again with --pmc WRITE_SIZE)
It launches (1) a calibration kernel with a KNOWN byte count in the same access pattern as the
filter kernel's z read (vq_assign_exact, K=32, codes only: reads N*D*4 bytes with 4-B-per-lane
loads, writes 8 B per token), then (2) the filter path on BASELINE configs[2] (B=256, K=1024)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device("cuda:0")
B = 256
E = synth.codebook_trained(1024, 256)
z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 2903)).to(dev)
mask = torch.ones(B, 1, 32, 32, device=dev)
E32 = torch.from_numpy(E[:32].copy()).to(dev)
Et = torch.from_numpy(E).to(dev)
p32, p = _CodebookPrep(), _CodebookPrep()
for _ in range(4):
    vq_assign(z, E32, p32, None, want_zq=False, want_loss=False, mode=_lib.MODE_EXACT)
torch.cuda.synchronize()
for _ in range(4):
    vq_assign(z, Et, p, mask, mode=_lib.MODE_FILTER)
torch.cuda.synchronize()
print("queued/exact", p.fallback_count())
