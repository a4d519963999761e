"""workload for rocprofv3 --kernel-trace --stats: routed pass 1 (PASS1 mode), mixed gate, B = 256, K = 1024, form given
by DVQ_ROUTED_DEDUP; argv[1] = zq | codes"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
B, K = 256, 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf, hc, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), t(En)
gmix = tile(t(synth.grain_gate_dual(77, b0, 16, 16)))
zq = torch.empty_like(hf) if sys.argv[1] == "zq" else None
codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
prep = _CodebookPrep()
for _ in range(120):
    vq_assign_routed_dual(hc, hf, E, prep, gate=gmix, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None, grain, cmask, None))
torch.cuda.synchronize()
