"""pass 1 alone (dense and routed, B = 256, K = 1024): microseconds per launch, HIP events over 100 launches each on a zeroed
workspace (bench._pass1_ms), for A/B of library variants (DVQ_LIBRARY=...)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _pass1_ms
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
B, K, D = 256, 1024, 256
En = synth.codebook_trained(K, D)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)
THR = 1.6777750253677368
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.ones((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)
pd, pr = _CodebookPrep(), _CodebookPrep()
dense = lambda: vq_assign(hf, E, pd, cmask, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None))
routed = lambda: vq_assign_routed_dual(hc, hf, E, pr, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None, grain, cmask, gate))
for _ in range(300): dense()
torch.cuda.synchronize()
r = {"lib": os.path.basename(os.environ.get("DVQ_LIBRARY", "product"))}
for rep in range(2):
    r["dense_us_%d" % rep] = round(1e3 * _pass1_ms(dense, pd, n=100), 1)
    r["routed_us_%d" % rep] = round(1e3 * _pass1_ms(routed, pr, n=100), 1)
print(json.dumps(r))
