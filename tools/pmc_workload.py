"""Workload for the PMC (HBM traffic) passes: run under
   rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and again with --pmc WRITE_SIZE)
It launches (1) a calibration kernel with a KNOWN byte count in the same access pattern as the
filter kernel's z read (vq_assign_exact, K=32, codes only: reads N*D*4 bytes with 4-B-per-lane
loads, writes 8 B per token), then (2) the dense filter path on BASELINE configs[2] (B=256, K=1024),
then (3) the routed op (select fused into pass 1, coarse branch staged through LDS) exactly as bench.py runs it, and
(4) -- tuning build only (DVQ_LIBRARY=.../libdvq_tuning.so) -- the routed op with the per-lane select form (round 2's),
whose coarse lines are fetched by every wave that shares them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
dev = torch.device("cuda:0")
B = 256
E = synth.codebook_trained(1024, 256)
b0 = 64
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
t = lambda a: torch.from_numpy(a).to(dev)
z = tile(t(synth.z_tokens(E, b0, 32, 32, 2903)))
hc = tile(t(synth.z_tokens(E, b0, 16, 16, 2913)))
ent = tile(t(synth.entropy_map(5903, b0, 16, 16)))
mask = torch.ones(B, 1, 32, 32, device=dev)
E32 = t(E[:32].copy())
Et = t(E)
p32, p, pr = _CodebookPrep(), _CodebookPrep(), _CodebookPrep()
for _ in range(4):
    vq_assign(z, E32, p32, None, want_zq=False, want_loss=False, mode=_lib.MODE_EXACT)
torch.cuda.synchronize()
for _ in range(4):
    vq_assign(z, Et, p, mask, mode=_lib.MODE_FILTER)
torch.cuda.synchronize()
for _ in range(4):
    vq_assign_routed_dual(hc, z, Et, pr, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER)
torch.cuda.synchronize()
if hasattr(_lib.lib, "dvq_tuning_set"):
    _lib.lib.dvq_tuning_set(b"sel_staged", 0)
    for _ in range(4):
        vq_assign_routed_dual(hc, z, Et, pr, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER)
    torch.cuda.synchronize()
    _lib.lib.dvq_tuning_set(b"sel_staged", 1)
# (5) the model order as one op: select + 1x1 quant_conv + assign (bench.py --path model)
conv = torch.nn.Conv2d(256, 256, 1).to(dev).eval()
with torch.no_grad():
    conv.weight.copy_(t(synth.normal(6012, (256, 256, 1, 1), 0.0, 1.0 / 16.0)))
pm = _CodebookPrep()
for _ in range(4):
    vq_assign_routed_dual(hc, z, Et, pm, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER, conv=conv)
torch.cuda.synchronize()
# (6)-(8) VERDICT r3 item 3: what of the routed pass 1's fetch is NOT branch lines.  Pass 1 only (DVQ_MODE_FILTER_PASS1), each
# variant 4 launches of the SAME kernel as (3) -- tools/pmc_traffic.py tells them apart by dispatch order:
#   (6) the full pass 1 (z_q + loss partials: e rows gathered)      (7) codes only: zq = NULL, partials = NULL -> no e-row gather
#   (8) codes only against a K = 32 codebook (one code tile: the code-image stream is 16 KiB per workgroup instead of 512 KiB)
pp = _CodebookPrep()
for _ in range(4):
    vq_assign_routed_dual(hc, z, Et, pp, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER_PASS1)
torch.cuda.synchronize()
for _ in range(4):
    vq_assign_routed_dual(hc, z, Et, pp, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER_PASS1,
                          want_zq=False, want_loss=False)
torch.cuda.synchronize()
pp32 = _CodebookPrep()
for _ in range(4):
    vq_assign_routed_dual(hc, z, E32, pp32, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER_PASS1,
                          want_zq=False, want_loss=False)
torch.cuda.synchronize()
# (9) the conv folded into the codebook (bench.py --path model_fold / tokens_fold): pass 1 on E W, then the whole op
pf = _CodebookPrep()
for _ in range(4):
    vq_assign_routed_dual(hc, z, Et, pf, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER, conv=conv, fold=True,
                          want_loss=False)
torch.cuda.synchronize()
print("queued/exact dense", p.fallback_count(), "routed", pr.fallback_count(), "fold", pf.fallback_count())
