"""Row-complete de-duplicated pass 1 (DVQ_ROUTED_DEDUP=2) against the fused form on one box, B = 256, K = 1024, mixed
gate: through pass 1 with z_q, codes only (no z_q: prologue + code loop), and the all-coarse / all-fine extremes."""
import json, os, sys
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
gco = torch.zeros_like(gmix); gco[..., 0] = 1
gfi = torch.zeros_like(gmix); gfi[..., 1] = 1
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
loss = torch.empty(2, device=dev)
def timeit(fn, n=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n * 1e3, 1)
out = {}
prep = _CodebookPrep()
for rep in range(2):
    for form in ("0", "2"):
        os.environ["DVQ_ROUTED_DEDUP"] = form
        for gname, g in (("mix", gmix), ("all_coarse", gco), ("all_fine", gfi)):
            out.setdefault("form%s_%s_pass1_zq" % (form, gname), []).append(timeit(lambda: vq_assign_routed_dual(
                hc, hf, E, prep, gate=g, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None, grain, cmask, None))))
            out.setdefault("form%s_%s_pass1_codes_only" % (form, gname), []).append(timeit(lambda: vq_assign_routed_dual(
                hc, hf, E, prep, gate=g, mode=_lib.MODE_FILTER_PASS1, out=(None, codes, None, grain, cmask, None))))
        out.setdefault("form%s_mix_step" % form, []).append(timeit(lambda: vq_assign_routed_dual(
            hc, hf, E, prep, gate=gmix, mode=_lib.MODE_FILTER, out=(zq, codes, loss, grain, cmask, None))))
os.environ.pop("DVQ_ROUTED_DEDUP", None)
print(json.dumps(out))
