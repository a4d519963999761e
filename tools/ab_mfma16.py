"""A/B of the code-loop MFMA shape (DVQ_MFMA16 = 0: 32x32x16, 1: 16x16x32) on the dense and the select-fused pass 1,
K = 1024 and 16384; results compared."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
THR = 1.6777750253677368
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n * 1e3, 1)
out = {}
for K, B in ((1024, 256), (16384, 64)):
    En = synth.codebook_trained(K, 256)
    b0 = 32
    tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
    hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)
    zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
    grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
    gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)
    ref = {}
    for rep in range(2):
        for m16 in ("0", "1"):
            os.environ["DVQ_MFMA16"] = m16
            prep = _CodebookPrep()
            if K == 1024:      # K = 16384 at this size dispatches the dense op to the wide kernel
                d = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
                vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER, out=(zq, codes, loss)); torch.cuda.synchronize()
                cur = (zq.clone(), codes.clone())
                ref.setdefault("d", cur)
                out.setdefault("K%d_dense_mfma16=%s" % (K, m16), []).append({"pass1_us": d, "same": bool(torch.equal(cur[0], ref["d"][0]) and torch.equal(cur[1], ref["d"][1])), "queue": prep.fallback_count()})
            o = (zq, codes, None, grain, cmask, gate)
            f = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1, out=o))
            o = (zq, codes, loss, grain, cmask, gate)
            step = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER, out=o))
            torch.cuda.synchronize()
            cur = (zq.clone(), codes.clone())
            ref.setdefault("f", cur)
            out.setdefault("K%d_fused_mfma16=%s" % (K, m16), []).append({"pass1_us": f, "step_us": step, "same": bool(torch.equal(cur[0], ref["f"][0]) and torch.equal(cur[1], ref["f"][1])), "queue": prep.fallback_count()})
os.environ["DVQ_MFMA16"] = "0"
print(json.dumps(out))
