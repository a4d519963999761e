"""Same-box A/B of the whole BASELINE configs[2] step (B = 256, K = 1024), us per step and per pass-1 launch:
  select+dense   route select kernel (writes h_dual) + dense assign, legacy pass 1        (round-1 path)
  fused          routed op, one token per output position, select fused into the legacy pass 1
  fused-lowreg   the same on low-register pass-1 variant v
  dedup          routed op on unique tokens, low-register variant v
  rows-dedup     routed op on unique tokens, whole rows of cells per legacy workgroup, z_q staged through LDS
Every form's codes / z_q / grain / mask are compared with the first one's."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.router import route_select_dual_entropy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
THR = 1.6777750253677368
En = synth.codebook_trained(K, 256)
b0 = min(B, 32)
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)
h_dual = torch.empty_like(hf); grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev)
cmask = torch.empty((B, 1, 32, 32), device=dev); zq = torch.empty_like(hf)
codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)
def timeit(fn, n=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n * 1e3, 1)
prep = _CodebookPrep()
def select_dense(mode):
    route_select_dual_entropy(ent, THR, hc, hf, out=(h_dual, grain, cmask, gate))
    vq_assign(h_dual, E, prep, cmask, mode=mode, out=(zq, codes, loss if mode == _lib.MODE_FILTER else None))
def routed(mode):
    vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, mode=mode,
                          out=(zq, codes, loss if mode == _lib.MODE_FILTER else None, grain, cmask, gate))
forms = [("select+dense", None, None, None, select_dense), ("fused", "0", "0", 0, routed)]
forms += [("fused-lowreg-v%d" % v, "0", "1", v, routed) for v in (1,)]
forms += [("dedup-v%d" % v, "1", "0", v, routed) for v in (1,)]
forms += [("rows-dedup", "2", "0", 0, routed)]
out, ref = {"B": B, "K": K}, None
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for name, dedup, lowreg, v, fn in forms:
        if dedup is not None:
            os.environ["DVQ_ROUTED_DEDUP"], os.environ["DVQ_ROUTED_DENSE_LOWREG"] = dedup, lowreg
            _lib.lib.dvq_set_pass1_variant(-1, v)
        step = timeit(lambda: fn(_lib.MODE_FILTER))
        p1 = timeit(lambda: fn(_lib.MODE_FILTER_PASS1))
        fn(_lib.MODE_FILTER); torch.cuda.synchronize()
        cur = (zq.clone(), codes.clone(), grain.clone(), cmask.clone(), float(loss[1]))
        if ref is None: ref = cur
        same = all(torch.equal(a, b) for a, b in zip(cur[:4], ref[:4])) and abs(cur[4] - ref[4]) <= 1e-6 * abs(ref[4])
        out.setdefault(name, []).append({"step_us": step, "through_pass1_us": p1, "same": bool(same)})
print(json.dumps(out))
