"""what the DROP-IN costs on the host.  Per call of
   encode_dual(router, quantize, h_fine, h_coarse, entropy, quant_conv)   (the models' encode glue, model order and conv-free)
   VectorQuantize2.forward(x, codebook_mask)                              (the nn.Module a reference YAML instantiates)
   vq_assign_routed_dual(..., out=preallocated)                           (the functional API bench.py times)
at B = 4 / 16 / 64 / 256: host issue time (perf_counter around the call, queue never full: one sync per call group) next to the GPU
time of the same call (HIP events over a back-to-back loop, and the loop's wall time per call).  host > GPU means the op is
launch-bound from Python at that size.  -> one JSON (profiles/r05_module_overhead.json)"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.encode import encode_dual
from dynamicvectorquantization_amd.quantize import VectorQuantize2, vq_assign_routed_dual
from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter

dev = torch.device("cuda:0")
K, D = 1024, 256
THR = 1.6777750253677368
En = synth.codebook_trained(K, D)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
router = DualGrainFixedEntropyRouter(os.path.join(root, "tests", "golden", "entropy_thresholds_imagenet_train_patch-16.json"), 0.5)
vq = VectorQuantize2(K, D).to(dev).eval()
vq.codebook.weight.data[:-1].copy_(torch.from_numpy(En))
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
out = {"K": K, "D": D, "rows": []}


def measure(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    host = []
    for _ in range(n):                         # host issue time: the GPU is idle again before every call
        t0 = time.perf_counter()
        fn()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):                         # back to back: the slower of host and GPU sets the pace
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    return {"host_issue_us_median": round(float(np.median(host)) * 1e6, 1), "host_issue_us_p10": round(float(np.percentile(host, 10)) * 1e6, 1),
            "loop_us_per_call_events": round(e0.elapsed_time(e1) * 1000 / n, 1), "loop_us_per_call_wall": round(wall * 1e6, 1)}


with torch.no_grad():
    for B in (4, 16, 64, 256):
        b0 = min(B, 32)
        tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
        t = lambda a: torch.from_numpy(a).to(dev)
        hf, hc = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913)))
        ent = tile(t(synth.entropy_map(5903, b0, 16, 16)))
        cm = torch.where(torch.rand(B, 1, 32, 32, device=dev) < 0.5, 1.0, 0.25)
        outs = (torch.empty_like(hf), torch.empty((B, 32, 32), dtype=torch.int64, device=dev), torch.empty(2, device=dev),
                torch.empty((B, 16, 16), dtype=torch.int64, device=dev), torch.empty((B, 1, 32, 32), device=dev),
                torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev))
        cb = vq.codebook
        cases = {
            "functional_routed_out_preallocated": lambda: vq_assign_routed_dual(hc, hf, cb.codes, cb._prep, entropy=ent, threshold=THR, out=outs),
            "functional_routed": lambda: vq_assign_routed_dual(hc, hf, cb.codes, cb._prep, entropy=ent, threshold=THR),
            "encode_dual": lambda: encode_dual(router, vq, hf, hc, entropy=ent),
            "encode_dual_model_order": lambda: encode_dual(router, vq, hf, hc, entropy=ent, quant_conv=conv),
            "VectorQuantize2_forward": lambda: vq(hf, cm),
        }
        if hasattr(vq, "graphed"):
            g = vq.graphed(hf, cm)
            cases["VectorQuantize2_graphed"] = lambda: g(hf, cm)
        for name, fn in cases.items():
            r = measure(fn)
            r.update(B=B, call=name)
            out["rows"].append(r)
            print(json.dumps(r), file=sys.stderr)
print(json.dumps(out))
