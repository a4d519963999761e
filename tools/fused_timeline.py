"""Timeline of the fused step (pass 1 + resolver as consumer workgroups of the same grid) from the tuning build's stamps:
   DVQ_LIBRARY=<...>/libdvq_tuning.so python tools/fused_timeline.py [B] [cps]
s_memrealtime (100 MHz) per workgroup, thread 0: token blocks entry / exit; consumers entry / exit and per chunk: ready, records
in LDS, enumerated, chains done, chunk done.  Steady state: the launch measured follows 100 back-to-back ops."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
assert hasattr(_lib.lib, "dvq_tuning_buffers")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = min(B, 32)
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
prep = _CodebookPrep()
if cps > 0:
    _lib.lib.dvq_tuning_set(b"cps", cps)
def launch():
    if cps < 0:                                    # pass 1 alone (the profiling mode): the baseline the token blocks are compared with
        vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=1.6777750253677368, out=(zq, codes, loss, grain, cmask, gate),
                              mode=_lib.MODE_FILTER_PASS1)
    else:
        vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=1.6777750253677368, out=(zq, codes, loss, grain, cmask, gate))
nb1 = B * 1024 // 128
G = nb1 + 1024
st = torch.zeros((G, 16), dtype=torch.int64, device=dev)
for _ in range(100):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    launch()
e1.record()
torch.cuda.synchronize()
op_us = e0.elapsed_time(e1) * 1000 / 50
_lib.lib.dvq_tuning_buffers(st.data_ptr(), 0)
launch()
torch.cuda.synchronize()
_lib.lib.dvq_tuning_buffers(0, 0)
s = st.cpu().numpy().astype(np.int64)
P = s[:nb1]
t0 = P[:, 0].min()
us = lambda x: (x - t0) / 100.0
pend = us(P[:, 1])
out = {"B": B, "cps": cps, "op_us_events": round(op_us, 2), "queued": prep.fallback_count()[0],
       "producer_entry_us_pct": [round(float(np.percentile(us(P[:, 0]), p)), 1) for p in (0, 25, 50, 75, 100)],
       "producer_exit_us_pct": [round(float(np.percentile(pend, p)), 1) for p in (0, 1, 5, 25, 50, 75, 95, 99, 100)],
       "producer_block_us_median": round(float(np.median(pend - us(P[:, 0]))), 1)}
Cn = s[nb1:]
Cn = Cn[Cn[:, 0] > 0]
T_end = float(pend.max())
if Cn.shape[0] == 0:
    out["kernel_span_us"] = round(T_end, 1)
    print(json.dumps(out))
    sys.exit(0)
out["n_consumers"] = int(Cn.shape[0])
out["consumer_entry_minus_Tend_pct"] = [round(float(np.percentile(us(Cn[:, 0]) - T_end, p)), 1) for p in (0, 5, 25, 50, 75, 95, 100)]
out["consumer_exit_minus_Tend_pct"] = [round(float(np.percentile(us(Cn[:, 1]) - T_end, p)), 1) for p in (0, 5, 25, 50, 75, 95, 100)]
for c in (0, 1):
    b = 2 + 7 * c
    m = (Cn[:, b] > 0) & (Cn[:, b + 5] > 0) & (Cn[:, b + 4] > 0)
    if not m.any():
        continue
    X = Cn[m]
    ph = lambda a, z: [round(float(np.percentile((X[:, z] - X[:, a]) / 100.0, p)), 2) for p in (5, 50, 95)]
    out["chunk%d" % c] = {"n": int(m.sum()), "nlive_median": float(np.median(X[:, b + 5])),
                          "ready_minus_Tend_p5_50_95": [round(float(np.percentile(us(X[:, b]) - T_end, p)), 1) for p in (5, 50, 95)],
                          "wait_for_chunk_us": ph(0 if c == 0 else b - 3, b) if c == 0 else ph(b - 3, b),
                          "load_records_us": ph(b, b + 1), "enumerate_us": ph(b + 1, b + 2), "chains_us": ph(b + 2, b + 3),
                          "winners_rewrite_us": ph(b + 3, b + 4), "total_us": ph(b, b + 4)}
out["last_consumer_exit_minus_Tend_us"] = round(float(us(Cn[:, 1]).max() - T_end), 1)
out["kernel_span_us"] = round(float(max(us(Cn[:, 1]).max(), T_end)), 1)
print(json.dumps(out))
