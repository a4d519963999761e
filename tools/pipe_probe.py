"""Phase timeline of the persistent pass-1 form (vq_assign_pipe.hip) from the tuning build's stamps:
   DVQ_LIBRARY=<...>/libdvq_tuning.so python tools/pipe_probe.py [dense|routed]
Per group: duration of every phase (us), and inside the memory phases the time to step 16 (epilogue part) and to step 24
(loads issued) -- medians over the workgroups of one launch in steady state."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
assert hasattr(_lib.lib, "dvq_tuning_pipe_stamps")
kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
B, K = int(os.environ.get("AB_B", "256")), 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, 256)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)
prep = _CodebookPrep()
_lib.lib.dvq_tuning_set(b"pipe", 1)
def launch():
    if kind == "dense":
        vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None))
    else:
        vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=1.6777750253677368, mode=_lib.MODE_FILTER_PASS1,
                              out=(zq, codes, None, grain, cmask, gate))
G = 256
st = torch.zeros((G, 2, 64), dtype=torch.int64, device=dev)
for _ in range(200):
    launch()
_lib.lib.dvq_tuning_pipe_stamps(st.data_ptr())
launch()
torch.cuda.synchronize()
_lib.lib.dvq_tuning_pipe_stamps(0)
s = st.cpu().numpy().astype(np.int64)
out = {"kind": kind, "B": B}
for g in (0, 1):
    starts = s[:, g, 0::4]                      # [G, 16]
    n = int((starts[0] > 0).sum())
    dur = np.diff(starts[:, :n], axis=1) / 100.0
    out["group%d_phase_us_median" % g] = [round(float(x), 2) for x in np.median(dur, axis=0)]
    s16 = (s[:, g, 1::4][:, :n - 1] - starts[:, :n - 1]) / 100.0
    s24 = (s[:, g, 2::4][:, :n - 1] - starts[:, :n - 1]) / 100.0
    out["group%d_to_step16_us" % g] = [round(float(x), 2) if x > 0 else None for x in np.median(s16, axis=0)]
    out["group%d_to_step24_us" % g] = [round(float(x), 2) if x > 0 else None for x in np.median(s24, axis=0)]
    if os.environ.get("PIPE_TICKS"):                     # DVQ_ABLATE & 1024 build: cycle accumulators in slot 4p + 3
        v = s[:, g, 3::4][:, :n - 1]
        f = lambda sh: [int(x) for x in np.median((v >> sh) & ((1 << 21) - 1), axis=0)]
        out["group%d_cycles_barrier" % g] = f(0)          # memory phases: all 32 barriers; compute phases: barriers
        out["group%d_cycles_wait_a" % g] = f(21)          # memory phases: gather waits; compute phases: ring waits
        out["group%d_cycles_wait_b" % g] = f(42)          # memory phases: load waits
    out["group%d_total_us" % g] = round(float(np.median(starts[:, n - 1] - starts[:, 0])) / 100.0, 2)
out["kernel_span_us"] = round(float(s[:, :, 0::4].max() - s[:, :, 0][s[:, :, 0] > 0].min()) / 100.0, 2)
print(json.dumps(out))
