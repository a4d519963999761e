"""Stage timeline of pass 1's workgroups from in-kernel stamps (100-MHz wall clock; a tuning build with the stamps of ALL forms
enabled: DVQ_LIBRARY=tools/tmpv/libdvq_tuning_st.so): dense pass 1 against the routed (select-fused) pass 1 at B = 256.
Stamps: 0 kernel entry, 1 prologue done (latents in registers, first code tiles issued), 2 code loop + merge done, 6 z_q / loss
phase done, 7 exit."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
stamps = torch.zeros((8192, 8), dtype=torch.int64, device=dev)
pd, pr = _CodebookPrep(), _CodebookPrep()
forms = {"dense": lambda: vq_assign(hf, E, pd, cmask, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)),
         "routed": lambda: vq_assign_routed_dual(hc, hf, E, pr, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1,
                                                 out=(zq, codes, None, grain, cmask, gate))}
out = {}
for name, fn in forms.items():
    for _ in range(200): fn()
    torch.cuda.synchronize()
    rows = []
    for rep in range(5):
        stamps.zero_()
        _lib.check(_lib.lib.dvq_tuning_buffers(stamps.data_ptr(), None), "buffers")
        for _ in range(3): fn()                      # the last launch's stamps stay
        torch.cuda.synchronize()
        _lib.check(_lib.lib.dvq_tuning_buffers(None, None), "buffers")
        s = stamps[:2048].cpu().numpy().astype(np.float64) * 0.01          # us
        t0 = s[:, 0].min()
        start = s[:, 0] - t0
        order = np.argsort(start)
        q = np.array_split(order, 4)                 # quarters by start time ~ generations
        rows.append({"kernel_us": float(s[:, 7].max() - t0),
                     "per_quarter": [{"start_med": float(np.median(start[i])), "prologue": float(np.median(s[i, 1] - s[i, 0])),
                                      "loop": float(np.median(s[i, 2] - s[i, 1])), "zq_phase": float(np.median(s[i, 6] - s[i, 2])),
                                      "exit": float(np.median(s[i, 7] - s[i, 6])), "total": float(np.median(s[i, 7] - s[i, 0]))} for i in q]})
    out[name] = rows[2:]                              # (the first repetitions re-warm)
print(json.dumps(out, indent=1))
