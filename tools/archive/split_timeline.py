"""Stage times of the split form of pass 1 (tuning build: DVQ_LIBRARY=.../libdvq_tuning.so): per workgroup eight 100-MHz wall-clock
stamps (0 start, 1 latents loaded and converted, 2 code loop + lane merge done, 3 slice results out and drained, 4 ticket known,
5 merged (last arriver only), 6 z_q / codes written, 7 end).  usage: python tools/split_timeline.py B H [W]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, quantize, synth

B, H = int(sys.argv[1]), int(sys.argv[2])
W = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else H
dev = torch.device("cuda:0")
K, D = 1024, 256
En = synth.codebook_trained(K, D)
cb = torch.from_numpy(En).to(dev)
z = torch.from_numpy(synth.z_tokens(En, B, H, W, 500 + B)).to(dev)
for a in sys.argv:
    if a.startswith("rs="):
        _lib.lib.dvq_tuning_set(b"res_slices", int(a[3:]))
prep = quantize._CodebookPrep()
out = quantize.vq_assign(z, cb, prep)
mode = _lib.MODE_FILTER_PASS1 if "pass1" in sys.argv else _lib.MODE_FILTER      # pass1: the kernel alone, back to back (warm instruction cache)
for _ in range(20):
    quantize.vq_assign(z, cb, prep, out=out, mode=mode)
nwg = 4096
st = torch.zeros(2 * nwg * 8, dtype=torch.int64, device=dev)
assert _lib.lib.dvq_tuning_buffers(st.data_ptr(), 0) == 0
quantize.vq_assign(z, cb, prep, out=out, mode=mode)
torch.cuda.synchronize()
_lib.lib.dvq_tuning_buffers(0, 0)
sall = st.cpu().numpy().reshape(2 * nwg, 8)
s = sall[:nwg]
used = s[:, 0] > 0
s = s[used].astype(np.float64)
t0 = s[:, 0].min()
print("workgroups", len(s))
us = (s - t0) / 100.0
us[s == 0] = np.nan
names = ["start", "prologue", "loop", "published", "ticket", "merged", "epilogue", "end"]
last = ~np.isnan(us[:, 5])
for nm, sel in (("all", np.ones(len(s), bool)), ("last arrivers", last)):
    print(nm, int(sel.sum()))
    for i, n in enumerate(names):
        col = us[sel, i]
        col = col[~np.isnan(col)]
        if len(col):
            print("  %-10s min %6.2f  median %6.2f  max %6.2f us" % (n, col.min(), np.median(col), col.max()))

# the resolver's workgroups (stamps 0 start, 1 shard count known, 2 records in LDS, 3 enumerated, 4 chains done, 5 rewritten, 6 end)
r = sall[nwg:].astype(np.float64)
act = r[:, 2] > 0
print("resolver workgroups", int((r[:, 0] > 0).sum()), "with records", int(act.sum()))
ru = (r - t0) / 100.0
ru[r == 0] = np.nan
for i, n in enumerate(["start", "count", "records", "enumerated", "chains", "rewritten", "end"]):
    col = ru[act, i]
    col = col[~np.isnan(col)]
    if len(col):
        print("  %-10s min %6.2f  median %6.2f  max %6.2f us" % (n, col.min(), np.median(col), col.max()))
idle = ru[(r[:, 0] > 0) & ~act]
if len(idle):
    print("  idle workgroups: start median %.2f, max %.2f" % (np.nanmedian(idle[:, 0]), np.nanmax(idle[:, 0])))
