"""Stage stamps of the resolver's workgroups at the headline size (dense op, B = 256, 32 x 32, K = 1024; tuning build):
how many chunks have records, how long each takes, how they spread over time.  usage: python tools/resolver_timeline.py [B]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib, quantize, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
K, D = 1024, 256
En = synth.codebook_trained(K, D)
cb = torch.from_numpy(En).to(dev)
z = torch.from_numpy(synth.z_tokens(En, min(B, 32), 32, 32, 77)).to(dev)
z = torch.cat([z] * max(1, B // z.shape[0]), 0).contiguous()
prep = quantize._CodebookPrep()
out = quantize.vq_assign(z, cb, prep)
for _ in range(10):
    quantize.vq_assign(z, cb, prep, out=out)
nwg = 4096
st = torch.zeros(2 * nwg * 8, dtype=torch.int64, device=dev)
assert _lib.lib.dvq_tuning_buffers(st.data_ptr(), 0) == 0
quantize.vq_assign(z, cb, prep, out=out)
torch.cuda.synchronize()
_lib.lib.dvq_tuning_buffers(0, 0)
print("queued / listed", prep.fallback_count())
r = st.cpu().numpy().reshape(2 * nwg, 8)[nwg:].astype(np.float64)
launched = r[:, 0] > 0
act = r[:, 2] > 0
t0 = r[launched, 0].min()
u = (r - t0) / 100.0
u[r == 0] = np.nan
print("resolver workgroups launched", int(launched.sum()), "with records", int(act.sum()))
names = ["start", "count", "records", "enumerated", "chains", "rewritten", "end"]
for i, n in enumerate(names):
    col = u[act, i]; col = col[~np.isnan(col)]
    print("  %-10s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (n, col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
dur = u[act, 6] - u[act, 0]
enum = u[act, 3] - u[act, 2]
print("  chunk duration: min %.2f median %.2f p90 %.2f max %.2f;  enumeration: median %.2f p90 %.2f max %.2f" %
      (np.nanmin(dur), np.nanmedian(dur), np.nanpercentile(dur, 90), np.nanmax(dur), np.nanmedian(enum), np.nanpercentile(enum, 90), np.nanmax(enum)))
hist, edges = np.histogram(dur[~np.isnan(dur)], bins=8)
print("  duration histogram:", [(round(float(edges[i]), 1), int(hist[i])) for i in range(len(hist))])
idle = u[launched & ~act]
print("  workgroups without records: last start %.2f, last end %.2f" % (np.nanmax(idle[:, 0]), np.nanmax(idle[:, 0])))
