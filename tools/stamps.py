"""Diagnostic: per-workgroup phase timeline of the register-resident filter kernel (DVQ_DEBUG_STAMPS=1)."""
import os, sys, ctypes
os.environ["DVQ_DEBUG_STAMPS"] = "1"
os.environ.setdefault("DVQ_TUNE_VARIANT", "1")
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device('cuda:0')
B = 256
E = synth.codebook_trained(1024, 256)
z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 2003)).to(dev)
Et = torch.from_numpy(E).to(dev)
p = _CodebookPrep()
for _ in range(3): vq_assign(z, Et, p, None, mode=1)
torch.cuda.synchronize()
nb = B * 1024 // 128
buf = (ctypes.c_ulonglong * (8 * nb))()
rc = ctypes.CDLL(_lib.LIB_PATH).dvq_debug_read_stamps(buf, nb)
raw = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64)
seg = raw[:, 4:] / 32.0   # shader cycles per code-tile step (32 steps)
print('per step cycles (wave 0 of each WG): wait+barrier %.0f | issue+seeds+lgkm %.0f | mfma chain %.0f | valu epilogue %.0f' % tuple(seg.mean(0)))
a = raw[:, :4] / 100.0   # us
t0 = a[:, 0].min()
a -= t0
print("rc", rc, "kernel span %.1f us" % a[:, 3].max())
print("prologue  mean %.1f  p50 %.1f  p90 %.1f" % ((a[:,1]-a[:,0]).mean(), np.median(a[:,1]-a[:,0]), np.percentile(a[:,1]-a[:,0], 90)))
print("code loop mean %.1f  p50 %.1f  p90 %.1f" % ((a[:,2]-a[:,1]).mean(), np.median(a[:,2]-a[:,1]), np.percentile(a[:,2]-a[:,1], 90)))
print("epilogue  mean %.1f  p50 %.1f  p90 %.1f" % ((a[:,3]-a[:,2]).mean(), np.median(a[:,3]-a[:,2]), np.percentile(a[:,3]-a[:,2], 90)))
order = np.argsort(a[:, 0])
for q in (0, 255, 511, 512, 767, 1023, 1024, 1535, 2047):
    i = order[q]
    print("wg#%4d start %.1f  prolog-end %.1f  loop-end %.1f  end %.1f" % (q, a[i,0], a[i,1], a[i,2], a[i,3]))
