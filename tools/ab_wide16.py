"""configs[4] (K = 16384, B = 512): wide pass-1 kernel with the 32x32x16 vs the 16x16x32 code loop; outputs compared."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
out = {}
for K in (16384, 2048):
    E = synth.codebook_trained(K, 256)
    zb = t(synth.z_tokens(E, 64, 32, 32, 2005)).repeat(8, 1, 1, 1)
    Et = t(E)
    def timeit(fn, n=6, warm=2):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return round(s.elapsed_time(e) / n, 3)
    ref = None
    for rep in range(2):
        for m16 in ("0", "1"):
            os.environ["DVQ_MFMA16"] = m16
            p = _CodebookPrep()
            ms = timeit(lambda: vq_assign(zb, Et, p, None, mode=_lib.MODE_FILTER))
            zq, c, l = vq_assign(zb, Et, p, None, mode=_lib.MODE_FILTER); torch.cuda.synchronize()
            if ref is None: ref = (zq.clone(), c.clone())
            out.setdefault("K%d_B512_mfma16=%s" % (K, m16), []).append({"ms": ms, "same": bool(torch.equal(zq, ref[0]) and torch.equal(c, ref[1])), "queue": p.fallback_count(),
                                                                      "tflops_equiv": round(2.0 * K * 256 * 512 * 1024 / (ms * 1e-3) / 1e12, 1)})
os.environ.pop("DVQ_MFMA16", None)
print(json.dumps(out))
