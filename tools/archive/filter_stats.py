import sys, torch, numpy as np
sys.path.insert(0, '.')
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device('cuda:0')
for kind, K, B in (("trained", 1024, 64), ("default", 1024, 8), ("trained", 16384, 16)):
    E = synth.codebook_trained(K, 256) if kind == "trained" else synth.codebook_default_init(K, 256)
    z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 2003)).to(dev)
    Et = torch.from_numpy(E).to(dev)
    pf, pe = _CodebookPrep(), _CodebookPrep()
    zq1, c1, l1 = vq_assign(z, Et, pf, None, mode=_lib.MODE_FILTER)
    zq0, c0, l0 = vq_assign(z, Et, pe, None, mode=_lib.MODE_EXACT)
    torch.cuda.synchronize()
    print(kind, K, "queued,exact-list", pf.fallback_count(), "of", B * 1024, "codes equal", bool(torch.equal(c0, c1)),
          "zq equal", bool(torch.equal(zq0, zq1)), "loss", l0.tolist(), l1.tolist())
    for mode, p in ((0, pe), (1, pf)):
        for _ in range(3): vq_assign(z, Et, p, None, mode=mode)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): vq_assign(z, Et, p, None, mode=mode)
        e.record(); torch.cuda.synchronize()
        print("   mode", mode, "ms/call", s.elapsed_time(e) / 10)
