import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device("cuda:0")
for K in (1024, 4096, 16384):
    E = synth.codebook_trained(K, 256)
    Et = torch.from_numpy(E).to(dev)
    z = torch.from_numpy(synth.z_tokens(E, int(os.environ.get("KS_B", "64")), 32, 32, 2005)).to(dev)
    p = _CodebookPrep()
    for _ in range(3): vq_assign(z, Et, p, None, mode=_lib.MODE_FILTER_PASS1, want_loss=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): vq_assign(z, Et, p, None, mode=_lib.MODE_FILTER_PASS1, want_loss=False)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("K=%d pass-1 %.3f ms  per-tile %.3f us" % (K, ms, ms * 1e3 / (K / 32)))
