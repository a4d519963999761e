import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dynamicvectorquantization_amd import _lib, quantize, synth
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from small_probe import t_op
dev = torch.device("cuda:0")
K, D = 1024, 256
En = synth.codebook_trained(K, D)
cb = torch.from_numpy(En).to(dev)
for (B, H) in ((4, 16), (16, 16), (16, 32)):
    z = torch.from_numpy(synth.z_tokens(En, B, H, H, 500 + B)).to(dev)
    ref = None
    for rs in (1, 2, 4, 8):
        _lib.lib.dvq_tuning_set(b"res_slices", rs)
        prep = quantize._CodebookPrep()
        out = quantize.vq_assign(z, cb, prep)
        out = tuple(o.clone() for o in out)
        t = t_op(lambda: quantize.vq_assign(z, cb, prep, out=out))
        if ref is None: ref = tuple(o.clone() for o in out)
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        print(B, H, "res_slices", rs, "%.2f us" % t, same, flush=True)
