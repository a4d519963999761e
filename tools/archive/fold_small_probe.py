import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from small_probe import t_op
dev = torch.device("cuda:0")
K, D = 1024, 256
En = synth.codebook_trained(K, D); cb = torch.from_numpy(En).to(dev)
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
has = hasattr(_lib.lib, "dvq_tuning_set")
for B in (1, 4, 8):
    hf = torch.from_numpy(synth.z_tokens(En, B, 32, 32, 2903)).to(dev); hc = torch.from_numpy(synth.z_tokens(En, B, 16, 16, 2913)).to(dev)
    ent = torch.from_numpy(synth.entropy_map(5903, B, 16, 16)).to(dev)
    row = {}
    for sp in ((0, 1) if has else (1,)):
        if has: _lib.lib.dvq_tuning_set(b"split", sp)
        prep = _CodebookPrep()
        with torch.no_grad():
            f = lambda: vq_assign_routed_dual(hc, hf, cb, prep, entropy=ent, threshold=1.6777750253677368, conv=conv, fold=True, want_loss=False)
            f()
            row["split%d" % sp] = round(t_op(f, 300), 1)
    print(B, row, flush=True)
