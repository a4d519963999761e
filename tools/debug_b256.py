import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device('cuda:0')
B = 256
E = synth.codebook_trained(1024, 256)
z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 2003)).to(dev)
mask = torch.from_numpy(np.where(synth.bernoulli(2004, (B, 1, 32, 32), 0.5), 1.0, 0.25).astype(np.float32)).to(dev)
Et = torch.from_numpy(E).to(dev)
pe, pf = _CodebookPrep(), _CodebookPrep()
zq0, c0, l0 = vq_assign(z, Et, pe, mask, mode=0)
for it in range(3):
    zq1, c1, l1 = vq_assign(z, Et, pf, mask, mode=1)
    torch.cuda.synchronize()
    dc = (c0 != c1).nonzero()
    dz = (zq0 != zq1).any(dim=1).nonzero()
    print("iter", it, "queued", pf.fallback_count(), "code diffs", len(dc), "zq token diffs", len(dz), "loss", l0.tolist(), l1.tolist())
    if len(dc):
        print(dc[:5].tolist(), c0[tuple(dc[0])].item(), c1[tuple(dc[0])].item())
    if len(dz):
        print("zq diff tokens", dz[:5].tolist())
b, y, x = 212, 29, 8
d = (zq0[b, :, y, x] != zq1[b, :, y, x]).nonzero().flatten()
print("channels differing", d.tolist()[:40], len(d))
code = c0[b, y, x].item()
e = Et[code]
zz = z[b, :, y, x]
ref = zz + (e - zz)
print("mode0 == ref", torch.equal(zq0[b, :, y, x], ref), "mode1 == ref", torch.equal(zq1[b, :, y, x], ref))
k = d[0].item()
print("k", k, "z", zz[k].item(), "e", e[k].item(), "zq0", zq0[b, k, y, x].item(), "zq1", zq1[b, k, y, x].item())
# which code gives zq1?
for cand in range(1024):
    if torch.equal(zq1[b, :, y, x], zz + (Et[cand] - zz)):
        print("zq1 matches code", cand)
