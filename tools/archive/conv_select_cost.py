"""What can a staged select (SEL = 2) still save in the CONV form of pass 1?  Upper bound: the CONV kernel with NO select at all (dense input =
the materialised h_dual, vq_assign_filter_kernel<256, 0, true>) against the shipped one with the per-lane select (<256, 1, true>), same tokens,
B = 256, pass 1 only (DVQ_MODE_FILTER_PASS1), HIP events over 200 launches, median of 5.   usage: python tools/conv_select_cost.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.router import route_select_dual_entropy
dev = torch.device("cuda:0")
THR = 1.6777750253677368
B, K, D = 256, 1024, 256
E = synth.codebook_trained(K, D)
Et = torch.from_numpy(E).to(dev)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf = tile(torch.from_numpy(synth.z_tokens(E, b0, 32, 32, 2903)).to(dev))
hc = tile(torch.from_numpy(synth.z_tokens(E, b0, 16, 16, 2913)).to(dev))
ent = tile(torch.from_numpy(synth.entropy_map(5903, b0, 16, 16)).to(dev))
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
with torch.no_grad():
    q, _ = torch.linalg.qr(torch.from_numpy(synth.normal(2911, (D, D))).double())
    conv.weight.copy_(q.float().reshape(D, D, 1, 1).to(dev)); conv.bias.copy_(torch.from_numpy(synth.normal(2912, (D,), 0.0, 0.1)).to(dev))
sel = route_select_dual_entropy(ent, THR, hc, hf)
hd, cm = sel["h_dual"].contiguous(), sel["codebook_mask"].contiguous()


def timeit(fn):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 200 * 1e3)
    return round(sorted(ts)[2], 2)


res = {}
with torch.no_grad():
    p1, p2, p3, p4 = _CodebookPrep(), _CodebookPrep(), _CodebookPrep(), _CodebookPrep()
    r = vq_assign_routed_dual(hc, hf, Et, p1, entropy=ent, threshold=THR, conv=conv, mode=_lib.MODE_FILTER_PASS1)
    o1 = (r["zq"], r["codes"], r["loss"], r["indices"], r["codebook_mask"], r["gate"])
    zq, codes, loss = vq_assign(hd, Et, p2, cm, conv=conv, mode=_lib.MODE_FILTER_PASS1)
    o2 = (zq, codes, loss)
    r3 = vq_assign_routed_dual(hc, hf, Et, p3, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1)
    o3 = (r3["zq"], r3["codes"], r3["loss"], r3["indices"], r3["codebook_mask"], r3["gate"])
    zq4, codes4, loss4 = vq_assign(hd, Et, p4, cm, mode=_lib.MODE_FILTER_PASS1)
    o4 = (zq4, codes4, loss4)
    for rep in range(2):
        res.setdefault("conv_per_lane_select_SEL1_us", []).append(timeit(lambda: vq_assign_routed_dual(hc, hf, Et, p1, entropy=ent, threshold=THR, conv=conv, mode=_lib.MODE_FILTER_PASS1, out=o1)))
        res.setdefault("conv_no_select_dense_input_SEL0_us", []).append(timeit(lambda: vq_assign(hd, Et, p2, cm, conv=conv, mode=_lib.MODE_FILTER_PASS1, out=o2)))
        res.setdefault("noconv_staged_select_SEL2_us", []).append(timeit(lambda: vq_assign_routed_dual(hc, hf, Et, p3, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1, out=o3)))
        res.setdefault("noconv_no_select_dense_SEL0_us", []).append(timeit(lambda: vq_assign(hd, Et, p4, cm, mode=_lib.MODE_FILTER_PASS1, out=o4)))
print(json.dumps(res))
