"""The 1x1 quant_conv fused into pass 1 (dvq_vq_assign_qconv_f32 / dvq_vq_assign_routed_qconv_dual_f32) against its parts:
h (with h_buf given the op writes the conv output of every token) vs an fp64 conv within 1e-5 * sum|w||x|, codes / z_q / loss
bit-identical to the dense assign run on that h, by-products equal to the unfused routed op's; then timings at B = 256."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.qconv import quant_conv, quant_conv_select
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K, D = 1024, 256
E = t(synth.codebook_trained(K, D))
torch.manual_seed(3)
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
with torch.no_grad():
    conv.weight.mul_(3.0); conv.bias.mul_(2.0)
out = {}

def check_h(h, x64, name):
    W = conv.weight.detach().double().reshape(D, D); b = conv.bias.detach().double()
    ref = torch.einsum("oc,bcn->bon", W, x64) + b[None, :, None]
    bound = torch.einsum("oc,bcn->bon", W.abs(), x64.abs()) + b.abs()[None, :, None]
    r = float(((h.double().reshape(ref.shape) - ref).abs() / bound).max())
    out[name + "_h_err_over_sum_abs"] = r
    assert r < 1e-5, r

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for B in (8, 64):
    En = E.cpu().numpy()
    x = t(synth.z_tokens(En, B, 32, 32, 7000 + B))
    x[0, :, 0, 0] *= 1e-3; x[1, 5, 3, 3] = 40.0; x[2, :16, 1, 1] = 1e-6      # scale changes between k-steps
    mask = t(np.where(synth.bernoulli(7100 + B, (B, 1, 32, 32), 0.5), 1.0, 0.25).astype(np.float32))
    hb = torch.empty_like(x)
    zq, codes, loss = vq_assign(x, E, _CodebookPrep(), mask, conv=conv, h_buf=hb)
    torch.cuda.synchronize()
    check_h(hb, x.double().reshape(B, D, -1), "dense_B%d" % B)
    zq2, codes2, loss2 = vq_assign(hb, E, _CodebookPrep(), mask)
    assert torch.equal(codes, codes2), int((codes != codes2).sum())
    assert torch.equal(zq, zq2) and torch.equal(loss, loss2), (loss, loss2)
    hq = quant_conv(conv, x)
    out["dense_B%d_max_abs_diff_vs_qconv_kernel" % B] = float((hq - hb).abs().max())
    # without h_buf (production): same outputs
    zq3, codes3, loss3 = vq_assign(x, E, _CodebookPrep(), mask, conv=conv)
    assert torch.equal(codes, codes3) and torch.equal(zq, zq3) and torch.equal(loss, loss3)
    # special values: NaN / Inf inputs go through the exact list with the spilled h rows
    xs = x.clone(); xs[3, 7, 2, 2] = float("nan"); xs[4, :, 5, 5] = float("inf"); xs[5, :, 6, 6] *= 1e30
    hb2 = torch.empty_like(x)
    zq4, codes4, loss4 = vq_assign(xs, E, _CodebookPrep(), mask, conv=conv, h_buf=hb2)
    zq5, codes5, loss5 = vq_assign(hb2, E, _CodebookPrep(), mask)
    same = (zq4 == zq5) | (torch.isnan(zq4) & torch.isnan(zq5))
    assert torch.equal(codes4, codes5) and bool(same.all())
    zq6, codes6, _ = vq_assign(xs, E, _CodebookPrep(), mask, conv=conv)
    same = (zq4 == zq6) | (torch.isnan(zq4) & torch.isnan(zq6))
    assert torch.equal(codes4, codes6) and bool(same.all())
    # routed dual, entropy gate
    hf = x; hc = t(synth.z_tokens(En, B, 16, 16, 7200 + B)); ent = t(synth.entropy_map(7300 + B, B, 16, 16))
    thr = 1.6777750253677368
    hb3 = torch.empty_like(hf)
    r = vq_assign_routed_dual(hc, hf, E, _CodebookPrep(), entropy=ent, threshold=thr, conv=conv, h_buf=hb3)
    u = quant_conv_select(conv, hc, hf, entropy=ent, threshold=thr)
    assert torch.equal(r["indices"], u["indices"]) and torch.equal(r["codebook_mask"], u["codebook_mask"]) and torch.equal(r["gate"], u["gate"])
    sel = torch.where(u["indices"].repeat_interleave(2, 1).repeat_interleave(2, 2)[:, None] == 1, hf, hc.repeat_interleave(2, 2).repeat_interleave(2, 3))
    check_h(hb3, sel.double().reshape(B, D, -1), "routed_B%d" % B)
    zq7, codes7, loss7 = vq_assign(hb3, E, _CodebookPrep(), r["codebook_mask"])
    assert torch.equal(r["codes"], codes7) and torch.equal(r["zq"], zq7) and torch.equal(r["loss"], loss7)
    out["routed_B%d_max_abs_diff_vs_qconv_kernel" % B] = float((u["h"] - hb3).abs().max())
out["parity"] = "ok"
# timings at B = 256
B = 256
En = E.cpu().numpy()
b0 = 32
tile = lambda a: torch.cat([torch.roll(a, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))); hc = tile(t(synth.z_tokens(En, b0, 16, 16, 2913))); ent = tile(t(synth.entropy_map(5903, b0, 16, 16)))
prep = _CodebookPrep()
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev); h = torch.empty_like(hf)
out["us_dense_fused"] = timeit(lambda: vq_assign(hf, E, prep, None, out=(zq, codes, loss), conv=conv))
out["us_dense_conv_then_assign"] = timeit(lambda: vq_assign(quant_conv(conv, hf), E, prep, None, out=(zq, codes, loss)))
out["us_dense_assign_alone"] = timeit(lambda: vq_assign(hf, E, prep, None, out=(zq, codes, loss)))
out["us_routed_fused"] = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=thr, out=(zq, codes, loss, grain, cmask, gate), conv=conv))
def two():
    u = quant_conv_select(conv, hc, hf, entropy=ent, threshold=thr, out=(h, grain, cmask, gate))
    vq_assign(h, E, prep, cmask, out=(zq, codes, loss))
out["us_routed_conv_select_then_assign"] = timeit(two)
out["us_routed_assign_alone_no_conv"] = timeit(lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=thr, out=(zq, codes, loss, grain, cmask, gate)))
out["us_dense_fused_pass1_only"] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None), conv=conv))
out["us_dense_pass1_only"] = timeit(lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))
print(json.dumps(out))
