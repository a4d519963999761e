"""Timing of the BASELINE.json configs other than the headline one (which bench.py owns): parity
for these is covered by tests/; this prints one JSON object with per-config times on 1 GPU."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import VectorQuantize2, VectorQuantizer2, _CodebookPrep, vq_assign
from dynamicvectorquantization_amd.router import (DualGrainFeatureRouter, TripleGrainFeatureRouter,
                                                  route_select_dual, route_select_triple)
from dynamicvectorquantization_amd.encode import encode_dual, encode_triple
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)

def timeit(fn, n=int(os.environ.get("DVQ_BC_ITERS", "20")), warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

out = {}
E = synth.codebook_trained(1024, 256)
# configs[0]: fixed-granularity VQ-VAE, 16x16x256, K=1024, B=4
vqg = VectorQuantizer2(1024, 256, beta=0.25, legacy=False).to(dev).eval()
z = t(synth.z_tokens(E, 4, 16, 16, 2001))
with torch.no_grad():
    out["cfg1_vqgan_B4_16x16_ms"] = timeit(lambda: vqg(z))
# configs[1]: dual feature router r=0.5, B=64
vq = VectorQuantize2(1024, 256).to(dev).eval(); vq.codebook.weight.data[:-1].copy_(t(E))
r = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
hf, hc = t(synth.z_tokens(E, 64, 32, 32, 2102)), t(synth.z_tokens(E, 64, 16, 16, 2112))
with torch.no_grad():
    out["cfg2_dual_feature_B64_ms"] = timeit(lambda: encode_dual(r, vq, hf, hc))
    out["cfg2_router_gate_fused_ms"] = timeit(lambda: r(h_fine=hf, h_coarse=hc))
out["cfg2_router_gate_torch_ops_ms"] = timeit(lambda: r(h_fine=hf, h_coarse=hc).detach())   # grad enabled -> torch ops
# configs[1] WITH the model's 1x1 quant_conv between select and quantizer (dqvae_dual_feat.py:66): the fused
# select + conv kernel (no h_dual) then the dense assign, vs route select + torch conv (MIOpen / hipBLASLt) + dense assign
qc = torch.nn.Conv2d(256, 256, 1).to(dev).eval()
with torch.no_grad():
    out["cfg2_dual_feature_B64_with_quant_conv_fused_ms"] = timeit(lambda: encode_dual(r, vq, hf, hc, quant_conv=qc))
    out["cfg2_dual_feature_B64_with_quant_conv_torch_ms"] = timeit(lambda: encode_dual(r, vq, hf, hc, quant_conv=torch.nn.Sequential(qc)))
    from dynamicvectorquantization_amd.qconv import quant_conv, quant_conv_select
    gate_ = r(h_fine=hf, h_coarse=hc)
    out["qconv_select_kernel_B64_ms"] = timeit(lambda: quant_conv_select(qc, hc, hf, gate=gate_))
    out["qconv_dense_kernel_B64_ms"] = timeit(lambda: quant_conv(qc, hf))
    out["torch_conv1x1_B64_ms"] = timeit(lambda: qc(hf))
# configs[3] per-rank share: triple F32/16/8, B=128 per GPU
r3 = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
hf, hm, hc = (t(synth.z_tokens(E, 128, 32, 32, 2104)), t(synth.z_tokens(E, 128, 16, 16, 2114)),
              t(synth.z_tokens(E, 128, 8, 8, 2124)))
with torch.no_grad():
    out["cfg4_triple_B128_per_gpu_ms"] = timeit(lambda: encode_triple(r3, vq, hf, hm, hc))
    out["cfg4_router_gate_fused_ms"] = timeit(lambda: r3(h_fine=hf, h_median=hm, h_coarse=hc))
out["cfg4_router_gate_torch_ops_ms"] = timeit(lambda: r3(h_fine=hf, h_median=hm, h_coarse=hc).detach())
# configs[4]: large-codebook stress K=16384, B=512 (exact fp32-MFMA path vs fp16 filter path)
E16 = synth.codebook_trained(16384, 256)
Et = t(E16)
zb = t(synth.z_tokens(E16, 64, 32, 32, 2005)).repeat(8, 1, 1, 1)      # configs[4]'s full B = 512 (8 x 64 distinct images)
pe, pf = _CodebookPrep(), _CodebookPrep()
te = timeit(lambda: vq_assign(zb, Et, pe, None, mode=_lib.MODE_EXACT), n=5, warm=2)
tf = timeit(lambda: vq_assign(zb, Et, pf, None, mode=_lib.MODE_FILTER), n=5, warm=2)
out["cfg5_K16384_B512_exact_ms"], out["cfg5_K16384_B512_filter_ms"] = te, tf
out["cfg5_filter_queue"] = pf.fallback_count()
zq0, c0, _ = vq_assign(zb, Et, pe, None, mode=_lib.MODE_EXACT)
zq1, c1, _ = vq_assign(zb, Et, pf, None, mode=_lib.MODE_FILTER)
out["cfg5_modes_bit_identical"] = bool(torch.equal(c0, c1) and torch.equal(zq0, zq1))
flops = 2.0 * 16384 * 256 * 512 * 1024
out["cfg5_exact_tflops"] = flops / (te * 1e-3) / 1e12
out["cfg5_filter_tflops_equiv"] = flops / (tf * 1e-3) / 1e12
# row f3: patch-entropy map of configs[2] (B=256 images 3x256x256), fused kernel vs the reference's op sequence as torch ops
from dynamicvectorquantization_amd.entropy import Entropy
img = t(synth.images_flat_noise(5000, 64)[0])
from oracle.entropy_torch import entropy_map as entropy_torch_ops
ef = Entropy(16, 256, 256).to(dev)
with torch.no_grad():
    out["entropy_map_B64_fused_ms"] = timeit(lambda: ef(img), n=10, warm=3)
    out["entropy_map_B64_torch_ops_ms"] = timeit(lambda: entropy_torch_ops(img), n=5, warm=2)
# row f1: permuter forward on configs[2]-shaped codes (B=256)
from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
perm = DualGrainSeperatePermuter(coarse_hw=16, fine_hw=32, content_pad_code=1024, content_eos_code=1025,
                                 coarse_position_pad_code=256, coarse_position_eos_code=257,
                                 fine_position_pad_code=1024, fine_position_eos_code=1025, fine_position_order="region-first").to(dev) if True else None
codes = torch.randint(0, 1024, (256, 16, 16), device=dev).repeat_interleave(2, 1).repeat_interleave(2, 2)
grainm = torch.from_numpy(synth.grain_gate_dual(4002, 256, 16, 16).argmax(-1)).to(dev)
try:
    out["permuter_forward_B256_ms"] = timeit(lambda: perm(codes, grainm), n=10, warm=3)
except Exception as ex:
    out["permuter_forward_B256_ms"] = "n/a: %s" % ex
print(json.dumps(out, indent=1))
