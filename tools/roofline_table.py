"""One roofline line per kernel of the path at BASELINE sizes (algorithmic bytes / flops per
launch divided by the launch's duration, against the MI355X peaks: HBM 8 TB/s, fp16 MFMA 2500 TFLOP/s dense, fp32 MFMA
157.3 TFLOP/s).  Durations are HIP-event means over back-to-back launches of the ABI op that consists of (almost) only
that kernel; multi-kernel ops are listed with what the bracket contains.  Prints one JSON object."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _pass1_ms          # pass 1 ALONE: counters zeroed in front of the bracket, workspace declared clean
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.router import (DualGrainFeatureRouter, TripleGrainFeatureRouter, route_select_dual_entropy,
                                                  route_select_triple)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
HBM, F16, F32 = 8000.0, 2500.0, 157.3      # GB/s, TFLOP/s, TFLOP/s
THR = 1.6777750253677368


def timeit(fn, n=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3       # seconds


rows = []
def row(kernel, what, seconds, nbytes=None, flops=None, peak_tf=None, note=""):
    r = {"kernel": kernel, "workload": what, "us": round(seconds * 1e6, 1)}
    if nbytes is not None:
        r["algorithmic_MB"] = round(nbytes / 1e6, 1)
        r["GB_per_s"] = round(nbytes / seconds / 1e9, 1)
        r["hbm_frac"] = round(nbytes / seconds / 1e9 / HBM, 3)
    if flops is not None:
        r["algorithmic_GFLOP"] = round(flops / 1e9, 1)
        r["TFLOP_per_s"] = round(flops / seconds / 1e12, 1)
        r["mfma_frac"] = round(flops / seconds / 1e12 / peak_tf, 3)
        r["mfma_peak"] = peak_tf
    if note:
        r["note"] = note
    rows.append(r)


B, K, D = 256, 1024, 256
En = synth.codebook_trained(K, D)
b0 = 32
tile = lambda x, BB=B: torch.cat([torch.roll(x, 5 * k, -1) for k in range((BB + b0 - 1) // b0)], 0)[:BB].contiguous()
hf, hc, ent, E = (tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))),
                  tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En))
N = B * 1024
vq_bytes = N * (D * 4 * 2 + 8 + 4) + K * D * 4
vq_flops = 2.0 * K * D * N
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev); h_dual = torch.empty_like(hf)
prep = _CodebookPrep()
for _ in range(300):                          # bring the part out of its idle power state before the first measurement
    vq_assign(hf, E, prep, cmask, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None))
torch.cuda.synchronize()
pd = _CodebookPrep()
s = 1e-3 * _pass1_ms(lambda: vq_assign_routed_dual(hc, hf, E, pd, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1,
                                                   out=(zq, codes, None, grain, cmask, gate)), pd, n=100)
row("vq_assign_filter_kernel<256,2,false,false> (pass 1, select fused in, coarse branch through LDS)", "configs[2] B=256 K=1024", s, vq_bytes, vq_flops, F16,
    "the kernel alone (HIP events; workspace declared clean); VQ-forward byte count (the kernel also does the select's work); issue-bound code loop between two HBM-bound phases, at the socket's power cap (DESIGN 5.1)")
from dynamicvectorquantization_amd import qconv as _qc
_q, _ = np.linalg.qr(synth.normal(6012, (D, D), 0.0, 1.0).astype(np.float64))
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
with torch.no_grad():
    conv.weight.copy_(t(_q.astype(np.float32).reshape(D, D, 1, 1))); conv.bias.copy_(t(synth.normal(6013, (D,), 0.0, 0.1)))
pc = _CodebookPrep()
s = 1e-3 * _pass1_ms(lambda: vq_assign_routed_dual(hc, hf, E, pc, entropy=ent, threshold=THR, mode=_lib.MODE_FILTER_PASS1, conv=conv,
                                                   out=(zq, codes, None, grain, cmask, gate)), pc, n=100)
row("vq_assign_filter_kernel<256,1,true,false> (pass 1 with the select AND the 1x1 quant_conv fused in: the model order)", "configs[2] B=256 K=1024", s, vq_bytes,
    vq_flops + 3 * 2.0 * D * D * N, F16, "flops incl. the conv's three split-fp16 terms")
pdn = _CodebookPrep()
s = 1e-3 * _pass1_ms(lambda: vq_assign(hf, E, pdn, cmask, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)), pdn, n=100)
row("vq_assign_filter_kernel<256,0,false,false> (dense pass 1)", "B=256 K=1024", s, vq_bytes, vq_flops, F16, "the kernel alone")
s_full = timeit(lambda: vq_assign(hf, E, prep, cmask, mode=_lib.MODE_FILTER, out=(zq, codes, loss)))
row("dense filter op (pass 1 + resolver + list/finalize; no zero kernel in the steady state)", "B=256 K=1024", s_full, vq_bytes, vq_flops, F16)
s = timeit(lambda: vq_assign(hf, E, prep, cmask, mode=_lib.MODE_EXACT, out=(zq, codes, loss)), n=10, warm=3)
row("vq_assign_exact_kernel<256> (fp32 MFMA chain)", "B=256 K=1024", s, vq_bytes, vq_flops, F32)
s = timeit(lambda: route_select_dual_entropy(ent, THR, hc, hf, out=(h_dual, grain, cmask, gate)))
row("route_select_kernel (dual, entropy gate fused)", "B=256", s, N * (D * 4 * 2 + 4) + B * 256 * (4 + 8 + 16))
# large codebook
E16n = synth.codebook_trained(16384, 256)
E16 = t(E16n)
zb = t(synth.z_tokens(E16n, 64, 32, 32, 2005)).repeat(8, 1, 1, 1)
p16 = _CodebookPrep()
s = timeit(lambda: vq_assign(zb, E16, p16, None, mode=_lib.MODE_FILTER_PASS1), n=5, warm=2)
row("vq_assign_filter_wide_kernel<256> (pass 1, K=16384)", "configs[4] B=512", s, 512 * 1024 * 2060 + 16384 * 1024, 2.0 * 16384 * 256 * 512 * 1024, F16,
    "matrix-bound config")
del zb, E16, p16
# feature-router gate, triple B=128 and dual B=64: rocprof splits pool / MLP (profiles/archive/r02_gate_kernels_*.txt); here the op
r3 = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
f3 = [t(synth.z_tokens(En, 128, 8 << i, 8 << i, 2124 - 10 * i)) for i in range(3)]
with torch.no_grad():
    s = timeit(lambda: r3(h_fine=f3[2], h_median=f3[1], h_coarse=f3[0]))
feat_bytes3 = sum(x.numel() * 4 for x in f3)
row("gate_pool_kernel + gate_gemm_kernel<3,48> + gate_finalize_kernel (triple gate op)", "configs[3] per-rank B=128", s, feat_bytes3,
    3 * 2.0 * 768 * 768 * 128 * 64, F16, "bytes = the branch features once; flops = the 3-term split hidden layer; per kernel: tools/gate_trace.sh")
r2 = DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").to(dev).eval()
f2 = [t(synth.z_tokens(En, 64, 16 << i, 16 << i, 2112 - 10 * i)) for i in range(2)]
with torch.no_grad():
    s = timeit(lambda: r2(h_fine=f2[1], h_coarse=f2[0]))
row("gate_pool_kernel + gate_gemm_kernel<2,32> + gate_finalize_kernel (dual gate op)", "configs[1] B=64", s, sum(x.numel() * 4 for x in f2),
    3 * 2.0 * 512 * 512 * 64 * 256, F16)
# quant_conv
from dynamicvectorquantization_amd.qconv import quant_conv, quant_conv_select
qc = torch.nn.Conv2d(256, 256, 1).to(dev).eval()
with torch.no_grad():
    g2 = r2(h_fine=f2[1], h_coarse=f2[0])
    s = timeit(lambda: quant_conv_select(qc, f2[0], f2[1], gate=g2))
npos = 64 * 1024
row("qconv_kernel<256,SEL> (select + 1x1 conv, split-fp16 MFMA)", "configs[1] B=64", s, npos * D * 4 * 2, 3 * 2.0 * 256 * 256 * npos, F16)
# entropy map
from dynamicvectorquantization_amd.entropy import Entropy
ef = Entropy(16, 256, 256).to(dev)
_ib = t(synth.images_flat_noise(5000, 32)[0])
for BB in (64, 256):
    img = torch.cat([torch.roll(_ib, 16 * k, -1) for k in range(BB // 32)], 0).contiguous()
    with torch.no_grad():
        s = timeit(lambda: ef(img), n=200, warm=200)
    row("entropy_map_kernel", "B=%d images 3x256x256" % BB, s, img.numel() * 4 + BB * 256 * 4,
        note="vector-instruction issue + LDS histogram round trips beside the image read (csrc/entropy_map.hip)")
del img
# EMA statistics
cs, vs = torch.zeros(K, device=dev), torch.zeros(K, D, device=dev)
cod = torch.randint(0, K, (B, 32, 32), device=dev)
def ema():
    _lib.check(_lib.lib.dvq_ema_accumulate_nchw_f32(hf.data_ptr(), cod.data_ptr(), B, D, 1024, K, cs.data_ptr(), vs.data_ptr(), _lib.stream_ptr(dev)), "ema")
s = timeit(ema, n=20, warm=5)
row("ema_accumulate_kernel", "B=256 K=1024 (uniform random codes)", s, N * (D * 4 + 8), note="float atomics into K*D sums: atomic throughput, not HBM")
# the same with the codes a dual-grain batch produces: half of the 2 x 2 cells carry one code (combined in LDS before the atomics)
_fine = torch.randint(0, K, (B, 32, 32), device=dev)
_up = lambda x: x.repeat_interleave(2, 1).repeat_interleave(2, 2)
cod = torch.where(_up(torch.rand((B, 16, 16), device=dev) < 0.5), _up(torch.randint(0, K, (B, 16, 16), device=dev)), _fine).contiguous()
s = timeit(ema, n=20, warm=5)
row("ema_accumulate_kernel", "B=256 K=1024 (dual-grain codes: half of the 2x2 cells coarse)", s, N * (D * 4 + 8), note="equal codes inside a 64-token tile are summed in LDS first: 0.625 of the atomics")
print(json.dumps({"peaks": {"hbm_GB_per_s": HBM, "fp16_mfma_TFLOP_per_s": F16, "fp32_mfma_TFLOP_per_s": F32}, "rows": rows}, indent=1))
