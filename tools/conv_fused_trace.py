"""workload for rocprofv3: the fused conv + assign op (dense and routed) and its two-kernel counterpart, B = 256"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.qconv import quant_conv, quant_conv_select
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K, D, B = 1024, 256, 256
En = synth.codebook_trained(K, D); E = t(En)
conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
with torch.no_grad():
    conv.weight.copy_(t(synth.normal(6012, (D, D, 1, 1), 0.0, 1.0 / 16.0))); conv.bias.copy_(t(synth.normal(6013, (D,), 0.0, 0.1)))
b0 = 32
tile = lambda a: torch.cat([torch.roll(a, 5 * k, -1) for k in range(B // b0)], 0).contiguous()
hf = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))); hc = tile(t(synth.z_tokens(En, b0, 16, 16, 2913))); ent = tile(t(synth.entropy_map(5903, b0, 16, 16)))
prep = _CodebookPrep(); thr = 1.6777750253677368
zq = torch.empty_like(hf); codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev); loss = torch.empty(2, device=dev)
grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev); cmask = torch.empty((B, 1, 32, 32), device=dev)
gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev); h = torch.empty_like(hf)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for _ in range(40):
    if which in ("all", "fused"):
        vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=thr, out=(zq, codes, loss, grain, cmask, gate), conv=conv)
    if which in ("all", "two"):
        quant_conv_select(conv, hc, hf, entropy=ent, threshold=thr, out=(h, grain, cmask, gate))
        vq_assign(h, E, prep, cmask, out=(zq, codes, loss))
torch.cuda.synchronize()
print("fallback", prep.fallback_count())
if os.environ.get("STAMPS") and hasattr(_lib.lib, "dvq_tuning_buffers"):
    import json
    G = (B * 1024 + 127) // 128
    stamps = torch.zeros((G, 8), dtype=torch.int64, device=dev)
    for name, fn in (("fused_routed", lambda: vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=thr, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None, grain, cmask, gate), conv=conv)),
                     ("fused_dense", lambda: vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None), conv=conv)),
                     ("plain_dense", lambda: vq_assign(h, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(zq, codes, None)))):
        for _ in range(30): fn()
        stamps.zero_()
        dbg = torch.zeros((B * 1024, 4), device=dev)
        _lib.lib.dvq_tuning_buffers(stamps.data_ptr(), dbg.data_ptr())
        fn(); torch.cuda.synchronize()
        _lib.lib.dvq_tuning_buffers(0, 0)
        if name.startswith("fused"):
            d = dbg[:G].cpu().numpy()
            print(json.dumps({"kernel": name, "conv_loop_cycles_median": float(np.median(d[:, 0])), "in_counted_waits": float(np.median(d[:, 1])),
                              "in_barriers": float(np.median(d[:, 2]))}))
        st = stamps.cpu().numpy().astype(np.int64)
        slot, r_in, r_pro, r_l0, c_l0, r_l1, c_l1, r_out = (st[:, i] for i in range(8))
        ok = (r_l1 > r_l0) & (r_out >= r_l1)
        us = lambda d: float(np.median(d[ok])) / 100.0
        print(json.dumps({"kernel": name, "prologue_us": us(r_pro - r_in), "loop_us": us(r_l1 - r_l0), "epilogue_us": us(r_out - r_l1),
                          "span_us": float(r_out[ok].max() - r_in[ok].min()) / 100.0, "loop_cycles": float(np.median((c_l1 - c_l0)[ok]))}))
