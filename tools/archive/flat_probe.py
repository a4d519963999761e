"""row-major [N, D] latents (N = 262 144, D = 256, K = 1024) through the filter path:
   form "nchw": HW == 1 through the NCHW kernel (lane = token, 4-byte accesses 1 KiB apart: round 4's path),
   form "flat": the row-major form of pass 1 (16-byte accesses along a token's row), next to the NCHW op on the SAME
   values laid out [B, D, 32, 32] (the kernel the form is measured against).
   DVQ_LIBRARY=<...>/libdvq_tuning.so python tools/flat_probe.py            -> JSON (HIP events)
   ... under rocprofv3 --kernel-trace [--pmc FETCH_SIZE | WRITE_SIZE]: FLAT_PROBE_ONE=nchw|flat|ref runs one form only"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
B, K, D = int(os.environ.get("AB_B", "256")), 1024, 256
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
En = synth.codebook_trained(K, D)
b0 = min(B, 32)
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
z_nchw = tile(t(synth.z_tokens(En, b0, 32, 32, 2903)))                       # [B, D, 32, 32]
z_rows = z_nchw.permute(0, 2, 3, 1).reshape(-1, D).contiguous()               # [N, D]: the same tokens row-major
E = t(En)
N = z_rows.shape[0]
mask_rows = torch.where(torch.rand(N, device=dev) < 0.5, 1.0, 0.25)
mask_nchw = mask_rows.reshape(B, 1, 32, 32)
one = os.environ.get("FLAT_PROBE_ONE")
has_switch = hasattr(_lib.lib, "dvq_tuning_set")


def run(form, reps, mode=_lib.MODE_FILTER):
    prep = _CodebookPrep()
    if form == "ref":
        out = (torch.empty_like(z_nchw), torch.empty((B, 32, 32), dtype=torch.int64, device=dev), torch.empty(2, device=dev))
        f = lambda: vq_assign(z_nchw, E, prep, mask_nchw, out=out, mode=mode)
    else:
        if has_switch:
            _lib.lib.dvq_tuning_set(b"flat", 1 if form == "flat" else 0)
        out = (torch.empty_like(z_rows), torch.empty((N,), dtype=torch.int64, device=dev), torch.empty(2, device=dev))
        f = lambda: vq_assign(z_rows, E, prep, mask_rows, out=out, mode=mode)
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps, out


res = {"N": N, "D": D, "K": K, "algorithmic_bytes_per_launch": N * 2060 + K * D * 4}
forms = [one] if one else ["ref", "nchw", "flat", "ref", "nchw", "flat"]
outs = {}
for fm in forms:
    if fm != "ref" and fm == "nchw" and not has_switch:
        continue
    us, out = run(fm, 200)
    res.setdefault(fm + "_op_us", []).append(round(us, 2))
    us1, _ = run(fm, 200, _lib.MODE_FILTER_PASS1)
    res.setdefault(fm + "_pass1_us_incl_zero_kernel", []).append(round(us1, 2))
    outs[fm] = out
if "flat" in outs and "ref" in outs:
    zq_f, c_f, l_f = outs["flat"]
    zq_r, c_r, l_r = outs["ref"]
    res["flat_equals_nchw_layout_op"] = bool(torch.equal(c_f.reshape(B, 32, 32), c_r) and
                                             torch.equal(zq_f.reshape(B, 32, 32, D).permute(0, 3, 1, 2), zq_r))
    res["loss_rel_diff"] = abs(float(l_f[1]) - float(l_r[1])) / abs(float(l_r[1]))
if "nchw" in outs and "flat" in outs:
    res["flat_equals_hw1_form"] = bool(torch.equal(outs["flat"][1], outs["nchw"][1]) and torch.equal(outs["flat"][0], outs["nchw"][0]))
print(json.dumps(res))
