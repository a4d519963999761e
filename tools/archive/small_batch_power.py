"""Socket power (rocm-smi) while pass 1 alone runs back to back at small batch sizes: a single generation of workgroups runs its
HBM phases and its matrix phase in lockstep -- is it at the power cap like the B = 256 launch?   usage: python tools/small_batch_power.py"""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
dev = torch.device("cuda:0")
E = synth.codebook_trained(1024, 256)
Et = torch.from_numpy(E).to(dev)


def smi():
    try:
        o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(o[o.index("{"):])
        card = next(iter(j.values()))
        return {k: v for k, v in card.items() if "ower" in k or "sclk" in k}
    except Exception as e:
        return {"error": repr(e)[:100]}


res = {}
for B in (32, 64, 128, 256):
    z = torch.from_numpy(synth.z_tokens(E, min(B, 32), 32, 32, 2903)).to(dev).repeat((B + 31) // 32, 1, 1, 1)[:B].contiguous()
    prep = _CodebookPrep()
    zq, codes, _ = vq_assign(z, Et, prep, None, mode=_lib.MODE_FILTER_PASS1)
    out = (zq, codes, None)
    samples, stop = [], [False]
    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            samples.append(smi()); time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(50):
            vq_assign(z, Et, prep, None, mode=_lib.MODE_FILTER_PASS1, out=out)
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    res["B%d" % B] = {"us_per_launch_incl_zero_kernel": round(dt / n * 1e6, 1), "us_per_image": round(dt / n * 1e6 / B, 3), "smi": samples[1:4]}
print(json.dumps(res))
