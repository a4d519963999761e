"""Two measurements on one box (BASELINE configs[2] step, B = 256, K = 1024):
 (1) whole routed step issued round-robin on 1 / 2 / 3 HIP streams (own outputs and workspace per stream): does the
     latency-bound tail of a step (resolver, list kernel, counter zero, launch gaps) hide under the next batch's pass 1?
 (2) socket power and shader clock (hwmon sysfs / rocm-smi, whichever the box lets an ordinary user read) sampled while
     each of these runs back to back for ~2.5 s: idle, the select kernel (pure HBM copy), pass 1 on random latents,
     pass 1 on zero latents, the whole step."""
import glob, json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual
from dynamicvectorquantization_amd.router import route_select_dual_entropy
B, K = 256, 1024
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
THR = 1.6777750253677368
En = synth.codebook_trained(K, 256)
b0 = 32
tile = lambda x: torch.cat([torch.roll(x, 5 * k, -1) for k in range((B + b0 - 1) // b0)], 0)[:B].contiguous()
hf, hc, ent, E = tile(t(synth.z_tokens(En, b0, 32, 32, 2903))), tile(t(synth.z_tokens(En, b0, 16, 16, 2913))), tile(t(synth.entropy_map(5903, b0, 16, 16))), t(En)


class Outs:
    def __init__(self):
        self.grain = torch.empty((B, 16, 16), dtype=torch.int64, device=dev)
        self.cmask = torch.empty((B, 1, 32, 32), device=dev)
        self.zq = torch.empty_like(hf)
        self.codes = torch.empty((B, 32, 32), dtype=torch.int64, device=dev)
        self.loss = torch.empty(2, device=dev)
        self.gate = torch.empty((B, 16, 16, 2), dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream()


prep = _CodebookPrep()
outs = [Outs() for _ in range(3)]


def routed(o, mode=_lib.MODE_FILTER):
    vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR, mode=mode,
                          out=(o.zq, o.codes, o.loss if mode == _lib.MODE_FILTER else None, o.grain, o.cmask, o.gate))


def run_streams(S, n):
    for i in range(n):
        o = outs[i % S]
        with torch.cuda.stream(o.stream):
            routed(o)


def time_streams(S, n=300, warm=60):
    run_streams(S, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_streams(S, n)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e6, 1)


routed(outs[0]); torch.cuda.synchronize()      # codebook image built before a second stream uses it
res = {"streams_us_per_step": {}}
for rep in range(2):
    for S in (1, 2, 3):
        res["streams_us_per_step"].setdefault(str(S), []).append(time_streams(S))
# outputs of the streams agree
torch.cuda.synchronize()
res["streams_same_outputs"] = bool(all(torch.equal(outs[0].codes, o.codes) and torch.equal(outs[0].zq, o.zq) for o in outs[1:]))

# ---- power / clock sampling ---------------------------------------------------------------------
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None
def sample_sysfs():
    s = {}
    for h in hw:
        for f in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
            v = read(os.path.join(h, f))
            if v is not None:
                s[f] = float(v)
    for c in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        v = read(c)
        if v:
            cur = [ln for ln in v.splitlines() if ln.endswith("*")]
            if cur:
                s["pp_dpm_sclk"] = cur[0]
    return s
def sample_smi():
    try:
        o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(o[o.index("{"):])
        card = next(iter(j.values()))
        return {k: v for k, v in card.items() if "ower" in k or "sclk" in k or "mclk" in k or "fclk" in k}
    except Exception as e:
        return {"error": repr(e)[:200]}

h_dual = torch.empty_like(hf)
z0 = torch.zeros_like(hf)
o0 = outs[0]
def w_idle():
    time.sleep(0.02)
def w_copy():
    for _ in range(20):
        route_select_dual_entropy(ent, THR, hc, hf, out=(h_dual, o0.grain, o0.cmask, o0.gate))
def w_p1():
    for _ in range(20):
        vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(o0.zq, o0.codes, None))
def w_p1_codes():
    for _ in range(20):
        vq_assign(hf, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(None, o0.codes, None))
def w_p1_zero():
    for _ in range(20):
        vq_assign(z0, E, prep, None, mode=_lib.MODE_FILTER_PASS1, out=(o0.zq, o0.codes, None))
def w_step():
    for _ in range(20):
        routed(o0)
res["power"] = {}
for name, fn in (("idle", w_idle), ("hbm_copy_select_kernel", w_copy), ("pass1_random", w_p1), ("pass1_codes_only", w_p1_codes),
                 ("pass1_zero_latents", w_p1_zero), ("whole_step", w_step), ("idle_after", w_idle)):
    samples, smi = [], [None]
    stop = [False]
    def sampler():
        time.sleep(0.8)
        smi[0] = sample_smi()
        while not stop[0]:
            samples.append(sample_sysfs())
            time.sleep(0.05)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3.0:
        fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    agg = {}
    for k in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
        v = [s[k] for s in samples if k in s]
        if v:
            agg[k + "_mean"] = sum(v) / len(v)
            agg[k + "_max"] = max(v)
    d = [s.get("pp_dpm_sclk") for s in samples if "pp_dpm_sclk" in s]
    if d:
        agg["pp_dpm_sclk_last"] = d[-1]
    res["power"][name] = {"us_per_call": round(dt / max(n, 1) * 1e6, 1) if name.startswith(("hbm", "pass1", "whole")) else None,
                          "sysfs": agg, "rocm_smi": smi[0], "n_samples": len(samples)}
print(json.dumps(res))
