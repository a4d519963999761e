"""Serial (one stream) routed dual step at several batch sizes with the resolver's code tiles cut into 1 / 2 / 4 / 8 slices
(tuning build: DVQ_LIBRARY=.../libdvq_tuning.so): a small batch queues few tokens (B = 64: ~70 resolver workgroups on 256 CUs), each
of which streams the whole code image; slices spread that over more CUs at the price of the merge (atomicMin + ticket)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth, _lib
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign_routed_dual
dev = torch.device("cuda:0")
THR = 1.6777750253677368
E = synth.codebook_trained(1024, 256)
Et = torch.from_numpy(E).to(dev)
res = {}
for B in (16, 64, 128, 256):
    hf = torch.from_numpy(synth.z_tokens(E, min(B, 32), 32, 32, 2903)).to(dev).repeat((B + 31) // 32, 1, 1, 1)[:B].contiguous()
    hc = torch.from_numpy(synth.z_tokens(E, min(B, 32), 16, 16, 2913)).to(dev).repeat((B + 31) // 32, 1, 1, 1)[:B].contiguous()
    ent = torch.from_numpy(synth.entropy_map(5903, min(B, 32), 16, 16)).to(dev).repeat((B + 31) // 32, 1, 1)[:B].contiguous()
    for ns in (1, 2, 4, 8):
        _lib.lib.dvq_tuning_set(b"res_slices", ns)
        prep = _CodebookPrep()
        r = vq_assign_routed_dual(hc, hf, Et, prep, entropy=ent, threshold=THR)
        out = (r["zq"], r["codes"], r["loss"], r["indices"], r["codebook_mask"], r["gate"])
        ref = r["codes"].clone() if ns == 1 else ref
        for _ in range(30):
            vq_assign_routed_dual(hc, hf, Et, prep, entropy=ent, threshold=THR, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(200):
                vq_assign_routed_dual(hc, hf, Et, prep, entropy=ent, threshold=THR, out=out)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 200 * 1e3)
        res["B%d_slices%d" % (B, ns)] = {"us": round(sorted(ts)[2], 2), "same_codes": bool(torch.equal(out[1], ref)), "queued": prep.fallback_count()[0]}
print(json.dumps(res))
