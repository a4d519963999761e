"""Diagnostic: time the filter path (pass 1 + resolver + exact list + finalize) under DVQ_TUNE_* settings."""
import sys, os, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dynamicvectorquantization_amd import synth, _lib
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    dev = torch.device('cuda:0')
    B = 256
    E = synth.codebook_trained(1024, 256)
    z = torch.from_numpy(synth.z_tokens(E, B, 32, 32, 2003)).to(dev)
    Et = torch.from_numpy(E).to(dev)
    p, pe = _CodebookPrep(), _CodebookPrep()
    zq0, c0, l0 = vq_assign(z, Et, pe, None, mode=0)
    zq1, c1, l1 = vq_assign(z, Et, p, None, mode=1)
    ok = bool(torch.equal(c0, c1) and torch.equal(zq0, zq1))
    for _ in range(5): vq_assign(z, Et, p, None, mode=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): vq_assign(z, Et, p, None, mode=1)
    e.record(); torch.cuda.synchronize()
    print("RESULT %.1f us exact-match %s loss %s %s" % (s.elapsed_time(e) / 20 * 1000, ok, l0.tolist(), l1.tolist()))
else:
    for cfg in sys.argv[1:]:
        env = dict(os.environ)
        for kv in cfg.split(","):
            k, v = kv.split("=")
            env["DVQ_TUNE_" + k] = v
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(cfg, [l for l in out.stdout.splitlines() if l.startswith("RESULT")] or out.stderr[-300:])
