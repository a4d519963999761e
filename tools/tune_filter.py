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
    p = _CodebookPrep()
    for _ in range(5): vq_assign(z, Et, p, None, mode=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): vq_assign(z, Et, p, None, mode=1)
    e.record(); torch.cuda.synchronize()
    print("RESULT %.1f us" % (s.elapsed_time(e) / 20 * 1000))
else:
    for nw in (8, 4):
        for st in (0, 2, 4, 8, 12):
            env = dict(os.environ, DVQ_TUNE_NW=str(nw), DVQ_TUNE_STAGGER=str(st))
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
            print("nw", nw, "stagger", st, [l for l in out.splitlines() if l.startswith("RESULT")])
