"""Where the host time of one module call goes (cProfile over 3000 calls of VectorQuantizer2.forward at configs[0] size, queue drained every 50).
usage: python tools/host_profile.py [vqgan|vq2|encode]"""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.quantize import VectorQuantizer2, VectorQuantize2

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "vqgan"
E = synth.codebook_trained(1024, 256)
z = torch.from_numpy(synth.z_tokens(E, 4, 16, 16, 2001)).to(dev)
if which == "vqgan":
    m = VectorQuantizer2(1024, 256, beta=0.25, legacy=False).to(dev).eval()
    fn = lambda: m(z)
else:
    m = VectorQuantize2(1024, 256).to(dev).eval()
    cm = torch.ones(4, 1, 16, 16, device=dev)
    fn = lambda: m(z, cm)
with torch.no_grad():
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        fn()
        if i % 50 == 49:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("wall per call %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3000):
        fn()
        if i % 50 == 49:
            torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
