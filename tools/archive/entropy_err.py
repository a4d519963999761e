"""fused entropy kernel vs the reference's op sequence (torch ops on the CPU): error statistics on the B = 64 synthetic batch"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.entropy import Entropy
from oracle.entropy_torch import entropy_map
dev = torch.device("cuda:0")
img, noisy = synth.images_flat_noise(5001, 64)
x = torch.from_numpy(img).to(dev)
with torch.no_grad():
    a = Entropy(16, 256, 256)(x).double().cpu().numpy()
    b = entropy_map(x.cpu(), chunk=8).double().numpy()        # CPU: the reference path of record (device exp flushes subnormals)
    b64 = entropy_map(x.double(), chunk=8).cpu().numpy() if os.environ.get("F64") else None
err = np.abs(a - b)
tol = 1e-5 + 1e-5 * np.abs(b)
print("max abs err", err.max(), "at", np.unravel_index(err.argmax(), err.shape), "ref", b.flat[err.argmax()], "violations", int((err > tol).sum()), "of", err.size)
w = np.argsort(err.ravel())[-5:]
for i in w: print(a.flat[i], b.flat[i], err.flat[i], "noisy" if noisy.flat[i] else "flat")
