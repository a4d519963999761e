"""Training-mode forward of VectorQuantize2 at BASELINE configs[2] size (B = 256, 32 x 32 x 256, K = 1024): ms per call and the
kernels it launches (run under rocprofv3 --kernel-trace --stats for the split)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import synth
from dynamicvectorquantization_amd.quantize import VectorQuantize2
dev = torch.device("cuda:0")
K, D, B = 1024, 256, int(os.environ.get("B", "256"))
E = synth.codebook_trained(K, D)
vq = VectorQuantize2(K, D).to(dev).train()
vq.codebook.weight.data[:-1].copy_(torch.from_numpy(E).to(dev))
vq.codebook.embed_ema.copy_(vq.codebook.weight.data[:-1])
vq.codebook.cluster_size_ema.fill_(10.0)
b0 = 32
x = torch.cat([torch.roll(torch.from_numpy(synth.z_tokens(E, b0, 32, 32, 2903)).to(dev), 5 * k, -1) for k in range(B // b0)], 0).contiguous()
mask = torch.ones((B, 1, 32, 32), device=dev)
x.requires_grad_(True)
def step():
    q, loss, info = vq(x, codebook_mask=mask)
    (q.sum() * 1e-6 + loss).backward()
    x.grad = None
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 30
for _ in range(n): step()
torch.cuda.synchronize()
print(json.dumps({"B": B, "train_forward_backward_ms": (time.perf_counter() - t0) / n * 1e3}))
with torch.no_grad():
    vq.eval()
    for _ in range(5): vq(x, codebook_mask=mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): vq(x, codebook_mask=mask)
    torch.cuda.synchronize()
    print(json.dumps({"eval_forward_ms": (time.perf_counter() - t0) / n * 1e3}))
