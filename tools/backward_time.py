"""dvq_vq_backward_nchw_f32 at BASELINE configs[2] size: bit-equality with the torch expression and microseconds per launch."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib
dev = torch.device("cuda:0")
B, D, HW, K = int(os.environ.get("B", "256")), 256, 1024, 1024
g = torch.Generator(device=dev).manual_seed(5)
z = torch.randn(B, D, HW, device=dev, generator=g); gq = torch.randn(B, D, HW, device=dev, generator=g)
E = torch.randn(K, D, device=dev, generator=g); codes = torch.randint(0, K, (B, HW), device=dev, generator=g)
mask = torch.where(torch.rand(B, HW, device=dev, generator=g) < 0.5, 1.0, 0.25)
gl = torch.tensor([0.7], device=dev); coef = 2 * 0.25 / z.numel()
out = torch.empty_like(z)
def run():
    _lib.check(_lib.lib.dvq_vq_backward_nchw_f32(z.data_ptr(), E.data_ptr(), codes.data_ptr(), mask.data_ptr(), gq.data_ptr(), gl.data_ptr(),
                                                 coef, B, D, HW, K, out.data_ptr(), _lib.stream_ptr(dev)), "bw")
run()
e = E[codes].permute(0, 2, 1)
ref = gq + (gl * torch.tensor(coef, dtype=torch.float32, device=dev)) * ((z - e) * mask[:, None, :])
same = bool(torch.equal(out, ref))
for _ in range(20): run()
torch.cuda.synchronize()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
t.record(); torch.cuda.synchronize()
us = s.elapsed_time(t) / 50 * 1e3
print(json.dumps({"lib": os.environ.get("DVQ_LIBRARY", "product"), "bit_equal_torch": same, "us": round(us, 1),
                  "TB_per_s": round(3 * z.numel() * 4 / us / 1e6, 2)}))
