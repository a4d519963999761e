"""dvq_ema_accumulate_nchw_f32 alone at B = 256 (K = 1024, 32 x 32 x 256): uniform random codes, dual-grain codes, one hot code"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicvectorquantization_amd import _lib
dev = torch.device("cuda:0")
B, D, K = 256, 256, 1024
z = torch.randn(B, D, 32, 32, device=dev)
cs, vs = torch.zeros(K, device=dev), torch.zeros(K, D, device=dev)
up = lambda x: x.repeat_interleave(2, 1).repeat_interleave(2, 2)
fine = torch.randint(0, K, (B, 32, 32), device=dev)
cases = {"uniform": fine,
         "dual_grain": torch.where(up(torch.rand((B, 16, 16), device=dev) < 0.5), up(torch.randint(0, K, (B, 16, 16), device=dev)), fine).contiguous(),
         "hot_code_10pct": torch.where(torch.rand((B, 32, 32), device=dev) < 0.1, torch.zeros_like(fine), fine).contiguous(),
         "hot_code_50pct": torch.where(torch.rand((B, 32, 32), device=dev) < 0.5, torch.full_like(fine, 7), fine).contiguous(),
         "all_one_code": torch.full_like(fine, 3),
         "some_invalid": torch.where(torch.rand((B, 32, 32), device=dev) < 0.05, torch.full_like(fine, -1), fine).contiguous()}
out = {}
for name, cod in cases.items():
    def run():
        _lib.check(_lib.lib.dvq_ema_accumulate_nchw_f32(z.data_ptr(), cod.data_ptr(), B, D, 1024, K, cs.data_ptr(), vs.data_ptr(), _lib.stream_ptr(dev)), "ema")
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    okm = cod.reshape(-1) >= 0
    ref_cs = torch.bincount(cod.reshape(-1)[okm], minlength=K).float()
    ref_vs = torch.zeros(K, D, device=dev, dtype=torch.float64).index_add_(0, cod.reshape(-1)[okm], z.permute(0, 2, 3, 1).reshape(-1, D).double()[okm])
    err = float(((vs.double() - ref_vs).abs() / (1e-6 + ref_vs.abs().max())).max())
    out[name] = {"us": e0.elapsed_time(e1) / 50 * 1e3, "count_ok": bool(torch.equal(cs, ref_cs)), "sum_max_err_rel_to_max": err}
print(json.dumps(out))
