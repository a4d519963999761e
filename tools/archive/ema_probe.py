import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from dynamicvectorquantization_amd import _lib, synth
dev = torch.device("cuda:0")
"""EMA statistics op (dvq_ema_accumulate_nchw_f32) against an fp64 index_add: counts equal, sums within rounding; time per
call for several code distributions: uniform, skew3 (rand^3: 10 % of the tokens on code 0), cells2x2 (half of the 2 x 2
cells of a 32 x 32 grid carry one code: dual grain at ratio 0.5)."""
for (B, K, D, HW, kind) in [(256, 1024, 256, 1024, "uniform"), (256, 1024, 256, 1024, "skew3"), (256, 1024, 256, 1024, "cells2x2"),
                            (64, 512, 64, 256, "uniform"), (16, 2048, 128, 1024, "skew3"), (3, 1000, 256, 100, "uniform"),
                            (8, 16384, 256, 1024, "uniform")]:
    N = B * HW
    g = torch.Generator(device="cpu").manual_seed(5)
    z = torch.randn((B, D, HW), generator=g).to(dev)
    if kind == "uniform":
        cod = torch.randint(0, K, (B, HW), generator=g).to(dev)
    elif kind == "skew3":
        cod = (torch.rand((B, HW), generator=g) ** 3 * K).long().clamp_(0, K - 1).to(dev)
    else:
        side = int(HW ** 0.5)
        fine = torch.randint(0, K, (B, side, side), generator=g)
        coarse = torch.randint(0, K, (B, side // 2, side // 2), generator=g)
        pick = torch.rand((B, side // 2, side // 2), generator=g) < 0.5
        up = lambda t: t.repeat_interleave(2, 1).repeat_interleave(2, 2)
        cod = torch.where(up(pick), up(coarse), fine).reshape(B, HW).to(dev)
    cod[0, :5] = -1
    cs = torch.empty(K, device=dev); vs = torch.empty((K, D), device=dev)
    def run():
        _lib.check(_lib.lib.dvq_ema_accumulate_nchw_f32(z.data_ptr(), cod.data_ptr(), B, D, HW, K, cs.data_ptr(), vs.data_ptr(), _lib.stream_ptr(dev)), "ema")
    run(); torch.cuda.synchronize()
    # reference in float64
    zt = z.permute(0, 2, 1).reshape(N, D).double(); cf = cod.reshape(N)
    ok = cf >= 0
    ref = torch.zeros((K, D), dtype=torch.float64, device=dev).index_add_(0, cf[ok], zt[ok])
    refc = torch.bincount(cf[ok], minlength=K).double()
    err = float((vs.double() - ref).abs().max()); scale = float(ref.abs().max())
    assert torch.equal(cs.double(), refc), "counts"
    for _ in range(5): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); us = (time.perf_counter() - t) / 20 * 1e6
    print(dict(B=B, K=K, D=D, HW=HW, kind=kind, us=round(us, 1), max_err=err, ref_max=scale, rel=err / scale))
