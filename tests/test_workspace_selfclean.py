"""The filter-path ops leave their workspace CLEAN (csrc/dvq_common.h: DVQ_C_*; include/dvq.h: DVQ_MODE_WS_CLEAN): the list
kernel's finishing workgroup puts the queue counters back to zero, each chunk's last resolver slice its ticket pair, so in the
steady state no zeroing kernel is launched in front of an op -- and the routed op queues ONE record per coarse cell (its rep x rep
output positions are copies of one vector; the resolver corrects all of them).

The reference has no counterpart of any of this (one `VectorQuantize2.forward`, quantize2_mask.py:157-191, is a chain of torch
ops); what must hold is that the op's OUTPUT is what the exact mode gives, bit for bit, on every call -- a counter that is not
back at zero, or a copy the resolver forgot, shows up as wrong codes on the NEXT call or only under load, so these tests repeat
the op many times on fresh near-tie data, with three streams in flight and a bandwidth hog beside them, and compare every call
with DVQ_MODE_EXACT on the device."""
import numpy as np
import pytest
import torch

from dynamicvectorquantization_amd import _lib, synth
from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign, vq_assign_routed_dual

pytestmark = pytest.mark.gpu
THR = 1.6777750253677368


def _near_tie_batch(E, B, H, W, gen, frac_tie=0.15):
    """latents on the device: clustered around codes, a share of them placed between TWO codes (undecided in pass 1)"""
    K, D = E.shape
    n = B * H * W
    j = torch.randint(0, K, (n,), device=E.device, generator=gen)
    j2 = torch.randint(0, K, (n,), device=E.device, generator=gen)
    z = E[j] + 0.3 * torch.randn(n, D, device=E.device, generator=gen)
    tie = torch.rand(n, device=E.device, generator=gen) < frac_tie
    mid = 0.5 * (E[j] + E[j2]) + 1e-4 * torch.randn(n, D, device=E.device, generator=gen)
    z = torch.where(tie[:, None], mid, z)
    return z.reshape(B, H, W, D).permute(0, 3, 1, 2).contiguous()


def test_every_call_equals_exact_mode_three_streams_under_load(dev):
    """B = 64 ... 256 dual-grain routed op, 3 streams x 12 calls on fresh near-tie data, a copy kernel hogging HBM beside them"""
    E = torch.from_numpy(synth.codebook_trained(1024, 256)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    hog_s = torch.cuda.Stream(dev)
    hog_a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    hog_b = torch.empty_like(hog_a)
    prep = _CodebookPrep()
    prep_x = _CodebookPrep()
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    queued_total = 0
    for B in (256, 64, 200):
        slots = []
        for s in streams:
            slots.append(dict(hf=None, hc=None, ent=None))
        for it in range(12):
            with torch.cuda.stream(hog_s):
                hog_b.copy_(hog_a, non_blocking=True)
            for si, s in enumerate(streams):
                hf = _near_tie_batch(E, B, 32, 32, gen)
                hc = _near_tie_batch(E, B, 16, 16, gen)
                ent = torch.rand(B, 16, 16, device=dev, generator=gen) * 3.3
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    r = vq_assign_routed_dual(hc, hf, E, prep, entropy=ent, threshold=THR)
                    x = vq_assign_routed_dual(hc, hf, E, prep_x, entropy=ent, threshold=THR, mode=_lib.MODE_EXACT)
                    bad += (r["codes"] != x["codes"]).sum() + (r["zq"] != x["zq"]).sum() + (r["indices"] != x["indices"]).sum()
                    bad += (~torch.isclose(r["loss"], x["loss"], rtol=1e-5, atol=0)).sum()
                    for t_ in (hf, hc, ent):
                        t_.record_stream(s)
            if it == 5:
                torch.cuda.synchronize(dev)
                queued_total += prep.fallback_count()[0]
    torch.cuda.synchronize(dev)
    assert int(bad.item()) == 0
    assert queued_total > 0                                    # the resolver had work


def test_small_batches_many_calls_dense_and_clean_flag(dev):
    """configs[0]-sized dense ops, 300 calls back to back on one workspace: it is zero-filled at allocation and every call runs
    on what its predecessor left (DVQ_MODE_WS_CLEAN: no zero kernel); odd token counts; D = 64 / 128 / 256; K = 2048 and 4096
    (sliced resolver: chunk tickets)"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    for D, K, B, H in ((256, 1024, 4, 16), (64, 300, 3, 9), (128, 513, 5, 8), (256, 2048, 1, 5), (256, 4096, 2, 16)):
        E = torch.from_numpy(synth.codebook_trained(K, D, seed=40 + D)).to(dev)
        prep, prep_x = _CodebookPrep(), _CodebookPrep()
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        for it in range(300 if D == 256 and K == 1024 else 60):
            z = _near_tie_batch(E, B, H, H, gen, frac_tie=0.3)
            m = torch.where(torch.rand(B, 1, H, H, device=dev, generator=gen) < 0.5, 1.0, 0.25)
            zq, codes, loss = vq_assign(z, E, prep, m)
            zx, cx, lx = vq_assign(z, E, prep_x, m, mode=_lib.MODE_EXACT)
            bad += (codes != cx).sum() + (zq != zx).sum() + (~torch.isclose(loss, lx, rtol=1e-5, atol=0)).sum()
        assert int(bad.item()) == 0, (D, K, B, H)
        ws = prep._last_ws[1]
        assert ws.clean
        # the op left every live word of the counter block zero: ints 2 .. 79 (0, 1 are the report, 80.. the mailbox)
        off = _lib.lib.dvq_vq_assign_fallback_count_offset(B, D, H * H, K)
        live = ws.t[off:off + 512].view(torch.int32)[2:80]
        assert int(live.abs().sum().item()) == 0


def test_dirty_workspace_without_the_flag_is_fine(dev):
    """a caller that does not track cleanliness (plain ABI use, mode without DVQ_MODE_WS_CLEAN) may hand in ANY bytes"""
    E = torch.from_numpy(synth.codebook_trained(1024, 256)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    z = _near_tie_batch(E, 8, 16, 16, gen, frac_tie=0.3)
    prep, prep_x = _CodebookPrep(), _CodebookPrep()
    zx, cx, lx = vq_assign(z, E, prep_x, mode=_lib.MODE_EXACT)
    for _ in range(3):
        vq_assign(z, E, prep)
        ws = prep._last_ws[1]
        ws.t.fill_(0x5A)                                       # garbage everywhere: counters, tickets, lists
        ws.clean = False
        zq, codes, loss = vq_assign(z, E, prep)
        assert torch.equal(codes, cx) and torch.equal(zq, zx)
        assert torch.allclose(loss, lx, rtol=1e-5, atol=0)
        # ... and after the profiling mode, which leaves a workspace of its own dirty
        vq_assign(z, E, prep, mode=_lib.MODE_FILTER_PASS1)
        assert not prep._last_ws[1].clean
        zq, codes, loss = vq_assign(z, E, prep, mode=_lib.MODE_FILTER_PASS1)
        zq, codes, loss = vq_assign(z, E, prep)
        assert torch.equal(codes, cx) and torch.equal(zq, zx)


def _routed_inputs(E, B, G, gen, dev, frac_tie, tiny=False):
    """branches of a dual (G = 2: 16x16 / 32x32) or triple (G = 3: 8x8 / 16x16 / 32x32) batch, near-tie latents in EVERY branch
    (tiny: 0.002 N(0, 1) latents, the scale of a default-init codebook -- every token within the bound of several codes)"""
    mk = (lambda h: 0.002 * torch.randn(B, E.shape[1], h, h, device=dev, generator=gen)) if tiny else \
         (lambda h: _near_tie_batch(E, B, h, h, gen, frac_tie))
    hf = mk(32)
    hm = mk(16)
    hc = mk(8) if G == 3 else None
    cells = 8 if G == 3 else 16
    gate = torch.randn(B, cells, cells, G, device=dev, generator=gen)
    return hc, hm, hf, gate


@pytest.mark.parametrize("kind", ["near_tie", "duplicates", "default_init"])
def test_coarse_cells_queue_one_record_and_every_copy_is_corrected(dev, kind):
    """The rep x rep copies of a coarser cell are undecided together; one record is queued for them (RecMeta.rep) and the
    resolver's correction -- code, z_q, loss term -- or its hand-off to the exact list must reach every copy:
    near_tie: ordinary resolver work; duplicates: 64 identical codes per vector -> candidate overflow -> every position of the
    token goes to the exact list; default_init (tie stress, everything undecided): the queue shards overflow in pass 1 -> the
    cell's first position lists all of them.  Dual and triple, with and without the models' quant_conv fused in."""
    from dynamicvectorquantization_amd.quantize import vq_assign_routed_triple
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    K, D, B = 1024, 256, 6
    if kind == "duplicates":
        E = torch.from_numpy(synth.codebook_trained(16, D, seed=3)).to(dev).repeat(64, 1).contiguous()
    elif kind == "default_init":
        E = torch.from_numpy(synth.codebook_default_init(K, D)).to(dev)
    else:
        E = torch.from_numpy(synth.codebook_trained(K, D)).to(dev)
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        q, _ = torch.linalg.qr(torch.randn(D, D, device=dev, generator=gen))
        conv.weight.copy_(q.reshape(D, D, 1, 1))
        conv.bias.copy_(0.01 * torch.randn(D, device=dev, generator=gen))
    for G in (2, 3):
        hc, hm, hf, gate = _routed_inputs(E, B, G, gen, dev, 0.3, tiny=(kind == "default_init"))
        prep, prep_x = _CodebookPrep(), _CodebookPrep()
        if G == 2:
            r = vq_assign_routed_dual(hm, hf, E, prep, gate=gate)
            x = vq_assign_routed_dual(hm, hf, E, prep_x, gate=gate, mode=_lib.MODE_EXACT)
        else:
            r = vq_assign_routed_triple(hc, hm, hf, E, prep, gate)
            x = vq_assign_routed_triple(hc, hm, hf, E, prep_x, gate, mode=_lib.MODE_EXACT)
        assert torch.equal(r["codes"], x["codes"]) and torch.equal(r["zq"], x["zq"]) and torch.equal(r["indices"], x["indices"]), (kind, G)
        assert torch.allclose(r["loss"], x["loss"], rtol=1e-5, atol=0), (kind, G, r["loss"], x["loss"])
        queued, listed = prep.fallback_count()
        coarse_positions = int((x["codebook_mask"] < 1.0).sum().item())
        assert coarse_positions > 0 and queued + listed > 0
        if kind == "duplicates":
            assert listed > 0                                  # candidate overflow (or shard overflow) reached the exact list
        # the model order with the conv fused into pass 1: the reference op on THAT h (the op returns it with h_buf)
        hb = torch.empty_like(hf)
        if G == 2:
            rc = vq_assign_routed_dual(hm, hf, E, prep, gate=gate, conv=conv, h_buf=hb)
        else:
            rc = vq_assign_routed_triple(hc, hm, hf, E, prep, gate, conv=conv, h_buf=hb)
        zx, cx, lx = vq_assign(hb, E, prep_x, rc["codebook_mask"], mode=_lib.MODE_EXACT)
        assert torch.equal(rc["codes"], cx) and torch.equal(rc["zq"], zx), (kind, G, "conv")
        assert torch.allclose(rc["loss"], lx, rtol=1e-5, atol=0)
        # ... and without h_buf (scratch rows only for the exact list's tokens): same bits
        if G == 2:
            rn = vq_assign_routed_dual(hm, hf, E, prep, gate=gate, conv=conv)
        else:
            rn = vq_assign_routed_triple(hc, hm, hf, E, prep, gate, conv=conv)
        assert torch.equal(rn["codes"], rc["codes"]) and torch.equal(rn["zq"], rc["zq"]), (kind, G, "conv, no h_buf")
