#!/usr/bin/env python3
"""Benchmark of the DQ-VAE vector-quantization hot path on MI355X.

One "step" = one pass of the hot path over one batch that is already resident in HBM:
    entropy-threshold gate + dual-granularity route select (+ codebook_mask), fused in one kernel
    -> VQ nearest-codebook assignment (codes, z_q, masked commitment loss)
on BASELINE.json configs[2] (dqvae-entropy-dual-r05: B=256 per GPU, 32x32x256 latents, K=1024).
The 1x1 quant_conv between select and VQ is a vendor GEMM outside the path
and is not run.  With N > 1 ranks every rank encodes its own 256 images (weak scaling) and the
step ends by launching the (single, packed) RCCL all-gather of the emitted code / grain indices and
the loss pair; that exchange runs asynchronously under the next step's kernels and every exchange
is waited for and unpacked inside the timed region.

Contract: python bench.py --gpus N --steps K --warmup W  -> ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--codes", type=int, default=1024)
    ap.add_argument("--mode", choices=["exact", "filter"], default="filter")
    ap.add_argument("--spinup", type=int, default=100,
                    help="untimed steps before the warmup that bring the GPU out of its idle power state "
                         "(the first ~30 ms after idle run ~10 %% slower); reported in config.spinup_steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs under a 16-CPU quota; 256 OpenMP threads there just get throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p_ + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(E, B_gpu, target_s):
    """the oracle (explicit-order C restatement of what the reference's torch-CPU path computes)
    timed on this box's host cores on a bounded sample of the same workload"""
    from dynamicvectorquantization_amd import synth
    from oracle import oracle
    oracle.build()
    cores = usable_cpus()
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)      # the oracle's OpenMP team
    except Exception:
        pass
    nb = min(B_gpu, 64)
    hf = synth.z_tokens(E, nb, 32, 32, 2903)
    hc = synth.z_tokens(E, nb, 16, 16, 2913)
    ent = synth.entropy_map(5903, nb, 16, 16)

    # preallocated outputs, like the GPU step (fresh 64-MB numpy arrays would page-fault on every pass)
    o_sel = (np.empty_like(hf), np.empty((nb, 16, 16), np.int64), np.empty((nb, 1, 32, 32), np.float32))
    o_vq = (np.empty_like(hf), np.empty((nb, 1024), np.int64))

    def one_pass():
        gate = oracle.entropy_gate(ent, 1.6777750253677368)
        sel = oracle.route_select_dual(gate, hc, hf, out=o_sel)
        oracle.vq_assign_nchw(sel["h_dual"], E, sel["codebook_mask"], out=o_vq)

    one_pass()                                                   # warm (page-in, OpenMP team)
    reps, t0 = 0, time.perf_counter()
    while True:                                                  # repeat the sample for ~target_s
        one_pass()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_s or reps >= 10000:
            break
    return {"value": nb * reps / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d passes over %d images of the same workload (gate + select + VQ assign), oracle C "
                      "port: OpenMP over 4-token x 32-code register tiles (%d threads) + AVX2 FMA chains, "
                      "%.1f s" % (reps, nb, cores, dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()      # (a 1-GPU box can still exercise the N > 1 code path)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DVQ_BENCH_BACKEND", "nccl")     # nccl = RCCL over xGMI; gloo only for tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from dynamicvectorquantization_amd import _lib, synth
    from dynamicvectorquantization_amd.encode import all_gather_codes
    from dynamicvectorquantization_amd.quantize import _CodebookPrep, vq_assign
    from dynamicvectorquantization_amd.router import route_select_dual_entropy

    B, K, D, H, W = a.batch, a.codes, 256, 32, 32
    mode = _lib.MODE_EXACT if a.mode == "exact" else _lib.MODE_FILTER
    E_np = synth.codebook_trained(K, D)
    off = rank * B
    h_fine = torch.from_numpy(synth.z_tokens(E_np, B, H, W, 2903, image_offset=off)).to(dev)
    h_coarse = torch.from_numpy(synth.z_tokens(E_np, B, H // 2, W // 2, 2913, image_offset=off)).to(dev)
    ent = torch.from_numpy(synth.entropy_map(5903, B, H // 2, W // 2, image_offset=off)).to(dev)
    E = torch.from_numpy(E_np).to(dev)
    thr = 1.6777750253677368                       # imagenet_train JSON, key "50" (ratio 0.5)
    prep = _CodebookPrep()
    # preallocated outputs: the step allocates nothing
    h_dual = torch.empty_like(h_fine)
    grain = torch.empty((B, H // 2, W // 2), dtype=torch.int64, device=dev)
    cmask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    zq = torch.empty_like(h_fine)
    codes = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    gate = torch.empty((B, H // 2, W // 2, 2), dtype=torch.int64, device=dev)

    def step(i=None):
        # entropy-threshold gate + routing tail: one kernel (the int64 gate is written as a by-product)
        route_select_dual_entropy(ent, thr, h_coarse, h_fine, out=(h_dual, grain, cmask, gate))
        if i is not None:
            ev[i][0].record()
        vq_assign(h_dual, E, prep, cmask, beta=0.25, mode=mode, out=(zq, codes, loss))
        if i is not None:
            ev[i][1].record()
        if world > 1:
            # one packed all-gather per step, in flight while the next step's kernels run; the previous
            # step's exchange is completed (stream wait + unpack) first, so at most one is pending
            if pending:
                pending.pop().wait()
            pending.append(all_gather_codes(codes, grain, loss[0] * (B * H * W * D), B * H * W * D, K, B * world,
                                            async_op=True))
        return codes, grain, loss[0]

    pending = []

    def fence():
        if pending:
            pending.pop().wait()           # the last exchange completes inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.spinup):              # clock / power-state ramp, see --spinup
        step()
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))     # the whole vq_assign op (all its kernels)
    # the dominant kernel alone (same launch, DVQ_MODE_FILTER_PASS1 / the single exact kernel), HIP events
    # on the launch stream, interleaved after the timed region so it does not perturb `value`
    dom_mode = _lib.MODE_EXACT if a.mode == "exact" else _lib.MODE_FILTER_PASS1
    prep_dom = _CodebookPrep()
    dom_loss = None if a.mode == "filter" else loss
    dom_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for i in range(-3, a.steps):
        if i >= 0:
            dom_ev[i][0].record()
        vq_assign(h_dual, E, prep_dom, cmask, beta=0.25, mode=dom_mode, out=(zq, codes, dom_loss))
        if i >= 0:
            dom_ev[i][1].record()
    torch.cuda.synchronize()
    dom_ms = float(np.mean([s.elapsed_time(e) for s, e in dom_ev]))
    N = B * H * W
    alg_bytes = N * (D * 4 * 2 + 8 + 4) + K * D * 4          # z read + z_q write + int64 code + mask, codebook once
    alg_flops = 2.0 * K * D * N
    gbs = alg_bytes / (dom_ms * 1e-3) / 1e9
    tfs = alg_flops / (dom_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(a.mode, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if a.mode == "filter":
        roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic}
    else:
        roof = {"bound": "mfma", "achieved": tfs, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tfs / FP32_MFMA_PEAK_TF, "traffic": traffic}
    roof.update({"kernel": "vq_assign_filter_kernel<256>" if a.mode == "filter" else "vq_assign_exact_kernel<256>",
                 "kernel_ms": dom_ms, "algorithmic_bytes": alg_bytes, "algorithmic_flops": alg_flops,
                 "hbm_gbps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "fp32_tflops_equiv": tfs,
                 "whole_op_ms": kern_ms, "whole_op_hbm_gbps": alg_bytes / (kern_ms * 1e-3) / 1e9,
                 "whole_op_note": "vq_assign op = all its kernels (filter: counter memset + filter kernel + resolver + exact-list kernel with the fused loss finalize; exact: kernel + finalize)"})
    if rank == 0:
        out = {
            "metric": "images encoded/sec (VQ hot path: gate + route select + VQ assign), 256x256 inputs, K=%d" % K,
            "value": B * world * a.steps / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: dqvae-entropy-dual-r05, B=%d per GPU, 32x32x256 latents, "
                                   "K=%d, entropy gate + dual route select + VectorQuantize2 assign "
                                   "(quant_conv not in the path)" % (B, K),
                       "global_batch": B * world, "assign_mode": a.mode, "spinup_steps": a.spinup,
                       "parallelism": "image-parallel x%d, RCCL all-gather of codes" % world},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(E_np, B, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
