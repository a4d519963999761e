#!/usr/bin/env python3
"""Benchmark of the DQ-VAE vector-quantization hot path on MI355X.

One "step" = one pass of the hot path over one batch that is already resident in HBM.

  --scaling weak   (default; BASELINE configs[2], dqvae-entropy-dual-r05): every rank encodes its own
                   B = 256 images: entropy-threshold gate + dual-granularity routing + VectorQuantize2
                   assignment (codes, z_q, codebook_mask, grain indices, masked commitment loss).
  --scaling strong (BASELINE configs[3], triple granularity F = 32/16/8): a fixed global batch of 1024
                   images is split over the ranks (128 per GPU at N = 8): fused feature-router gate +
                   triple routing + assignment.

--path routed (default) is the hot path north_star names: gate + routing + assign as ONE op (no quant_conv between
select and quantizer); --path model adds the stage-1 models' 1x1 quant_conv in the reference's order (select ->
quant_conv -> quantizer, dqvae_dual_entropy.py:124-134: select + conv as one kernel, then the dense assign);
--path tokens is the codes-only tokenisation stage 2 consumes (routed assign without z_q + permuter, no host sync);
--path select is round 1's two-kernel path.  Every stream slot owns its INPUTS as well as its outputs (different seeds:
no step re-reads bytes the previous step read).  With N > 1 ranks the step ends by launching the single packed RCCL
all-gather of the emitted code / grain indices and the loss pair; it runs asynchronously under the next step's kernels
and every exchange is waited for and unpacked inside the timed region.

Contract: python bench.py --gpus N --steps K --warmup W  -> ONE JSON line (rank 0).
`--gpus N` with no RANK in the environment makes this process a launcher: it starts N fresh rank
processes (one per GPU, RCCL rendezvous on 127.0.0.1) BEFORE anything touches a GPU, relays rank 0's
JSON line and exits non-zero if any rank fails.  Under `python -m torch.distributed.run` (RANK set) it is
a rank.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
THR_R05 = 1.6777750253677368   # imagenet_train JSON, key "50" (fine ratio 0.5)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--repeats", type=int, default=7,
                    help="the timed region is measured this many times back to back inside one run (each block = exactly --steps "
                         "steps, barrier + synchronize on both sides, max over ranks); ms_per_step / value are the MEDIAN block, "
                         "min / max are reported beside it (a 20-step block is 4 ms: launch ramp and box variance dominate one block)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--batch", type=int, default=None,
                    help="weak: images per GPU (default 256); strong: GLOBAL batch (default 1024)")
    ap.add_argument("--codes", type=int, default=1024)
    ap.add_argument("--mode", choices=["exact", "filter"], default="filter")
    ap.add_argument("--path", choices=["routed", "select", "model", "model2", "tokens", "tokens_model", "tokens_fold", "model_fold"], default="routed",
                    help="routed: ONE assign op straight from the encoder branches (the router select is fused into "
                         "pass 1, h_dual is never written); select: route-select kernel writing h_dual, then the "
                         "dense assign (round-1 path); model: select -> the models' 1x1 quant_conv -> assign, the order a "
                         "reference checkpoint runs, as ONE op (the conv is pass 1's prologue; neither h_dual nor the conv's "
                         "output is written); model2: the same order as two kernels (select + conv kernel writing h, then the "
                         "dense assign); tokens: routed assign, codes only, + permuter; tokens_model: the same behind the models' "
                         "quant_conv (the fused op, codes only), what stage 2's encode_to_z runs on a stage-1 checkpoint; "
                         "tokens_fold: that tokenisation with the quant_conv FOLDED into the codebook (opt-in fold=True: pass 1 "
                         "scores the branches against E W, no conv is computed for decided tokens; same codes); model_fold: the "
                         "model order for loss-free inference by the fold (codes + z_q = codebook[code], no loss)")
    ap.add_argument("--no-model-order", action="store_true",
                    help="default command only (1 GPU, weak, --path routed): skip the short second measurement of the "
                         "model order (--path model: the same step behind the models' 1x1 quant_conv, as one op) that is "
                         "reported under 'model_order' beside the headline")
    ap.add_argument("--spinup", type=int, default=100,
                    help="untimed steps before the warmup that bring the GPU out of its idle power state "
                         "(the first ~30 ms after idle run ~10 %% slower); reported in config.spinup_steps")
    ap.add_argument("--streams", type=int, default=3,
                    help="consecutive steps (independent batches) are issued round-robin on this many HIP streams, each "
                         "with its own outputs / workspace / exchange buffers, so a step's latency-bound tail (resolver, "
                         "list kernel, exchange) runs under the next batch's pass 1; 1 = strictly serial")
    ap.add_argument("--configs", choices=["auto", "full", "small", "off"], default="auto",
                    help="default command only (1 GPU, weak, --path routed, filter): after the timed region, measure and check every "
                         "BASELINE.json config in the reference's own op order (pixels -> entropy map / feature-router gate -> select + "
                         "quant_conv + assign) and report them under 'configs'; auto = full at the default batch, small when --batch is given")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started before this process has made any GPU call
# ------------------------------------------------------------------------------------------------
def launch_ranks(a):
    """Start one rank process per GPU and FAIL FAST: every process is polled; the first non-zero exit (a rank that dies in
    set-up would otherwise leave the others in the rendezvous until torch's 10-30 min timeout) kills the rest and the
    launcher returns non-zero within seconds.  The whole job is bounded by DVQ_BENCH_LAUNCH_TIMEOUT seconds (default 900)."""
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()                                   # rank 0's pipe is drained while everybody is polled
    deadline = time.time() + float(os.environ.get("DVQ_BENCH_LAUNCH_TIMEOUT", "900"))
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            why = "rank %d exited with code %d" % bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() > deadline:
            why = "wall-time bound exceeded (DVQ_BENCH_LAUNCH_TIMEOUT)"
            break
        time.sleep(0.1)
    if why is not None:
        for p in procs:                              # the exact processes started above, nothing by pattern
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 3.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    line = None
    for ln in out0:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)
    if why is not None or line is None:
        print("bench.py launcher: %s; rank exit codes %s" % (why or "no JSON line from rank 0", [p.returncode for p in procs]),
              file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def source_sha16():
    """hash of what decides the timed kernels: bench.py + the csrc sources (rocprof summaries under profiles/ carry it)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "dynamicvectorquantization_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the oracle C port, and the reference's torch-CPU op sequence
# ------------------------------------------------------------------------------------------------
def cpu_baseline(E, target_s):
    """two legs on a bounded sample of the weak-scaling workload (gate + select + VQ assign):
    (1) the oracle: explicit-order C restatement (OpenMP + AVX2) of what the reference computes;
    (2) the reference's own op sequence as torch-CPU ops (addmm + argmin + embedding gather + masked loss,
        quantize2_mask.py:39-46,53,131,172-182; EncoderDual.py:134-149), own code, MKL underneath --
        this is what the reference actually executes and is ~3x slower than (1)."""
    import ctypes

    import numpy as np
    import torch

    from dynamicvectorquantization_amd import synth
    from oracle import oracle
    oracle.build()
    cores = usable_cpus()
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)      # the oracle's OpenMP team
    except Exception:
        pass
    nb = 64
    hf = synth.z_tokens(E, nb, 32, 32, 2903)
    hc = synth.z_tokens(E, nb, 16, 16, 2913)
    ent = synth.entropy_map(5903, nb, 16, 16)
    # preallocated outputs, like the GPU step (fresh 64-MB numpy arrays would page-fault on every pass)
    o_sel = (np.empty_like(hf), np.empty((nb, 16, 16), np.int64), np.empty((nb, 1, 32, 32), np.float32))
    o_vq = (np.empty_like(hf), np.empty((nb, 1024), np.int64))

    def one_pass():
        gate = oracle.entropy_gate(ent, THR_R05)
        sel = oracle.route_select_dual(gate, hc, hf, out=o_sel)
        oracle.vq_assign_nchw(sel["h_dual"], E, sel["codebook_mask"], out=o_vq)

    def loop(fn, seconds):
        fn()                                                     # warm (page-in, thread teams)
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or reps >= 10000:
                return reps, dt

    reps, dt = loop(one_pass, target_s * 0.6)
    res = {"value": nb * reps / dt, "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "%d passes x %d images (gate + select + VQ assign), oracle C port, OpenMP %d threads + AVX2, %.1f s"
                     % (reps, nb, cores, dt)}

    torch.set_num_threads(cores)
    nt = 16
    tE = torch.from_numpy(E)
    thf, thc, tent = torch.from_numpy(hf[:nt]), torch.from_numpy(hc[:nt]), torch.from_numpy(ent[:nt])
    en = (tE * tE).sum(1).unsqueeze(0)

    def torch_pass():
        with torch.no_grad():
            fine = tent > THR_R05
            up = fine.repeat_interleave(2, -1).repeat_interleave(2, -2).unsqueeze(1)
            h = torch.where(up, thf, thc.repeat_interleave(2, -1).repeat_interleave(2, -2))
            m = torch.where(up, 1.0, 0.25).reshape(nt, -1, 1)
            x = h.permute(0, 2, 3, 1).reshape(-1, 256)
            d = torch.addmm((x * x).sum(1, keepdim=True) + en, x, tE.t(), alpha=-2.0)
            code = d.argmin(-1)
            e = tE.index_select(0, code)
            loss = 1.25 * (((e - x) ** 2).reshape(nt, -1, 256) * m).mean()
            zq = (x + (e - x)).reshape(nt, 32, 32, 256).permute(0, 3, 1, 2).contiguous()
        return zq, code, loss

    reps2, dt2 = loop(torch_pass, target_s * 0.4)
    res["torch_ops"] = {"value": nt * reps2 / dt2, "unit": "images/s", "cores": cores, "kind": "port",
                        "sample": "%d passes x %d images, the reference's op sequence as torch-CPU ops, torch %s, %d threads, %.1f s"
                                  % (reps2, nt, torch.__version__, cores, dt2)}
    return res


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def tile_images(base, B):
    """[b0, ...] device tensor -> [B, ...]: image i = base[i % b0] rolled by 5 (i // b0) positions along the last
    axis (the tokens of later images are those of the base images at other positions: same statistics)"""
    import torch
    b0 = base.shape[0]
    if B <= b0:
        return base[:B].contiguous()
    parts = [torch.roll(base, shifts=5 * k, dims=-1) for k in range((B + b0 - 1) // b0)]
    return torch.cat(parts, 0)[:B].contiguous()


class Slot:
    """one set of output tensors (and, for N > 1, exchange buffers) bound to one HIP stream"""
    pass


class WeakDual:
    """BASELINE configs[2]: entropy router + dual routing + VectorQuantize2 assign, B images per rank.
    paths: routed (one op) | select (select kernel + dense assign) | model (select + quant_conv kernel, dense assign) |
    tokens (routed assign without z_q + permuter)"""
    name = "dual"

    def __init__(self, a, rank, world, dev):
        import torch

        from dynamicvectorquantization_amd import _lib, synth
        from dynamicvectorquantization_amd.quantize import _CodebookPrep
        self.a, self.dev, self.world = a, dev, world
        B = self.B = a.batch or 256
        self.Bglobal = B * world
        K, D, H, W = a.codes, 256, 32, 32
        self.K, self.D, self.H, self.W = K, D, H, W
        self.mode = _lib.MODE_EXACT if a.mode == "exact" else _lib.MODE_FILTER
        self.E_np = synth.codebook_trained(K, D)
        self.E = torch.from_numpy(self.E_np).to(dev)
        self.prep = _CodebookPrep()
        self.rank = rank
        self.conv = None
        self.conv_q = None
        if a.path in ("model", "model2", "tokens_model", "tokens_fold", "model_fold"):
            # the models' quant_conv: a random orthogonal 256 x 256 matrix and a bias.  The encoder-branch inputs are generated as
            # x = Q^T (t - b) from tokens t of the usual z_tokens distribution (inputs_np), so that the QUANTIZER sees the same
            # distribution as on the other paths -- conv(x) = t up to rounding, as in a trained model whose conv output sits near
            # its codebook -- instead of tokens scrambled away from every code.
            import numpy as np
            q, _ = np.linalg.qr(synth.normal(6012, (D, D), 0.0, 1.0).astype(np.float64))
            self.conv_q = q.astype(np.float32)
            self.conv_b = synth.normal(6013, (D,), 0.0, 0.1)
            self.conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
            with torch.no_grad():
                self.conv.weight.copy_(torch.from_numpy(self.conv_q.reshape(D, D, 1, 1)).to(dev))
                self.conv.bias.copy_(torch.from_numpy(self.conv_b).to(dev))
        self.permuter = None
        if a.path in ("tokens", "tokens_model", "tokens_fold"):
            from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
            self.permuter = DualGrainSeperatePermuter(coarse_hw=H // 2, fine_hw=H, content_pad_code=K, content_eos_code=K + 1)
        # one set of inputs AND outputs per stream slot, preallocated: the step allocates nothing and no step reads the
        # bytes the previous step read
        self.slots = [self.new_slot(k) for k in range(a.streams)]

    def inputs_np(self, k):
        """slot k's inputs on the host (seeds differ per slot; images b0.. of a batch are the first b0 rolled)"""
        import numpy as np

        from dynamicvectorquantization_amd import synth
        B, H, W = self.B, self.H, self.W
        off = self.rank * B
        b0 = min(B, 128)

        def tile(base):
            if B <= b0:
                return base[:B]
            parts = [np.roll(base, 5 * j, axis=-1) for j in range((B + b0 - 1) // b0)]
            return np.ascontiguousarray(np.concatenate(parts, 0)[:B])
        hf = synth.z_tokens(self.E_np, b0, H, W, 2903 + 100 * k, image_offset=off)
        hc = synth.z_tokens(self.E_np, b0, H // 2, W // 2, 2913 + 100 * k, image_offset=off)
        if self.conv_q is not None:                          # pre-images under the conv (see __init__); float64, on the GPU (set-up)
            import torch
            q64 = torch.from_numpy(self.conv_q.astype(np.float64)).to(self.dev)
            b64 = torch.from_numpy(self.conv_b.astype(np.float64)).to(self.dev)

            def pre(t):
                t64 = torch.from_numpy(t).to(self.dev).double() - b64[None, :, None, None]
                return np.ascontiguousarray(torch.einsum("oc,bohw->bchw", q64, t64).float().cpu().numpy())
            hf, hc = pre(hf), pre(hc)
        hf, hc = tile(hf), tile(hc)
        ent = tile(synth.entropy_map(5903 + 100 * k, b0, H // 2, W // 2, image_offset=off))
        return hf, hc, ent

    def new_slot(self, k=0):
        import torch
        o = Slot()
        B, H, W, dev, K = self.B, self.H, self.W, self.dev, self.K
        o.k = k
        hf, hc, ent = self.inputs_np(k)
        o.h_fine, o.h_coarse, o.ent = (torch.from_numpy(x).to(dev) for x in (hf, hc, ent))
        o.h_dual = torch.empty_like(o.h_fine) if self.a.path == "select" else None
        o.grain = torch.empty((B, H // 2, W // 2), dtype=torch.int64, device=dev)
        o.cmask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        o.zq = torch.empty_like(o.h_fine) if self.a.path not in ("tokens", "tokens_model", "tokens_fold") else None
        o.codes = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        o.loss = torch.empty(2, dtype=torch.float32, device=dev)
        o.gate = torch.empty((B, H // 2, W // 2, 2), dtype=torch.int64, device=dev)
        o.h_full = torch.empty_like(o.h_fine) if self.a.path == "model2" else None       # the conv's output (two-kernel form)
        if self.a.path in ("tokens", "tokens_model", "tokens_fold"):
            Lc, Lf = self.permuter.max_lengths()
            o.seq = [torch.empty((B, Lc), dtype=torch.int64, device=dev) for _ in range(3)] + \
                    [torch.empty((B, Lf), dtype=torch.int64, device=dev) for _ in range(3)]
        return o

    def describe(self):
        tail = {"routed": "entropy gate + dual routing + VectorQuantize2 assign as one op (quant_conv not in the path)",
                "select": "entropy gate + route-select kernel + VectorQuantize2 assign (quant_conv not in the path)",
                "model": "entropy gate + dual routing + 1x1 quant_conv + VectorQuantize2 assign as ONE op (the conv is the "
                         "prologue of the assign's pass 1): the order a reference checkpoint runs",
                "model2": "entropy gate + dual routing + 1x1 quant_conv (one kernel writing h) + VectorQuantize2 assign: the "
                          "order a reference checkpoint runs, as two ops",
                "tokens": "entropy gate + dual routing + VectorQuantize2 assign (codes only) + DualGrainSeperatePermuter: "
                          "the tokenisation stage 2 consumes",
                "tokens_model": "entropy gate + dual routing + 1x1 quant_conv + VectorQuantize2 assign (ONE op, codes only) + "
                                "DualGrainSeperatePermuter: the tokenisation stage 2 runs on a stage-1 checkpoint",
                "tokens_fold": "entropy gate + dual routing + VectorQuantize2 assign with the 1x1 quant_conv FOLDED into the "
                               "codebook (ONE op, codes only; no conv for decided tokens) + DualGrainSeperatePermuter: the "
                               "tokenisation stage 2 runs on a stage-1 checkpoint, opt-in fold=True form",
                "model_fold": "entropy gate + dual routing + VectorQuantize2 assign with the 1x1 quant_conv FOLDED into the "
                              "codebook (ONE op: codes + z_q = codebook[code], no loss): loss-free inference in the model order"}[self.a.path]
        return "BASELINE configs[2]: dqvae-entropy-dual-r05, B=%d per GPU, 32x32x256 latents, K=%d, %s" % (self.B, self.K, tail)

    def step(self, o, ev=None):
        from dynamicvectorquantization_amd import qconv
        from dynamicvectorquantization_amd.quantize import vq_assign, vq_assign_routed_dual
        from dynamicvectorquantization_amd.router import route_select_dual_entropy
        path = self.a.path
        if ev:
            ev[0].record()
        if path == "select":
            route_select_dual_entropy(o.ent, THR_R05, o.h_coarse, o.h_fine, out=(o.h_dual, o.grain, o.cmask, o.gate))
            vq_assign(o.h_dual, self.E, self.prep, o.cmask, beta=0.25, mode=self.mode, out=(o.zq, o.codes, o.loss))
        elif path == "model":
            vq_assign_routed_dual(o.h_coarse, o.h_fine, self.E, self.prep, entropy=o.ent, threshold=THR_R05,
                                  beta=0.25, mode=self.mode, out=(o.zq, o.codes, o.loss, o.grain, o.cmask, o.gate),
                                  conv=self.conv)
        elif path == "model2":
            qconv.quant_conv_select(self.conv, o.h_coarse, o.h_fine, entropy=o.ent, threshold=THR_R05,
                                    out=(o.h_full, o.grain, o.cmask, o.gate))
            vq_assign(o.h_full, self.E, self.prep, o.cmask, beta=0.25, mode=self.mode, out=(o.zq, o.codes, o.loss))
        elif path in ("tokens", "tokens_model", "tokens_fold"):
            vq_assign_routed_dual(o.h_coarse, o.h_fine, self.E, self.prep, entropy=o.ent, threshold=THR_R05, beta=0.25,
                                  mode=self.mode, out=(None, o.codes, None, o.grain, o.cmask, o.gate), conv=self.conv,
                                  fold=(path == "tokens_fold"))
            self.permuter(o.codes, o.grain, max_len=self.permuter.max_lengths(), out=o.seq)
        elif path == "model_fold":
            vq_assign_routed_dual(o.h_coarse, o.h_fine, self.E, self.prep, entropy=o.ent, threshold=THR_R05,
                                  mode=self.mode, out=(o.zq, o.codes, None, o.grain, o.cmask, o.gate), conv=self.conv, fold=True)
        else:
            vq_assign_routed_dual(o.h_coarse, o.h_fine, self.E, self.prep, entropy=o.ent, threshold=THR_R05,
                                  beta=0.25, mode=self.mode, out=(o.zq, o.codes, o.loss, o.grain, o.cmask, o.gate))
        if ev:
            ev[1].record()
        return o.codes, o.grain, o.loss

    def dominant(self, o, ev):
        """the dominant kernel alone (pass 1 of the assign), same launch geometry"""
        from dynamicvectorquantization_amd import _lib
        from dynamicvectorquantization_amd.quantize import vq_assign, vq_assign_routed_dual
        path = self.a.path
        if self.a.mode == "exact":
            if ev: ev[0].record()
            vq_assign(o.h_dual if o.h_dual is not None else o.h_fine, self.E, self.prep_dom, o.cmask,
                      beta=0.25, mode=_lib.MODE_EXACT, out=(o.zq, o.codes, o.loss))
            if ev: ev[1].record()
        elif path in ("select", "model2"):
            src = o.h_dual if path == "select" else o.h_full
            if ev: ev[0].record()
            vq_assign(src, self.E, self.prep_dom, o.cmask, beta=0.25, mode=_lib.MODE_FILTER_PASS1,
                      out=(o.zq, o.codes, None))
            if ev: ev[1].record()
        elif path in ("model", "tokens_model", "tokens_fold", "model_fold"):
            if ev: ev[0].record()
            vq_assign_routed_dual(o.h_coarse, o.h_fine, self.E, self.prep_dom, entropy=o.ent,
                                  threshold=THR_R05, beta=0.25, mode=_lib.MODE_FILTER_PASS1,
                                  out=(o.zq, o.codes, None, o.grain, o.cmask, o.gate), conv=self.conv,
                                  fold=path in ("tokens_fold", "model_fold"))
            if ev: ev[1].record()
        else:
            if ev: ev[0].record()
            vq_assign_routed_dual(o.h_coarse, o.h_fine, self.E, self.prep_dom, entropy=o.ent,
                                  threshold=THR_R05, beta=0.25, mode=_lib.MODE_FILTER_PASS1,
                                  out=(o.zq, o.codes, None, o.grain, o.cmask, o.gate))
            if ev: ev[1].record()

    def dominant_tokens(self):
        """tokens one launch of `dominant` processes"""
        return self.B * self.H * self.W

    def dominant_kernel_name(self):
        if self.a.mode == "exact":
            return "vq_assign_exact_kernel<256>"
        if self.a.path in ("select", "model2"):
            return "vq_assign_filter_kernel<256, 0, false, false> (dense pass 1)"
        if self.a.path in ("model", "tokens_model"):
            return "vq_assign_filter_kernel<256, 1, true, false> (pass 1 with the router select and the 1x1 quant_conv fused in)"
        if self.a.path in ("tokens_fold", "model_fold"):
            return ("vq_assign_filter_kernel<256, 2, false, true> (pass 1 on the conv-folded codebook E W, router select fused in, "
                    "coarse branch staged through LDS)")
        return "vq_assign_filter_kernel<256, 2, false, false> (pass 1, router select fused in, coarse branch staged through LDS)"

    def parity(self, slot):
        """the step's outputs, still in HBM, against the oracle on ALL images of this rank (the slot's own inputs)"""
        import numpy as np

        from oracle import oracle
        oracle.build()
        hf, hc, ent = self.inputs_np(slot.k)
        assert np.array_equal(hf, slot.h_fine.cpu().numpy())
        t0 = time.perf_counter()
        og = oracle.entropy_gate(ent, THR_R05)
        osel = oracle.route_select_dual(og, hc, hf)
        res = {"images_checked": int(self.B),
               "grain_mismatches": int((slot.grain.cpu().numpy() != osel["indices"]).sum()),
               "mask_mismatches": int((slot.cmask.cpu().numpy() != osel["codebook_mask"]).sum()),
               "gate_mismatches": int((slot.gate.cpu().numpy() != og).sum())}
        if self.a.path in ("tokens_fold", "model_fold"):
            # the fold's contract: codes = the reference chain on h = dvq_qconv_f32(x) for every token (decided ones by the bound,
            # the others because resolver / list kernel compute that h); h itself within 1e-5 * sum |w||x| of the float64 conv
            import torch

            from dynamicvectorquantization_amd import qconv
            h = qconv.quant_conv_select(self.conv, slot.h_coarse, slot.h_fine, entropy=slot.ent, threshold=THR_R05)["h"].cpu().numpy()
            w64 = self.conv.weight.detach().double().cpu().numpy()[:, :, 0, 0]
            x64 = osel["h_dual"][:8].astype(np.float64)
            ref = np.einsum("ok,bkhw->bohw", w64, x64) + self.conv.bias.detach().double().cpu().numpy()[None, :, None, None]
            bound = np.einsum("ok,bkhw->bohw", np.abs(w64), np.abs(x64))
            res["h_max_err_over_bound"] = float((np.abs(h[:8] - ref) / (1e-5 * bound + 1e-30)).max())
            res["h_bound_mismatches"] = int(not res["h_max_err_over_bound"] <= 1.0)
            o = oracle.vq_assign_nchw(h, self.E_np, osel["codebook_mask"])
            ref_fp64 = oracle.vq_assign_nchw(ref.astype(np.float32), self.E_np, osel["codebook_mask"][:8])
            res["codes_match_rate_vs_fp64_conv"] = float((slot.codes.cpu().numpy()[:8].reshape(8, -1) == ref_fp64["codes"]).mean())
            res["loss_rel_err"] = 0.0
            q, ls = self.prep.fallback_count()
            res["tokens_resolved_with_a_conv"] = int(q + ls)
            if slot.zq is not None:                              # z_q := codebook[code]: within 1e-6 of the reference's z + (e - z)
                e = np.moveaxis(self.E_np[o["codes"]], 2, 1).reshape(slot.zq.shape)
                res["zq_mismatches"] = int((np.abs(slot.zq.cpu().numpy() - e) > 1e-6 * np.maximum(1.0, np.abs(e))).sum())
        elif self.a.path in ("model", "model2", "tokens_model"):
            # h is a tolerance-level quantity (1e-5 * sum |w||x| vs the conv in float64, checked on 8 images); codes, z_q and
            # the loss are exact GIVEN the h the kernels scored.  Two-kernel form: h is the conv kernel's output tensor.  One-op
            # form: h exists only inside pass 1 -- the op is run once more with an h_buf, which makes it write the conv output
            # of every token beside the same codes / z_q / loss (checked to be the same bits as the timed step's).
            import torch

            from dynamicvectorquantization_amd.quantize import vq_assign_routed_dual
            if self.a.path in ("model", "tokens_model"):
                hb = torch.empty_like(slot.h_fine)
                chk = [torch.empty_like(slot.h_fine), torch.empty_like(slot.codes), torch.empty_like(slot.loss),
                       torch.empty_like(slot.grain), torch.empty_like(slot.cmask), torch.empty_like(slot.gate)]
                vq_assign_routed_dual(slot.h_coarse, slot.h_fine, self.E, self.prep, entropy=slot.ent, threshold=THR_R05,
                                      beta=0.25, mode=self.mode, out=tuple(chk), conv=self.conv, h_buf=hb)
                torch.cuda.synchronize()
                same = torch.equal(chk[1], slot.codes)
                if slot.zq is not None:
                    same = same and torch.equal(chk[0], slot.zq) and torch.equal(chk[2], slot.loss)
                res["rerun_with_h_buf_mismatches"] = int(not same)
                h = hb.cpu().numpy()
                del hb, chk
            else:
                h = slot.h_full.cpu().numpy()
            w64 = self.conv.weight.detach().double().cpu().numpy()[:, :, 0, 0]
            x64 = osel["h_dual"][:8].astype(np.float64)
            ref = np.einsum("ok,bkhw->bohw", w64, x64) + self.conv.bias.detach().double().cpu().numpy()[None, :, None, None]
            bound = np.einsum("ok,bkhw->bohw", np.abs(w64), np.abs(x64))
            res["h_max_err_over_bound"] = float((np.abs(h[:8] - ref) / (1e-5 * bound + 1e-30)).max())
            res["h_bound_mismatches"] = int(not res["h_max_err_over_bound"] <= 1.0)
            o = oracle.vq_assign_nchw(h, self.E_np, osel["codebook_mask"])
            ref_fp64 = oracle.vq_assign_nchw(ref.astype(np.float32), self.E_np, osel["codebook_mask"][:8])
            res["codes_match_rate_vs_fp64_conv"] = float((slot.codes.cpu().numpy()[:8].reshape(8, -1) == ref_fp64["codes"]).mean())
            ol = float(oracle.vq_loss(o["sqerr"], o["numel"], 0.25))
            res["loss_rel_err"] = abs(float(slot.loss[1]) - ol) / abs(ol) if self.a.path != "tokens_model" else 0.0
        else:
            o = oracle.vq_assign_nchw(osel["h_dual"], self.E_np, osel["codebook_mask"])
        self.oracle_seconds = time.perf_counter() - t0          # one cold pass of the CPU port over the FULL batch
        codes = slot.codes.cpu().numpy().reshape(self.B, -1)
        res["code_mismatches"] = int((codes != o["codes"]).sum())
        if slot.zq is not None and self.a.path != "model_fold":
            res["zq_mismatches"] = int((slot.zq.cpu().numpy() != o["zq"]).sum())
        if self.a.path in ("tokens", "tokens_model", "tokens_fold"):
            from oracle import permuter as operm
            ref = operm.forward(o["codes"].reshape(self.B, self.H, self.W), osel["indices"], coarse_hw=self.H // 2,
                                fine_hw=self.H, content_pad=self.K, content_eos=self.K + 1)
            names = ["coarse_content", "coarse_position", "coarse_segment", "fine_content", "fine_position", "fine_segment"]
            pads = {"coarse_content": self.K, "fine_content": self.K, "coarse_position": 256, "fine_position": 1024,
                    "coarse_segment": 0, "fine_segment": 1}
            bad = 0
            for nme, tns in zip(names, slot.seq):
                got, want = tns.cpu().numpy(), ref[nme]
                L = want.shape[1]
                bad += int((got[:, :L] != want).sum()) + int((got[:, L:] != pads[nme]).sum())   # beyond the batch maximum: PAD
            res["token_stream_mismatches"] = bad
            res["loss_rel_err"] = 0.0
        elif self.a.path not in ("model", "model2", "model_fold"):
            ol = float(oracle.vq_loss(o["sqerr"], o["numel"], 0.25))
            res["loss_rel_err"] = abs(float(slot.loss[1]) - ol) / abs(ol)
        return res


class StrongTriple:
    """BASELINE configs[3]: triple granularity, global batch split over the ranks: fused feature-router gate
    + triple routing + assign"""
    name = "triple"

    def __init__(self, a, rank, world, dev):
        import torch

        from dynamicvectorquantization_amd import _lib, synth
        from dynamicvectorquantization_amd.encode import shard_slice
        from dynamicvectorquantization_amd.quantize import _CodebookPrep
        from dynamicvectorquantization_amd.router import TripleGrainFeatureRouter
        self.a, self.dev, self.world = a, dev, world
        self.Bglobal = a.batch or 1024
        s, e = shard_slice(self.Bglobal, rank, world)
        B = self.B = e - s
        K, D, H, W = a.codes, 256, 32, 32
        self.K, self.D, self.H, self.W = K, D, H, W
        self.mode = _lib.MODE_EXACT if a.mode == "exact" else _lib.MODE_FILTER
        self.E_np = synth.codebook_trained(K, D)
        self.s0 = s
        self.E = torch.from_numpy(self.E_np).to(dev)
        self.router = TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu")
        sd = {k: torch.from_numpy(synth.seeded_param(6104, i, k, tuple(v.shape)))
              for i, (k, v) in enumerate(self.router.state_dict().items())}
        self.router.load_state_dict(sd)
        self.router = self.router.to(dev).eval()
        self.prep = _CodebookPrep()
        self.slots = [self.new_slot(k) for k in range(a.streams)]

    def inputs_np(self, k):
        import numpy as np

        from dynamicvectorquantization_amd import synth
        B = self.B
        b0 = min(B, 128)

        def tile(base):
            if B <= b0:
                return base[:B]
            return np.ascontiguousarray(np.concatenate([np.roll(base, 5 * j, axis=-1) for j in range((B + b0 - 1) // b0)], 0)[:B])
        return (tile(synth.z_tokens(self.E_np, b0, 32, 32, 2104 + 100 * k, image_offset=self.s0)),
                tile(synth.z_tokens(self.E_np, b0, 16, 16, 2114 + 100 * k, image_offset=self.s0)),
                tile(synth.z_tokens(self.E_np, b0, 8, 8, 2124 + 100 * k, image_offset=self.s0)))

    def new_slot(self, k=0):
        import torch
        o = Slot()
        B, H, W, dev = self.B, self.H, self.W, self.dev
        o.k = k
        o.h_fine, o.h_median, o.h_coarse = (torch.from_numpy(x).to(dev) for x in self.inputs_np(k))
        o.h_triple = torch.empty_like(o.h_fine) if self.a.path == "select" else None
        o.grain = torch.empty((B, 8, 8), dtype=torch.int64, device=dev)
        o.cmask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        o.zq = torch.empty_like(o.h_fine)
        o.codes = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        o.loss = torch.empty(2, dtype=torch.float32, device=dev)
        o.logits = None
        return o

    def describe(self):
        assert self.a.path in ("routed", "select"), "--scaling strong runs --path routed or select"
        return ("BASELINE configs[3]: triple granularity F=32/16/8, global B=%d split image-parallel (%d on this "
                "rank), 32x32x256 latents, K=%d, fused feature-router gate + triple routing + VectorQuantize2 assign "
                "(quant_conv not in the path)" % (self.Bglobal, self.B, self.K))

    def step(self, o, ev=None):
        import torch

        from dynamicvectorquantization_amd.quantize import vq_assign, vq_assign_routed_triple
        from dynamicvectorquantization_amd.router import route_select_triple
        with torch.no_grad():
            o.logits = self.router(h_fine=o.h_fine, h_median=o.h_median, h_coarse=o.h_coarse)
            if ev:
                ev[0].record()
            if self.a.path == "select":
                route_select_triple(o.logits, o.h_coarse, o.h_median, o.h_fine,
                                    out=(o.h_triple, o.grain, o.cmask))
                vq_assign(o.h_triple, self.E, self.prep, o.cmask, beta=0.25, mode=self.mode,
                          out=(o.zq, o.codes, o.loss))
            else:
                vq_assign_routed_triple(o.h_coarse, o.h_median, o.h_fine, self.E, self.prep, o.logits,
                                        beta=0.25, mode=self.mode,
                                        out=(o.zq, o.codes, o.loss, o.grain, o.cmask))
        if ev:
            ev[1].record()
        return o.codes, o.grain, o.loss

    def dominant(self, o, ev):
        from dynamicvectorquantization_amd import _lib
        from dynamicvectorquantization_amd.quantize import vq_assign, vq_assign_routed_triple
        if ev: ev[0].record()
        if self.a.path == "select" or self.a.mode == "exact":
            vq_assign(o.h_triple if o.h_triple is not None else o.h_fine, self.E, self.prep_dom, o.cmask,
                      beta=0.25, mode=_lib.MODE_EXACT if self.a.mode == "exact" else _lib.MODE_FILTER_PASS1,
                      out=(o.zq, o.codes, o.loss if self.a.mode == "exact" else None))
        else:
            vq_assign_routed_triple(o.h_coarse, o.h_median, o.h_fine, self.E, self.prep_dom, o.logits,
                                    beta=0.25, mode=_lib.MODE_FILTER_PASS1,
                                    out=(o.zq, o.codes, None, o.grain, o.cmask))
        if ev: ev[1].record()

    def dominant_tokens(self):
        return self.B * self.H * self.W

    def dominant_kernel_name(self):
        if self.a.mode == "exact":
            return "vq_assign_exact_kernel<256>"
        return "vq_assign_filter_kernel<256, 0, false, false> (dense pass 1)" if self.a.path == "select" else \
            "vq_assign_filter_kernel<256, 2, false, false> (pass 1, router select fused in, median / coarse branches staged through LDS)"

    def parity(self, slot):
        """select + assign against the oracle GIVEN the logits the GPU router produced (the feature router
        itself is a 1e-4 tolerance kernel, covered by tests/)"""
        from oracle import oracle
        oracle.build()
        lg = slot.logits.cpu().numpy()
        hf, hm, hc = self.inputs_np(slot.k)
        osel = oracle.route_select_triple(lg, hc, hm, hf)
        o = oracle.vq_assign_nchw(osel["h_triple"], self.E_np, osel["codebook_mask"])
        codes = slot.codes.cpu().numpy().reshape(self.B, -1)
        ol = float(oracle.vq_loss(o["sqerr"], o["numel"], 0.25))
        return {"images_checked": int(self.B), "code_mismatches": int((codes != o["codes"]).sum()),
                "zq_mismatches": int((slot.zq.cpu().numpy() != o["zq"]).sum()),
                "grain_mismatches": int((slot.grain.cpu().numpy() != osel["indices"]).sum()),
                "mask_mismatches": int((slot.cmask.cpu().numpy() != osel["codebook_mask"]).sum()),
                "loss_rel_err": abs(float(slot.loss[1]) - ol) / abs(ol)}



# ------------------------------------------------------------------------------------------------
# every BASELINE.json config in the reference's own op order (rank 0, N = 1, default command)
# ------------------------------------------------------------------------------------------------
def _ev_ms(fn, n, warm=5):
    """mean milliseconds of fn() over n back-to-back calls on the current stream (HIP events)"""
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _pass1_ms(run, prep, n=60):
    """pass 1 of a filter-path op ALONE: `run()` issues the op in MODE_FILTER_PASS1 through `prep` (a _CodebookPrep of its
    own); the profiling mode leaves the queue counters dirty, so they are zeroed in FRONT of every bracket and the workspace is
    declared clean -- the events enclose the pass-1 kernel and nothing else"""
    import numpy as np
    import torch
    evs = []
    for i in range(-3, n):
        lw = getattr(prep, "_last_ws", None)
        if lw is not None and hasattr(lw[1], "clean"):
            lw[1].t[:min(lw[1].t.numel(), 4 << 20)].zero_()
            lw[1].clean = True
        e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e[0].record()
        run()
        e[1].record()
        if i >= 0:
            evs.append(e)
    torch.cuda.synchronize()
    return float(np.mean([s.elapsed_time(e) for s, e in evs]))


def configs_block(dev, small=False):
    """BASELINE.json's five configs, each in the op order the reference's `encode` runs after the CNN trunk
    (models/stage1/vqgan.py:68-72; dqvae_dual_feat.py:59-68; dqvae_dual_entropy.py:124-134 incl. the Entropy module on the
    PIXELS; dqvae_triple_feat.py:68-77), through the drop-in modules, strictly serial on one stream: `ms` = HIP events over
    back-to-back module calls; `stage_ms` = the same for each stage alone (+ pass 1 of the assign alone); `frac` = SURVEY 8d
    algorithmic bytes of the stage / its time / 8 TB/s (K = 16384: flops against the matrix peak); `share` = stage / step;
    parity on a stated sample of images against the oracle (codes / z_q / grain / mask bit-exact GIVEN the conv output h the op
    scored, h within 1e-5 sum|w||x| of the float64 conv, entropy map within 1e-5, loss 1e-5 when the sample is the batch).
    `small`: reduced batches (tests)."""
    import numpy as np
    import torch

    from dynamicvectorquantization_amd import _lib, qconv, synth
    from dynamicvectorquantization_amd.encode import encode_dual, encode_fixed, encode_triple
    from dynamicvectorquantization_amd.entropy import Entropy
    from dynamicvectorquantization_amd.quantize import (VectorQuantize2, VectorQuantizer2, _CodebookPrep, vq_assign,
                                                        vq_assign_routed_dual, vq_assign_routed_triple)
    from dynamicvectorquantization_amd.router import (DualGrainFeatureRouter, DualGrainFixedEntropyRouter,
                                                      TripleGrainFeatureRouter)
    from oracle import oracle
    from oracle.entropy_torch import entropy_map as entropy_ref
    oracle.build()
    t_start = time.perf_counter()
    K, D = 1024, 256
    nsteps = 20 if small else 200
    E_np = synth.codebook_trained(K, D)
    E = torch.from_numpy(E_np).to(dev)
    # the models' quant_conv (orthogonal matrix + bias) and branch inputs = pre-images of z_tokens under it: the QUANTIZER sees
    # the headline distribution, as in WeakDual (see there)
    q, _ = np.linalg.qr(synth.normal(6012, (D, D), 0.0, 1.0).astype(np.float64))
    cb = synth.normal(6013, (D,), 0.0, 0.1)
    conv = torch.nn.Conv2d(D, D, 1).to(dev).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(q.astype(np.float32).reshape(D, D, 1, 1)).to(dev))
        conv.bias.copy_(torch.from_numpy(cb).to(dev))
    q64 = conv.weight.detach().double().reshape(D, D)
    b64 = conv.bias.detach().double()

    def pre(t_np, B):
        t = tile_images(torch.from_numpy(t_np).to(dev), B).double() - b64[None, :, None, None]
        return torch.einsum("oc,bohw->bchw", q64, t).float().contiguous()

    nb = 8 if small else 32
    base32 = synth.z_tokens(E_np, nb, 32, 32, 2301)
    base16 = synth.z_tokens(E_np, nb, 16, 16, 2311)
    base8 = synth.z_tokens(E_np, nb, 8, 8, 2321)
    w64 = q64.cpu().numpy()
    bias64 = b64.cpu().numpy()

    def vq2():
        m = VectorQuantize2(K, D).to(dev).eval()
        with torch.no_grad():
            m.codebook.weight[:-1].copy_(E)
        m.invalidate_codebook_cache()
        return m

    def seeded(router, seed):
        sd = {k: torch.from_numpy(synth.seeded_param(seed, i, k, tuple(v.shape))) for i, (k, v) in enumerate(router.state_dict().items())}
        router.load_state_dict(sd)
        return router.to(dev).eval()

    def check_h(h, x_sel, n):
        """largest |h - conv64(x)| / (1e-5 sum |w||x|) over the first n images"""
        x64 = x_sel[:n].astype(np.float64)
        ref = np.einsum("ok,bkhw->bohw", w64, x64) + bias64[None, :, None, None]
        bound = np.einsum("ok,bkhw->bohw", np.abs(w64), np.abs(x64)) + np.abs(bias64)[None, :, None, None]
        return float((np.abs(h[:n] - ref) / (1e-5 * bound + 1e-30)).max())

    def routed(name, B, ns, feats, router=None, images=None, no_conv_too=False):
        """one routed config: [Entropy ->] router gate -> select + quant_conv + assign (one op); feats = (h_coarse, [h_median,] h_fine)"""
        triple = len(feats) == 3
        vq = vq2()
        cbk = vq.codebook
        hc, hf = feats[0], feats[-1]
        hm = feats[1] if triple else None
        ent_mod = Entropy(16, 256, 256).to(dev) if images is not None else None
        with torch.no_grad():
            if images is not None:
                def full():
                    return encode_dual(router, vq, hf, hc, entropy=ent_mod(images), quant_conv=conv)
            elif triple:
                def full():
                    return encode_triple(router, vq, hf, hm, hc, quant_conv=conv)
            else:
                def full():
                    return encode_dual(router, vq, hf, hc, quant_conv=conv)
            res = full()
            torch.cuda.synchronize()
            ms = _ev_ms(full, nsteps)
            stage, nbytes = {}, {}
            if images is not None:
                stage["entropy_map"] = _ev_ms(lambda: ent_mod(images), nsteps)
                nbytes["entropy_map"] = images.numel() * 4 + B * 256 * 4
                ent = ent_mod(images)
                gate_kw = dict(entropy=ent, threshold=router.fine_grain_threshold)
            else:
                if triple:
                    stage["router_gate"] = _ev_ms(lambda: router(h_fine=hf, h_median=hm, h_coarse=hc), nsteps)
                    gate = router(h_fine=hf, h_median=hm, h_coarse=hc)
                else:
                    stage["router_gate"] = _ev_ms(lambda: router(h_fine=hf, h_coarse=hc), nsteps)
                    gate = router(h_fine=hf, h_coarse=hc)
                nbytes["router_gate"] = sum(t.numel() for t in feats) * 4 + sum(p.numel() for p in router.parameters()) * 4 + gate.numel() * 4
                gate_kw = dict(gate=gate)

            def op(mode=_lib.MODE_FILTER, prep=cbk._prep, h_buf=None, cv=conv, out=None, want_loss=True):
                if triple:
                    return vq_assign_routed_triple(hc, hm, hf, cbk.codes, prep, gate_kw["gate"], beta=vq.beta, mode=mode, conv=cv,
                                                   h_buf=h_buf, out=out, want_loss=want_loss)
                return vq_assign_routed_dual(hc, hf, cbk.codes, prep, beta=vq.beta, mode=mode, conv=cv, h_buf=h_buf, out=out,
                                             want_loss=want_loss, **gate_kw)
            stage["assign_op"] = _ev_ms(op, nsteps)
            pdom = _CodebookPrep()
            r0 = op()
            outs = (r0["zq"], r0["codes"], None, r0["indices"], r0["codebook_mask"]) + ((r0["gate"],) if images is not None else ())
            if not triple and images is None:
                outs = outs + (None,)
            stage["pass1"] = _pass1_ms(lambda: op(mode=_lib.MODE_FILTER_PASS1, prep=pdom, out=outs, want_loss=False), pdom,
                                       n=20 if small else 60)
            ntok = B * 1024
            nbytes["pass1"] = ntok * 2060 + K * D * 4
            ent_d = {"workload": name, "B": B, "ms": ms, "images_per_s": B / (ms * 1e-3), "stage_ms": stage,
                     "frac": {k: nbytes[k] / (stage[k] * 1e-3) / 1e9 / HBM_PEAK_GBS for k in nbytes},
                     "share": {k: stage[k] / ms for k in stage if k != "pass1"}}
            if no_conv_too:
                ent_d["ms_no_conv"] = _ev_ms(lambda: encode_dual(router, vq, hf, hc, entropy=ent_mod(images)), nsteps)
            # parity: the op once more with a full h_buf (it then writes the conv output of EVERY token beside the same codes)
            hb = torch.empty_like(hf)
            r1 = op(h_buf=hb)
            torch.cuda.synchronize()
            quant, emb_loss, info, grain, gate_out = res
            same = (torch.equal(r1["codes"], info[2]) and torch.equal(r1["zq"], quant) and torch.equal(r1["indices"], grain)
                    and torch.equal(r1["loss"][1], emb_loss))
            h = hb[:ns].cpu().numpy()
            if images is not None:
                eref = entropy_ref(images[:min(ns, 8)].cpu()).numpy()
                ent_d["entropy_max_abs_err"] = float(np.abs(ent[:min(ns, 8)].cpu().numpy() - eref).max())
                og = oracle.entropy_gate(ent[:ns].cpu().numpy(), router.fine_grain_threshold)
                osel = oracle.route_select_dual(og, hc[:ns].cpu().numpy(), hf[:ns].cpu().numpy())
                xsel = osel["h_dual"]
            elif triple:
                osel = oracle.route_select_triple(gate[:ns].cpu().numpy(), hc[:ns].cpu().numpy(), hm[:ns].cpu().numpy(), hf[:ns].cpu().numpy())
                xsel = osel["h_triple"]
            else:
                osel = oracle.route_select_dual(gate[:ns].cpu().numpy(), hc[:ns].cpu().numpy(), hf[:ns].cpu().numpy())
                xsel = osel["h_dual"]
            o = oracle.vq_assign_nchw(h, E_np, osel["codebook_mask"])
            mism = {"codes": int((r1["codes"][:ns].cpu().numpy().reshape(ns, -1) != o["codes"]).sum()),
                    "zq": int((r1["zq"][:ns].cpu().numpy() != o["zq"]).sum()),
                    "grain": int((grain[:ns].cpu().numpy() != osel["indices"]).sum()),
                    "mask": int((r1["codebook_mask"][:ns].cpu().numpy() != osel["codebook_mask"]).sum()),
                    "rerun_with_h_buf": int(not same)}
            ent_d.update({"checked_images": ns, "mismatches": mism, "code_mismatches": mism["codes"],
                          "h_err_over_bound": check_h(h, xsel, min(ns, 8)),
                          "fine_ratio": float((grain != 0).float().mean()) if not triple else None})
            if ns == B:
                ol = float(oracle.vq_loss(o["sqerr"], o["numel"], 0.25))
                ent_d["loss_rel_err"] = abs(float(emb_loss) - ol) / abs(ol)
        del hb, r1, r0, res
        return ent_d

    out = {}
    # configs[0]: VQModel.encode (models/stage1/vqgan.py:68-72): quant_conv -> VectorQuantizer2 (beta .25, legacy False), B = 4, 16 x 16
    with torch.no_grad():
        B0 = 4
        vqg = VectorQuantizer2(K, D, beta=0.25, legacy=False).to(dev).eval()
        vqg.embedding.weight.copy_(E)
        vqg.invalidate_codebook_cache()
        x0 = pre(base16, B0)
        f0 = lambda: encode_fixed(vqg, x0, quant_conv=conv)
        zq0, loss0, info0 = f0()
        ms0 = _ev_ms(f0, nsteps)
        st0 = {"quant_conv": _ev_ms(lambda: qconv.quant_conv(conv, x0), nsteps)}
        h0 = qconv.quant_conv(conv, x0)
        st0["assign_op"] = _ev_ms(lambda: vqg(h0), nsteps)
        o0 = oracle.vq_assign_nchw(h0.cpu().numpy(), E_np, None)
        ol0 = float(oracle.vq_loss(o0["sqerr"], o0["numel"], 0.25))
        out["cfg0"] = {"workload": "VQModel.encode: quant_conv -> VectorQuantizer2 (one op; stages timed apart), 16x16x256, K=1024", "B": B0,
                       "ms": ms0, "images_per_s": B0 / (ms0 * 1e-3), "stage_ms": st0, "share": {k: v / ms0 for k, v in st0.items()},
                       "frac": {"assign_op": (B0 * 256 * 2056 + K * D * 4) / (st0["assign_op"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                       "checked_images": B0, "code_mismatches": int((info0[2].cpu().numpy().reshape(B0, -1) != o0["codes"]).sum()),
                       "mismatches": {"zq": int((zq0.cpu().numpy() != o0["zq"]).sum())},
                       "h_err_over_bound": check_h(h0.cpu().numpy(), x0.cpu().numpy(), B0),
                       "loss_rel_err": abs(float(loss0) - ol0) / abs(ol0)}
    # configs[1]: DualGrainVQModel.encode with the feature router (dqvae_dual_feat.py:59-68), B = 64
    B1 = 8 if small else 64
    r2 = seeded(DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu"), 6102)
    out["cfg1"] = routed("dual feature router: gate -> select+quant_conv+assign (one op)", B1, B1,
                         (pre(base16, B1), pre(base32, B1)), router=r2)
    # configs[2]: DualGrainVQModel.encode with the entropy router (dqvae_dual_entropy.py:124-134): PIXELS -> Entropy -> gate -> ...
    B2 = 16 if small else 256
    jpath = os.path.join(ROOT, "tests", "golden", "entropy_thresholds_imagenet_train_patch-16.json")
    rent = DualGrainFixedEntropyRouter(jpath, 0.5)
    nimg = min(B2, 32)
    ibase = torch.from_numpy(synth.images_flat_noise(5000, nimg)[0]).to(dev)     # (tiled by whole patches: the flat / noise mixture stays what it is)
    imgs = torch.cat([torch.roll(ibase, 16 * k, -1) for k in range((B2 + nimg - 1) // nimg)], 0)[:B2].contiguous()
    del ibase
    out["cfg2"] = routed("dual entropy router r05: pixels -> entropy map -> gate+select+quant_conv+assign (one op)",
                         B2, min(B2, 64), (pre(base16, B2), pre(base32, B2)), router=rent, images=imgs, no_conv_too=True)
    del imgs
    # configs[3]: TripleGrainVQModel.encode (dqvae_triple_feat.py:68-77): the whole batch on one GPU, and one rank's share of 8
    r3 = seeded(TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu"), 6104)
    for key, B3 in (("cfg3", 16 if small else 1024), ("cfg3_per_rank_of_8", 8 if small else 128)):
        out[key] = routed("triple feature router: gate -> select+quant_conv+assign (one op)", B3, min(B3, 32),
                          (pre(base8, B3), pre(base16, B3), pre(base32, B3)), router=r3)
        torch.cuda.empty_cache()
    # configs[4]: K = 16384 stress, dense VectorQuantize2 assign: fp16-MFMA filter path vs the exact fp32-MFMA chain
    with torch.no_grad():
        K4, B4 = 16384, (4 if small else 512)
        E4_np = synth.codebook_trained(K4, D)
        E4 = torch.from_numpy(E4_np).to(dev)
        z4 = tile_images(torch.from_numpy(synth.z_tokens(E4_np, 4 if small else 16, 32, 32, 2005)).to(dev), B4)
        pf, pe = _CodebookPrep(), _CodebookPrep()
        o4 = (torch.empty_like(z4), torch.empty((B4, 32, 32), dtype=torch.int64, device=dev), torch.empty(2, device=dev))
        tf = _ev_ms(lambda: vq_assign(z4, E4, pf, None, mode=_lib.MODE_FILTER, out=o4), 5 if small else 30, warm=2)
        zq_f, c_f = o4[0].clone(), o4[1].clone()
        te = _ev_ms(lambda: vq_assign(z4, E4, pe, None, mode=_lib.MODE_EXACT, out=o4), 2 if small else 3, warm=1)
        ns4 = min(B4, 8)
        oo = oracle.vq_assign_nchw(z4[:ns4].cpu().numpy(), E4_np, None)
        fl = 2.0 * K4 * D * B4 * 1024
        out["cfg4"] = {"workload": "K=16384 dense assign: fp16-MFMA filter + exact resolution vs all-exact fp32-MFMA chain", "B": B4, "ms": tf, "images_per_s": B4 / (tf * 1e-3), "ms_exact_mode": te,
                       "tflops_equiv": fl / (tf * 1e-3) / 1e12, "tflops_exact_mode": fl / (te * 1e-3) / 1e12,
                       "frac": {"filter_vs_fp16_mfma_2500": fl / (tf * 1e-3) / 1e12 / 2500.0,
                                "exact_vs_fp32_mfma_157": fl / (te * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                                "filter_hbm": (B4 * 1024 * 2056 + K4 * D * 4) / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS},
                       "modes_bit_identical": bool(torch.equal(c_f, o4[1]) and torch.equal(zq_f, o4[0])),
                       "checked_images": ns4, "code_mismatches": int((c_f[:ns4].cpu().numpy().reshape(ns4, -1) != oo["codes"]).sum()),
                       "mismatches": {"zq": int((zq_f[:ns4].cpu().numpy() != oo["zq"]).sum())}}
    torch.cuda.empty_cache()
    # SURVEY 8f rows that no config's encode reaches: the permuter (f1) and the training-mode quantizer (f2), timed at configs[2]'s size
    try:
        from dynamicvectorquantization_amd.permuter import DualGrainSeperatePermuter
        Bn = 16 if small else 256
        perm = DualGrainSeperatePermuter(coarse_hw=16, fine_hw=32, content_pad_code=K, content_eos_code=K + 1)
        codes_p = torch.randint(0, K, (Bn, 16, 16), device=dev).repeat_interleave(2, 1).repeat_interleave(2, 2).contiguous()
        grain_p = torch.from_numpy(synth.grain_gate_dual(4002, Bn, 16, 16).argmax(-1)).to(dev)
        with torch.no_grad():
            t_perm = _ev_ms(lambda: perm(codes_p, grain_p, max_len=perm.max_lengths()), nsteps)
        vqt = VectorQuantize2(K, D).to(dev).train()
        with torch.no_grad():
            vqt.codebook.weight[:-1].copy_(E)
            vqt.codebook.embed_ema.copy_(E)
            vqt.codebook.cluster_size_ema.fill_(10.0)
        xt = tile_images(torch.from_numpy(base32).to(dev), Bn).requires_grad_(True)
        mt = torch.ones((Bn, 1, 32, 32), device=dev)
        gq = torch.full_like(xt, 1e-6)

        def train_step():
            q, loss, _ = vqt(xt, codebook_mask=mt)
            torch.autograd.backward([q, loss], [gq, torch.ones_like(loss)])
            xt.grad = None
        # median of three blocks: a block that meets an allocator miss or a clock dip reads 15-20 % high (seen once in four runs)
        t_blocks = sorted(_ev_ms(train_step, 10 if small else 30, warm=5) for _ in range(3))
        t_train = t_blocks[1]
        # f1: DualGrainSeperatePermuter.forward (no host sync); f2: VectorQuantize2 in training mode (assign + EMA statistics + restart +
        # codebook update) and its backward
        out["next_rows"] = {"B": Bn, "permuter_forward_ms": t_perm, "train_forward_backward_ms": t_train,
                            "train_forward_backward_ms_blocks": t_blocks}
        del vqt, xt, mt, gq
    except Exception as ex:                                      # side measurement: never fatal
        out["next_rows"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    torch.cuda.empty_cache()

    def rnd(v):
        if isinstance(v, float):
            return float("%.5g" % v)
        if isinstance(v, dict):
            return {k: rnd(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [rnd(x) for x in v]
        return v
    out = rnd(out)
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    out["all_parity_ok"] = all(c.get("code_mismatches", 0) == 0 and not any(c.get("mismatches", {}).values())
                               and c.get("h_err_over_bound", 0.0) <= 1.0 and c.get("loss_rel_err", 0.0) <= 1e-5
                               and c.get("entropy_max_abs_err", 0.0) <= 1e-5 and c.get("modes_bit_identical", True)
                               for k, c in out.items() if isinstance(c, dict))
    return out

# ------------------------------------------------------------------------------------------------
def run_rank(a):
    import numpy as np
    import torch
    import torch.distributed as dist

    t_proc0 = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line (rank 0's JSON): until then file descriptor 1 points at stderr, so that native
    # libraries that print there (RCCL's version banner at the first collective) cannot add lines to it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("DVQ_BENCH_TEST_FAIL_RANK") == str(rank):    # launcher test: this rank dies in set-up, before the rendezvous
        raise RuntimeError("bench.py: rank %d told to fail in set-up (DVQ_BENCH_TEST_FAIL_RANK)" % rank)
    if os.environ.get("DVQ_BENCH_TEST_SLEEP_RANK") == str(rank):   # launcher test: this rank never joins (wall-time bound)
        time.sleep(3600)
    backend = os.environ.get("DVQ_BENCH_BACKEND", "nccl")     # nccl = RCCL over xGMI; gloo only for tests
    if world > 1 and backend != "nccl":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)   # needs no GPU: the rendezvous comes first
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev > 1 and (world > ndev or local >= ndev):
        # one process per GPU (the reference: Lightning DDP, train.py:227-233): RCCL cannot put two ranks on one device, and a
        # silent modulo would report a scaling figure measured on fewer GPUs than claimed.  (ONE visible device per rank is the
        # launch that masks devices per rank -- every rank drives its device 0; should it be a one-GPU box instead, RCCL refuses
        # the duplicate device at init and the identity check below counts the distinct devices.)
        raise SystemExit("bench.py: %d ranks (local rank %d) but %d visible GPU(s): the RCCL run needs one GPU per rank "
                         "(DVQ_BENCH_BACKEND=gloo folds ranks onto the visible devices, for tests only)" % (world, local, ndev))
    local = local % ndev                           # gloo tests only: a 1-GPU box can still exercise the N > 1 code path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_xch = world == 1 and os.environ.get("DVQ_BENCH_FORCE_EXCHANGE") == "1"   # 1-rank RCCL group: exchange overhead probe
    if force_xch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1 and backend == "nccl":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dynamicvectorquantization_amd.encode import CodeExchange, StreamSlots
    from dynamicvectorquantization_amd.quantize import _CodebookPrep

    # pre-flight record: which physical device every rank drives (so a SCALE record can show that RCCL saw N distinct GPUs)
    pr = torch.cuda.get_device_properties(local)
    ident = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": local, "device_name": pr.name,
             "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
             "device_uuid": str(getattr(pr, "uuid", "")) or None, "host": socket.gethostname()}
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    distinct_devices = len({(d["host"], d["device_uuid"] or d["pci_bus_id"]) for d in idents})
    backend_name = (dist.get_backend() if dist.is_initialized() else None)
    if world > 1 and backend == "nccl" and distinct_devices != world:
        raise SystemExit("bench.py: %d ranks drive only %d distinct GPUs: %s" % (world, distinct_devices, idents))

    wl = (WeakDual if a.scaling == "weak" else StrongTriple)(a, rank, world, dev)
    wl.prep_dom = _CodebookPrep()
    B, K, D, H, W = wl.B, wl.K, wl.D, wl.H, wl.W
    # stream slots: step i runs on slot i % S (own stream, outputs, workspace).  Exchange objects (wire / result buffers):
    # max(2, S), so the all-gather of step i is completed when its object comes round again, two or more steps later
    # (a 0.6-MB all-gather is latency-bound on xGMI, and RCCL's kernel competes for CUs with a pass 1 that fills the chip)
    S = max(1, a.streams)
    streams = [torch.cuda.current_stream(dev)] if S == 1 else [sl.stream for sl in StreamSlots(S, dev).slots]
    nx = max(2, S)
    xchs = ([CodeExchange(wl.slots[0].codes, wl.slots[0].grain, K, wl.Bglobal, numel_per_image=H * W * D) for _ in range(nx)]
            if (world > 1 or force_xch) else [])
    nstep = [0]

    def step(i=None):
        n = nstep[0]
        nstep[0] += 1
        with torch.cuda.stream(streams[n % S]):
            codes, grain, loss = wl.step(wl.slots[n % S])
            if xchs:
                x = xchs[n % nx]
                x.finish()                     # this object's previous exchange: stream wait + unpack kernel
                x.start(codes, grain, loss)    # pack kernel + async all-gather

    def fence():
        for j, x in enumerate(xchs):
            with torch.cuda.stream(streams[j % S]):
                x.finish()                 # the last exchanges complete inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    torch.cuda.synchronize()               # codebook image built before a second stream reads it
    setup_s = time.perf_counter() - t_proc0   # imports, rendezvous, synthetic inputs, first step: what a rank needs before it can time
    if world > 1:
        ts = torch.tensor([setup_s], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        setup_s = float(ts.item())
    for _ in range(a.spinup):              # clock / power-state ramp, see --spinup
        step()
    for _ in range(a.warmup):
        step()
    R = max(1, a.repeats)
    dts, t_issue = [], 0.0
    for r in range(R):                            # R blocks of exactly K steps, each fenced on both sides
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i)
        if r == 0:
            t_issue = time.perf_counter() - t0    # host time to queue the K steps (launch-bound if close to dt)
        fence()
        dts.append(time.perf_counter() - t0)
    if world > 1:
        tt = torch.tensor(dts, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)   # per block: the slowest rank
        dts = [float(v) for v in tt.tolist()]
    dt = float(np.median(dts))

    # the exchange alone: pack kernel -> all-gather -> unpack kernel, serial on one stream (HIP events), max over ranks.  Inside the
    # timed region it is launched async and completed two or three steps later (hidden under the next batches' pass 1); this is
    # what an UN-hidden exchange would add to a step (DESIGN.md section 6 states both predictions).
    exchange_ms = None
    if xchs:
        torch.cuda.synchronize()
        o0 = wl.slots[0]
        x = CodeExchange(o0.codes, o0.grain, K, wl.Bglobal, numel_per_image=H * W * D)    # its own buffers: the slots' results stay for the parity check
        with torch.cuda.stream(streams[0]):
            for _ in range(3):
                x.start(o0.codes, o0.grain, o0.loss)
                x.finish()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nx_it = 50
            e0.record()
            for _ in range(nx_it):
                x.start(o0.codes, o0.grain, o0.loss)
                x.finish()
            e1.record()
        torch.cuda.synchronize()
        exchange_ms = e0.elapsed_time(e1) / nx_it
        if world > 1:
            tx = torch.tensor([exchange_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tx, op=dist.ReduceOp.MAX)
            exchange_ms = float(tx.item())

    parity = None
    if not a.no_parity:
        # every rank checks ITS images against the oracle, after the timed region, on the last step's outputs
        try:
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(max(1, usable_cpus() // world))   # ranks share the host cores
        except Exception:
            pass
        last = wl.slots[(nstep[0] - 1) % S]
        parity = wl.parity(last)
        # every stream slot ran its OWN batch (own seeds): each against the oracle on all of its images
        nbad, nslots = 0, 1
        for o in wl.slots:
            if o is last or nstep[0] <= S:
                continue
            po = wl.parity(o)
            nbad += int(any(v for k_, v in po.items() if k_.endswith("_mismatches")) or po["loss_rel_err"] > 1e-5)
            nslots += 1
        parity["slots_checked"] = nslots                      # stream slots (own inputs each) checked in full on this rank
        parity["slot_mismatches"] = nbad
        if world > 1:
            keys = sorted(k for k, v in parity.items() if isinstance(v, (int, bool)))          # counts: summed over the ranks
            fkeys = sorted(k for k, v in parity.items() if isinstance(v, float) and k != "codes_match_rate_vs_fp64_conv")
            tot = torch.tensor([int(parity[k]) for k in keys], dtype=torch.int64, device=dev)
            dist.all_reduce(tot)
            fl = torch.tensor([parity[k] for k in fkeys], dtype=torch.float64, device=dev)     # errors: the worst rank
            dist.all_reduce(fl, op=dist.ReduceOp.MAX)
            rate = parity.get("codes_match_rate_vs_fp64_conv")
            parity = dict(zip(keys, (int(v) for v in tot.tolist())))
            parity.update(zip(fkeys, (float(v) for v in fl.tolist())))
            if rate is not None:
                rt = torch.tensor([rate], dtype=torch.float64, device=dev)
                dist.all_reduce(rt, op=dist.ReduceOp.MIN)
                parity["codes_match_rate_vs_fp64_conv"] = float(rt.item())
            if xchs:                       # the gathered global tensors agree with the local shard
                xch = xchs[(nstep[0] - 1) % nx]
                g_codes, g_grain, _ = xch.result()
                s0 = sum(b for b in xch.shard_sizes[:rank])
                parity["exchange_ok"] = bool(torch.equal(g_codes[s0:s0 + B], last.codes) and
                                             torch.equal(g_grain[s0:s0 + B], last.grain))

    # the model order beside the headline: the same step behind the models' 1x1 quant_conv (select -> conv -> assign as ONE op,
    # --path model), a short second measurement on the same streams, its own inputs and its own parity check
    model_order = None
    if (world == 1 and a.scaling == "weak" and a.path == "routed" and a.mode == "filter" and not a.no_model_order and not force_xch):
        import copy

        def side_leg(path, note):
            """a short second measurement of another path on the same streams, its own inputs and its own parity check"""
            a2 = copy.copy(a)
            a2.path = path
            wl2 = WeakDual(a2, rank, world, dev)
            wl2.prep_dom = wl.prep_dom
            n2 = [0]

            def step2():
                k = n2[0]
                n2[0] += 1
                with torch.cuda.stream(streams[k % S]):
                    wl2.step(wl2.slots[k % S])
            step2()
            torch.cuda.synchronize()
            for _ in range(50):
                step2()
            torch.cuda.synchronize()
            k2 = min(a.steps, 200)
            dts2 = []
            for _ in range(R):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(k2):
                    step2()
                torch.cuda.synchronize()
                dts2.append(time.perf_counter() - t2)
            dt2 = float(np.median(dts2))
            leg = {"path": path, "steps": k2, "ms_per_step": dt2 / k2 * 1e3,
                   "ms_per_step_min_max": [min(dts2) / k2 * 1e3, max(dts2) / k2 * 1e3],
                   "value": wl2.Bglobal * k2 / dt2, "unit": "images/s"}
            if not a.no_parity:
                p2 = wl2.parity(wl2.slots[(n2[0] - 1) % S])
                bad2 = sum(v for k_, v in p2.items() if k_.endswith("_mismatches"))
                leg["parity_checked"] = bool(bad2 == 0 and p2["loss_rel_err"] <= 1e-5)
                leg["parity"] = {k_: p2[k_] for k_ in ("images_checked", "code_mismatches", "zq_mismatches", "h_max_err_over_bound",
                                                        "codes_match_rate_vs_fp64_conv", "rerun_with_h_buf_mismatches", "loss_rel_err",
                                                        "tokens_resolved_with_a_conv", "token_stream_mismatches") if k_ in p2}
            del wl2
            torch.cuda.empty_cache()
            return leg

        # the model order beside the headline: the same step behind the models' 1x1 quant_conv -- as ONE op with the conv computed in
        # pass 1 (codes, z_q AND loss: what training-time callers of `encode` get), and in the opt-in loss-free form with the conv
        # FOLDED into the codebook (inference / stage-2 tokenisation: no conv is computed for tokens the filter decides)
        model_order = side_leg("model", "bench.py --path model is the full measurement of this path (own roofline, serial step, all slots checked)")
        model_order["fold"] = side_leg("model_fold", "bench.py --path model_fold: the model order for loss-free inference, quant_conv folded "
                                                     "into the codebook (codes + z_q = codebook[code]; same codes as the conv-then-assign order)")
        model_order["tokens_fold"] = side_leg("tokens_fold", "bench.py --path tokens_fold: stage 2's tokenisation of a stage-1 checkpoint (codes only "
                                                             "+ permuter) with the quant_conv folded into the codebook")

    # the same K steps strictly serial (one stream, one slot's buffers): what a caller without stream slots gets
    # (HIP events around blocks of >= 200 steps: with the driver's --steps 20 a wall-clock bracket of 20 steps = 4 ms carried the
    # synchronize / first-launch latency and the clock ramp after the idle gap -- 0.241 against 0.223 ms on the same box)
    serial_ms, serial_blocks = None, None
    n_ser = max(a.steps, 200)
    if S > 1:
        serial_blocks = []
        for _ in range(10):
            wl.step(wl.slots[0])
        for r in range(R):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            for _ in range(5):
                wl.step(wl.slots[0])
            e0.record()
            for _ in range(n_ser):
                wl.step(wl.slots[0])
            e1.record()
            torch.cuda.synchronize()
            serial_blocks.append(e0.elapsed_time(e1) / n_ser)
        serial_ms = float(np.median(serial_blocks))
    # the step's ops (all their kernels) and, below, the dominant kernel alone: HIP events on the launch stream, serial,
    # after the timed region (inside it the ops of consecutive steps overlap across the stream slots)
    nev = 200                                                 # (not --steps: 20 brackets are too few for a 2 % question)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nev)]
    for i in range(-3, nev):
        wl.step(wl.slots[0], ev[i] if i >= 0 else None)
    torch.cuda.synchronize()
    op_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    if serial_ms is None:
        serial_ms = dt / a.steps * 1e3
    # the dominant kernel alone, HIP events on the launch stream, after the timed region
    dom_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nev)]
    scratch = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for i in range(-3, nev):
        # the profiling mode leaves its workspace's queue counters dirty; they are zeroed HERE, in front of the bracket, and the
        # workspace declared clean, so that the events enclose the pass-1 kernel and nothing else (rounds 1-4 had the op's own
        # 5-us counter-zero kernel inside the bracket)
        lw = getattr(wl.prep_dom, "_last_ws", None)
        if lw is not None and hasattr(lw[1], "clean"):
            lw[1].t[:min(lw[1].t.numel(), 4 << 20)].zero_()
            lw[1].clean = True
        wl.dominant(wl.slots[0], dom_ev[i] if i >= 0 else scratch)
    torch.cuda.synchronize()
    dom_ms_bracketed = float(np.mean([s.elapsed_time(e) for s, e in dom_ev]))
    dom_ms = dom_ms_bracketed
    if a.mode == "filter":
        # The same kernel BACK TO BACK: NP launches between ONE pair of events, each through a _CodebookPrep (workspace) of its
        # own that was zeroed and declared clean in front of the bracket -- so the bracket holds NP pass-1 kernels and nothing
        # else, and the event / first-launch latency (4-7 % of a single-launch bracket; rocprofv3's per-kernel duration has none)
        # is amortised.  This is `roofline.kernel_ms`; the single-launch bracket stays beside it as `kernel_ms_bracketed`.
        NP = 8
        keep = wl.prep_dom
        preps = [_CodebookPrep() for _ in range(NP)]

        def zero_all():
            for pp in preps:
                lw = getattr(pp, "_last_ws", None)
                if lw is not None and hasattr(lw[1], "clean"):
                    lw[1].t[:min(lw[1].t.numel(), 4 << 20)].zero_()
                    lw[1].clean = True
        blocks = []
        for r in range(-2, 12):
            zero_all()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for pp in preps:
                wl.prep_dom = pp
                wl.dominant(wl.slots[0], None)
            e1.record()
            torch.cuda.synchronize()
            if r >= 0:
                blocks.append(e0.elapsed_time(e1) / NP)
        wl.prep_dom = keep
        del preps
        dom_ms = float(np.median(blocks))
    N = wl.dominant_tokens()                                  # tokens per launch of the dominant kernel
    per_token = D * 4 + 8 + 4 + (D * 4 if a.path not in ("tokens", "tokens_model", "tokens_fold") else 0)   # z read + int64 code + mask (+ z_q write)
    alg_bytes = N * per_token + K * D * 4                     # codebook once per launch
    alg_flops = 2.0 * K * D * N
    gbs = alg_bytes / (dom_ms * 1e-3) / 1e9
    tfs = alg_flops / (dom_ms * 1e-3) / 1e12
    traffic, tsrc = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(a.mode if a.path in ("select", "model2") else ("routed" if a.path not in ("model", "tokens_fold", "model_fold") else a.path), {}).get("hbm_bytes_per_launch")
            tsrc = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command, collected by tools/pmc_traffic.py; not re-measured in this run)"
        except Exception:
            traffic = None
    # rocprofv3 --kernel-trace --stats duration of the same kernel, if a summary made from THESE sources is committed
    rocprof_ms = None
    for mname in ("bench_kernel_stats.%s.meta.json" % a.path, "bench_kernel_stats.meta.json"):     # per path, then the default path's
        try:
            meta = json.load(open(os.path.join(ROOT, "profiles", mname)))
            if meta.get("source_sha16") == source_sha16() and meta.get("path") == a.path and meta.get("scaling") == a.scaling:
                rocprof_ms = meta.get("dominant_kernel_avg_ms")
                break
        except Exception:
            pass
    if a.mode == "filter":
        roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic}
    else:
        roof = {"bound": "mfma", "achieved": tfs, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tfs / FP32_MFMA_PEAK_TF, "traffic": traffic}
    roof.update({"kernel": wl.dominant_kernel_name(), "kernel_ms": dom_ms, "kernel_ms_bracketed": dom_ms_bracketed,
                 "kernel_ms_rocprof": rocprof_ms,
                 "serial_over_kernel": (serial_ms / dom_ms) if (serial_ms and dom_ms) else None,
                 "algorithmic_bytes": alg_bytes, "algorithmic_flops": alg_flops, "fp16_mfma_frac_of_2500": tfs / 2500.0,
                 "whole_op_ms": op_ms, "target_frac": 0.70, "target_met": bool(gbs / HBM_PEAK_GBS >= 0.70)})
    # (what the fields mean -- the brackets, the byte count, where `traffic` and `kernel_ms_rocprof` come from -- is DESIGN.md
    # section 5; the line itself stays short enough for the driver's record to hold all of it)
    configs = None
    if (rank == 0 and world == 1 and a.scaling == "weak" and a.path == "routed" and a.mode == "filter" and not force_xch
            and a.configs != "off" and not a.no_parity):
        mode_c = a.configs if a.configs != "auto" else ("full" if a.batch is None else "small")
        del wl.slots[1:]
        torch.cuda.empty_cache()
        try:
            configs = configs_block(dev, small=(mode_c == "small"))
        except Exception as ex:                                  # the headline line must survive a failure of the side measurements
            import traceback
            traceback.print_exc()
            configs = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300]), "all_parity_ok": False}
    if rank == 0:
        par = ("one process per GPU, image-parallel x%d; per step ONE packed all-gather of codes / grain / loss pair (%s), "
               "async under the next batches" % (world, backend_name)) if xchs else "single GPU: no exchange runs"
        out = {
            "metric": "images encoded/sec (VQ hot path: gate + routing + VQ assign), 256x256 inputs, K=%d" % K,
            "value": wl.Bglobal * a.steps / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "ms_per_step_min": min(dts) / a.steps * 1e3, "ms_per_step_max": max(dts) / a.steps * 1e3,
            "serial_ms_per_step": serial_ms,
            "serial_ms_per_step_min_max": [min(serial_blocks), max(serial_blocks)] if serial_blocks else None,
            "serial_value": wl.Bglobal / (serial_ms * 1e-3) if world == 1 else None, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.describe(), "global_batch": wl.Bglobal, "assign_mode": a.mode, "path": a.path,
                       "source_sha16": source_sha16(), "spinup_steps": a.spinup, "streams": S, "repeats": R,
                       "setup_seconds_slowest_rank": setup_s, "host_issue_ms_per_step": t_issue / a.steps * 1e3,
                       "parallelism": par, "backend": backend_name, "world": world, "distinct_devices": distinct_devices,
                       "ranks": [{k: d[k] for k in ("rank", "local_rank", "device_index", "pci_bus_id", "device_uuid", "device_name")}
                                 for d in idents]},
            "roofline": roof,
        }
        if exchange_ms is not None:
            out["exchange_ms_per_step"] = exchange_ms          # pack -> all-gather -> unpack, un-hidden (serial, max over ranks)
        if parity is not None:
            bad = sum(v for k, v in parity.items() if k.endswith("_mismatches"))
            out["parity_checked"] = bool(bad == 0 and parity["loss_rel_err"] <= 1e-5 and parity.get("exchange_ok", True))
            out["code_mismatches"] = parity["code_mismatches"]
            out["parity"] = parity
            if world == 1 and getattr(wl, "oracle_seconds", None):
                out["parity"]["oracle_full_batch_images_per_s"] = wl.B / wl.oracle_seconds   # cold, fresh output arrays
        if model_order is not None:
            out["model_order"] = model_order
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl.E_np, a.cpu_seconds)
        if configs is not None:
            out["configs"] = configs                           # last: a truncated record keeps the END of the line
        def sig7(v):                                           # side blocks at 7 significant digits: the line stays inside the
            if isinstance(v, float):                           # driver's record (value / ms_per_step keep every digit)
                return float("%.7g" % v)
            if isinstance(v, dict):
                return {k: sig7(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [sig7(x) for x in v]
            return v
        for k in list(out):
            if k not in ("value", "ms_per_step"):
                out[k] = sig7(out[k])
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out, separators=(",", ":")), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
    if world > 1 or force_xch:
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))          # nothing above has touched a GPU
    run_rank(a)


if __name__ == "__main__":
    main()
