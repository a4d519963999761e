"""CPU, world_size 2, gloo: the exchange step of the image-parallel encode (all-gather of codes and
grain indices, all-reduce of the loss numerator) reproduces the single-process result.  The
per-rank encode itself is played by the oracle here (the HIP path needs a GPU; it is covered by
-m gpu tests); what is under test is sharding + collectives + ragged batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dynamicvectorquantization_amd import synth
        from dynamicvectorquantization_amd.encode import all_gather_codes, shard_slice
        from oracle import oracle
        K, D = 64, 64
        E = synth.codebook_trained(K, D, seed=9)
        s, e = shard_slice(B, rank, world)
        z = synth.z_tokens(E, e - s, 8, 8, 444, image_offset=s)          # this rank's images only
        gate = synth.grain_gate_dual(445, e - s, 4, 4, image_offset=s)
        o = oracle.vq_assign_nchw(z, E, None)
        codes = torch.from_numpy(o["codes"].reshape(e - s, 8, 8))
        grain = torch.from_numpy(gate.argmax(-1))
        g_codes, g_grain, mean = all_gather_codes(codes, grain, torch.tensor(o["sqerr"]), o["numel"], K, B)
        # two exchanges in flight (the pipelined form bench.py uses), waited in order
        h1 = all_gather_codes(codes, grain, torch.tensor(o["sqerr"]), o["numel"], K, B, async_op=True)
        h2 = all_gather_codes(codes + 1, None, torch.tensor(2.0 * o["sqerr"]), o["numel"], K, B, async_op=True)
        a_codes, a_grain, a_mean = h1.wait()
        b_codes, b_grain, b_mean = h2.wait()
        assert torch.equal(a_codes, g_codes) and torch.equal(a_grain, g_grain) and float(a_mean) == float(mean)
        assert torch.equal(b_codes, g_codes + 1) and b_grain is None and abs(float(b_mean) - 2 * float(mean)) < 1e-6 * float(mean)
        # the preallocated-exchange object bench.py uses (CPU tensors: same wire format through torch ops)
        from dynamicvectorquantization_amd.encode import CodeExchange
        xch = CodeExchange(codes, grain, K, B, numel_per_image=8 * 8 * D)
        assert xch.shard_sizes == [shard_slice(B, r, world)[1] - shard_slice(B, r, world)[0] for r in range(world)]
        xch.start(codes, grain, torch.tensor([o["sqerr"] / o["numel"]], dtype=torch.float32))
        c_codes, c_grain, c_mean = xch.finish()
        assert torch.equal(c_codes, g_codes) and torch.equal(c_grain, g_grain)
        assert abs(float(c_mean) - float(mean)) <= 1e-6 * abs(float(mean))
        q.put((rank, g_codes.numpy(), g_grain.numpy(), float(mean)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_all_gather_codes_world2(B):
    from dynamicvectorquantization_amd import synth
    from oracle import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    K, D = 64, 64
    E = synth.codebook_trained(K, D, seed=9)
    z = synth.z_tokens(E, B, 8, 8, 444)
    gate = synth.grain_gate_dual(445, B, 4, 4)
    o = oracle.vq_assign_nchw(z, E, None)
    for rank, codes, grain, mean in res:
        assert codes.dtype == np.int64 and codes.shape == (B, 8, 8)
        assert np.array_equal(codes, o["codes"].reshape(B, 8, 8))
        assert np.array_equal(grain, gate.argmax(-1))
        assert abs(mean - o["sqerr"] / o["numel"]) <= 1e-6 * abs(mean)


def _ema_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dynamicvectorquantization_amd.quantize import VQEmbedding
        torch.manual_seed(5)                                   # same initial codebook on both ranks
        emb = VQEmbedding(16, 8, decay=0.0, restart_unused_codes=True)   # decay 0: EMA = this step's statistics
        g = torch.Generator().manual_seed(100 + rank)          # different tokens / assignments per rank
        vec = torch.randn(40, 8, generator=g)
        idx = torch.randint(0, 12, (40,), generator=g)         # codes 12..15 stay unused -> restarted
        emb._update_buffers(vec, idx)
        emb._update_embedding()
        q.put((rank, vec.numpy(), idx.numpy(), emb.cluster_size_ema.numpy().copy(),
               emb.embed_ema.numpy().copy(), emb.weight.detach().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_ema_update_world2_single_collective():
    """training-mode EMA statistics under data parallelism (reference quantize2_mask.py:86-100): the
    fused all_reduce over [K*D + K] leaves every rank with the statistics of the union of the shards,
    and the dead-code restart vectors are rank 0's on every rank"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ema_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, v0, i0, cs0, ee0, w0), (_, v1, i1, cs1, ee1, w1) = res
    np.testing.assert_array_equal(cs0, cs1)
    np.testing.assert_array_equal(ee0, ee1)
    np.testing.assert_array_equal(w0[:-1], w1[:-1])
    idx = np.concatenate([i0, i1]); vec = np.concatenate([v0, v1])
    counts = np.bincount(idx, minlength=16).astype(np.float64)
    used = counts > 0
    sums = np.zeros((16, 8)); np.add.at(sums, idx, vec.astype(np.float64))
    np.testing.assert_array_equal(cs0[used], counts[used])                 # all-reduced counts of BOTH shards
    np.testing.assert_allclose(ee0[used], sums[used], rtol=1e-5, atol=1e-6)
    assert used.sum() == 12 and np.all(cs0[~used] == 1.0)                   # dead codes restarted ...
    rows0 = {tuple(np.round(r, 5)) for r in v0}
    assert all(tuple(np.round(r, 5)) in rows0 for r in ee0[~used])          # ... with rank 0's vectors everywhere


def test_bench_launcher_fails_fast_when_a_rank_dies_in_setup():
    """a rank that dies in set-up must not leave the others (and the launcher) in the rendezvous until torch's
    timeout.  Rank 1 raises before it joins (DVQ_BENCH_TEST_FAIL_RANK), rank 0 and 2 sit in the gloo rendezvous (which needs no
    GPU): the launcher sees the non-zero exit, ends the other ranks it started and returns non-zero within seconds.  Also the
    wall-time bound: nobody fails, nobody can finish (3 ranks announced, rank 2 never joins -- it is told to sleep)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DVQ_BENCH_BACKEND="gloo", DVQ_BENCH_TEST_FAIL_RANK="1")
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    dt = time.time() - t0
    assert p.returncode != 0
    assert dt < 30.0, "launcher took %.1f s to notice a dead rank" % dt
    assert "rank 1 exited with code" in p.stderr, p.stderr[-2000:]
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # the wall-time bound
    env = dict(os.environ, DVQ_BENCH_BACKEND="gloo", DVQ_BENCH_TEST_SLEEP_RANK="2", DVQ_BENCH_LAUNCH_TIMEOUT="4")
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    dt = time.time() - t0
    assert p.returncode != 0 and dt < 30.0, (p.returncode, dt)
    assert "wall-time bound" in p.stderr, p.stderr[-2000:]


@pytest.mark.gpu
def test_bench_launcher_spawns_ranks_and_exchanges_on_device():
    """`python bench.py --gpus 2` is a launcher -- it starts 2 rank processes itself (here both
    on the one visible GPU, gloo rendezvous), the ranks run the routed step + pack kernel + all-gather + unpack
    kernel, rank 0's JSON line reports n_gpus == 2 with the parity check of every rank's images"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DVQ_BENCH_BACKEND="gloo")
    env.pop("RANK", None)
    for extra in (["--batch", "8"], ["--scaling", "strong", "--batch", "12"], ["--batch", "8", "--path", "model"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                            "--spinup", "1", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True,
                           timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == ("strong" if "strong" in extra else "weak")
        assert d["parity_checked"] is True and d["code_mismatches"] == 0 and d["parity"]["exchange_ok"] is True
        assert d["parity"]["images_checked"] == (12 if "strong" in extra else 16)


@pytest.mark.gpu
def test_bench_launcher_eight_ranks_on_one_device():
    """the world size the launcher exists for.  `bench.py --gpus 8` on the ONE visible GPU (gloo rendezvous):
    8-way rendezvous, BASELINE configs[3] strong scaling (global B = 1024 -> 128-image shards, fused feature-router gate + triple
    routing + assign) and weak scaling at 16 images per rank; every rank checks its own images against the oracle, the gathered
    global tensors against its shard (exchange_ok), and rank 0 reports the slowest rank's set-up time, so that the real 8-GPU run
    is known to fit the driver's timeout on 16 host cores.  NOT a scaling measurement: eight ranks share one device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DVQ_BENCH_BACKEND="gloo")
    env.pop("RANK", None)
    for extra, nimg in ((["--scaling", "strong", "--batch", "1024"], 1024), (["--batch", "16"], 128)):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                            "--spinup", "1", "--repeats", "2", "--no-cpu-baseline"] + extra, env=env, capture_output=True,
                           text=True, timeout=1500)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["n_gpus"] == 8 and d["config"]["global_batch"] == nimg
        assert d["parity_checked"] is True and d["code_mismatches"] == 0 and d["parity"]["exchange_ok"] is True
        assert d["parity"]["images_checked"] == nimg
        # the pre-flight record: backend, world, every rank's device identity, the un-hidden exchange time
        cfg = d["config"]
        assert cfg["backend"] == "gloo" and cfg["world"] == 8 and len(cfg["ranks"]) == 8
        assert [r["rank"] for r in cfg["ranks"]] == list(range(8)) and all(r["device_name"] and r["pci_bus_id"] for r in cfg["ranks"])
        assert cfg["distinct_devices"] == 1          # eight ranks folded onto the one visible GPU: allowed under gloo ONLY
        assert d["exchange_ms_per_step"] > 0
        print("8 ranks on one device, %s: slowest rank's set-up %.1f s, %.3f ms per step (not a scaling figure)"
              % (" ".join(extra), d["config"]["setup_seconds_slowest_rank"], d["ms_per_step"]))
        assert d["config"]["setup_seconds_slowest_rank"] < 600


@pytest.mark.gpu
def test_code_exchange_rccl_world1(dev):
    """the REAL nccl (= RCCL) backend through CodeExchange at world size 1 (what a 1-GPU box can run): pack kernel ->
    ncclAllGather -> unpack kernel on the device, two exchanges in flight, against the local tensors.  In a child
    process: the process group must be created before anything else touches the GPU state it binds."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from dynamicvectorquantization_amd.encode import CodeExchange
B, K = 8, 1024
g = torch.Generator().manual_seed(3)
codes = torch.randint(0, K, (B, 32, 32), generator=g).to(dev)
grain = torch.randint(0, 2, (B, 16, 16), generator=g).to(dev)
loss = torch.tensor([0.125, 0.15625], device=dev)
xs = [CodeExchange(codes, grain, K, B, numel_per_image=32 * 32 * 256) for _ in range(2)]
assert xs[0].on_gpu and xs[0].world == 1
xs[0].start(codes, grain, loss)
xs[1].start(codes + 1 - 2 * (codes == K - 1).long(), grain, None)
c0, g0, m0 = xs[0].finish()
c1, g1, m1 = xs[1].finish()
torch.cuda.synchronize()
assert torch.equal(c0, codes) and torch.equal(g0, grain) and abs(float(m0) - 0.125) < 1e-7
assert torch.equal(c1, codes + 1 - 2 * (codes == K - 1).long()) and torch.equal(g1, grain)
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
