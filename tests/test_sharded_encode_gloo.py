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
