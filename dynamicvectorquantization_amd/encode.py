"""Encode glue of the hot path and its image-parallel (one process per GPU) form.

Mirrors the part of the reference's `encode` methods that follows the CNN trunk:
  DualGrainVQModel.encode    models/stage1_dynamic/dqvae_dual_feat.py:59-68, dqvae_dual_entropy.py:124-134
  TripleGrainVQModel.encode  models/stage1_dynamic/dqvae_triple_feat.py:68-77
  VQModel.encode             models/stage1/vqgan.py:68-72
i.e. router -> routing tail (select + codebook_mask) -> quant_conv -> quantizer, returning
(quant, emb_loss, info, grain_indices, gate) with the reference's shapes and dtypes.  The trunk
(dense convs / attention) is not part of this package; feed its branch outputs.

Sharding: images are independent, so rank r of G encodes the contiguous
slice [r*B/G, (r+1)*B/G) with the codebook replicated; the only exchange is ONE all-gather per
batch of a packed byte buffer (codes as int16/int32, grain indices as int8, the (loss-sum,
element-count) pair as float64) -- RCCL over xGMI when the process group is "nccl" -- which can run
asynchronously under the next batch's kernels.
"""
import torch
import torch.distributed as dist

from .router import DualGrainFixedEntropyRouter, route_select_dual, route_select_dual_entropy, route_select_triple


def encode_dual(router, quantize, h_fine, h_coarse, entropy=None, quant_conv=None, temp=0.0):
    """-> (quant, emb_loss, info, grain_indices, gate) as DualGrainVQModel.encode."""
    if isinstance(router, DualGrainFixedEntropyRouter) and entropy is not None and entropy.is_cuda:
        sel = route_select_dual_entropy(entropy, router.fine_grain_threshold, h_coarse, h_fine)   # gate fused in
    else:
        gate = router(h_fine=h_fine, h_coarse=h_coarse, entropy=entropy)
        sel = route_select_dual(gate, h_coarse, h_fine)
    h = sel["h_dual"]
    if quant_conv is not None:
        h = quant_conv(h)
    quant, emb_loss, info = quantize(x=h, temp=temp, codebook_mask=sel["codebook_mask"])
    return quant, emb_loss, info, sel["indices"], sel["gate"]


def encode_triple(router, quantize, h_fine, h_median, h_coarse, quant_conv=None, temp=0.0):
    """-> (quant, emb_loss, info, grain_indices, gate) as TripleGrainVQModel.encode."""
    gate = router(h_fine=h_fine, h_median=h_median, h_coarse=h_coarse, entropy=None)
    sel = route_select_triple(gate, h_coarse, h_median, h_fine)
    h = sel["h_triple"]
    if quant_conv is not None:
        h = quant_conv(h)
    quant, emb_loss, info = quantize(x=h, temp=temp, codebook_mask=sel["codebook_mask"])
    return quant, emb_loss, info, sel["indices"], sel["gate"]


def encode_fixed(quantize, h, quant_conv=None):
    """-> (quant, emb_loss, info) as VQModel.encode (fixed granularity)."""
    if quant_conv is not None:
        h = quant_conv(h)
    return quantize(h)


def shard_slice(global_batch, rank, world_size):
    """contiguous image slice of `rank`; the first (global_batch % world_size) ranks get one extra"""
    base, extra = divmod(global_batch, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _wire_dtype(num_codes):
    return torch.int16 if num_codes <= 32768 else torch.int32


class _PendingGather:
    """Handle of an exchange in flight (all_gather_codes(..., async_op=True)): `.wait()` makes the
    current stream wait for the collective and returns (codes, grain_indices, mean)."""

    def __init__(self, work, unpack):
        self._work, self._unpack, self._result = work, unpack, None

    def wait(self):
        if self._result is None:
            if self._work is not None:
                self._work.wait()
            self._result = self._unpack()
        return self._result


def all_gather_codes(codes, grain_indices, loss_sum, numel, num_codes, global_batch, group=None,
                     async_op=False):
    """Exchange step of the image-parallel encode: ONE all-gather per batch.

    codes [b_local, H, W] int64, grain_indices [b_local, h, w] int64 or None, loss_sum 0-dim tensor
    (local sum of the loss numerator), numel = local element count.  Every rank packs
    [codes as int16/int32 | grain indices as int8 | (loss_sum, numel) as 2 x float64] into one byte
    buffer (neither RCCL nor gloo has an int16 type, so everything travels as uint8), the buffers
    are all-gathered, and each rank adds the per-rank loss pairs in rank order (same bits on every
    rank).  Returns (codes [B, H, W] int64, grain_indices [B, h, w] int64 or None, global mean) on
    every rank; with async_op=True a handle whose .wait() returns that tuple, so the exchange of
    batch i overlaps the kernels of batch i+1.  Ragged shards (B % world != 0) are padded to the
    largest shard on the wire."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_slice(global_batch, r, world) for r in range(world)]
    assert sizes[rank][1] - sizes[rank][0] == codes.shape[0]
    bmax = max(e - s for s, e in sizes)
    wd = _wire_dtype(num_codes)
    dev = codes.device
    esize = torch.empty((), dtype=wd).element_size()
    cshape, gshape = tuple(codes.shape[1:]), (tuple(grain_indices.shape[1:]) if grain_indices is not None else None)
    n_c = bmax * int(torch.Size(cshape).numel()) * esize
    n_g = bmax * int(torch.Size(gshape).numel()) if gshape is not None else 0
    n_c_pad = (n_c + 7) // 8 * 8
    n_g_pad = (n_g + 7) // 8 * 8
    nbytes = n_c_pad + n_g_pad + 16
    b_local = codes.shape[0]

    local = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    local[:b_local * (n_c // bmax)] = codes.to(wd).contiguous().view(torch.uint8).reshape(-1)
    if grain_indices is not None:
        local[n_c_pad:n_c_pad + b_local * (n_g // bmax)] = grain_indices.to(torch.int8).contiguous().view(torch.uint8).reshape(-1)
    pair = torch.stack([loss_sum.detach().to(torch.float64).reshape(()),
                        torch.full((), float(numel), dtype=torch.float64, device=dev)])
    local[n_c_pad + n_g_pad:] = pair.view(torch.uint8)
    out = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)

    def unpack():
        buf = out.view(world, nbytes)
        c_parts, g_parts = [], []
        for r, (s, e) in enumerate(sizes):
            c = buf[r, :n_c].view(wd).reshape((bmax,) + cshape)[: e - s]
            c_parts.append(c)
            if gshape is not None:
                g_parts.append(buf[r, n_c_pad:n_c_pad + n_g].view(torch.int8).reshape((bmax,) + gshape)[: e - s])
        pairs = buf[:, n_c_pad + n_g_pad:].contiguous().view(torch.float64).reshape(world, 2)
        tot = pairs.sum(0)                                   # rank order: identical on every rank
        g_codes = torch.cat(c_parts, 0).to(torch.int64)
        g_grain = torch.cat(g_parts, 0).to(torch.int64) if gshape is not None else None
        return g_codes, g_grain, (tot[0] / tot[1]).to(torch.float32)

    if async_op:
        return _PendingGather(work, unpack)
    return unpack()
