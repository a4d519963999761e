"""Encode glue of the hot path and its image-parallel (one process per GPU) form.

Mirrors the part of the reference's `encode` methods that follows the CNN trunk:
  DualGrainVQModel.encode    models/stage1_dynamic/dqvae_dual_feat.py:59-68, dqvae_dual_entropy.py:124-134
  TripleGrainVQModel.encode  models/stage1_dynamic/dqvae_triple_feat.py:68-77
  VQModel.encode             models/stage1/vqgan.py:68-72
i.e. router -> routing tail (select + codebook_mask) -> quant_conv -> quantizer, returning
(quant, emb_loss, info, grain_indices, gate) with the reference's shapes and dtypes.  The trunk
(dense convs / attention) is not part of this package; feed its branch outputs.

Sharding: images are independent, so rank r of G encodes the contiguous
slice [r*B/G, (r+1)*B/G) with the codebook replicated; the only exchange is an all-gather of the
emitted integers (codes as int16/int32 on the wire, grain indices as int8) and an all-reduce of the
(loss-sum, element-count) pair -- RCCL over xGMI when the process group is "nccl".
"""
import torch
import torch.distributed as dist

from .router import route_select_dual, route_select_triple


def encode_dual(router, quantize, h_fine, h_coarse, entropy=None, quant_conv=None, temp=0.0):
    """-> (quant, emb_loss, info, grain_indices, gate) as DualGrainVQModel.encode."""
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


def all_gather_codes(codes, grain_indices, loss_sum, numel, num_codes, global_batch, group=None):
    """Exchange step of the image-parallel encode.

    codes [b_local, H, W] int64, grain_indices [b_local, h, w] int64 or None, loss_sum 0-dim tensor
    (local sum of the loss numerator), numel = local element count.  Returns
    (codes [B, H, W] int64, grain_indices [B, h, w] int64 or None, global mean) on every rank.
    Ragged shards (B % world != 0) are padded to the largest shard on the wire."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_slice(global_batch, r, world) for r in range(world)]
    bmax = max(e - s for s, e in sizes)
    wd = _wire_dtype(num_codes)

    def gather(t, dtype):
        # the narrow integers travel as raw bytes (uint8 view): neither RCCL nor gloo has an int16 type
        local = t.to(dtype)
        if local.shape[0] < bmax:
            pad = torch.zeros((bmax - local.shape[0],) + tuple(local.shape[1:]), dtype=dtype, device=t.device)
            local = torch.cat([local, pad], 0)
        local = local.contiguous()
        out = torch.empty((world * bmax,) + tuple(local.shape[1:]), dtype=dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(torch.uint8).reshape(-1), local.view(torch.uint8).reshape(-1),
                                    group=group)
        parts = [out[r * bmax: r * bmax + (e - s)] for r, (s, e) in enumerate(sizes)]
        return torch.cat(parts, 0).to(torch.int64)

    g_codes = gather(codes, wd)
    g_grain = gather(grain_indices, torch.int8) if grain_indices is not None else None
    acc = torch.stack([loss_sum.detach().to(torch.float64).reshape(()),
                       torch.tensor(float(numel), dtype=torch.float64, device=loss_sum.device)])
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    assert sizes[rank][1] - sizes[rank][0] == codes.shape[0]
    return g_codes, g_grain, (acc[0] / acc[1]).to(torch.float32)
