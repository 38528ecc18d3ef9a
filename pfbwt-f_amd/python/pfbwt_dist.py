"""Sharded parse across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI; "gloo" in the CPU tests).  Every rank parses a run of whole sequences on its own GPU, the per-rank phrase
dictionaries and parses travel in ONE all-gather (padded to the largest rank: all_gather needs equal sizes, this is
the all-gather-v of the design), and rank 0 merges them (pfp_merge_shards) and runs the single-GPU stages.
torch is plumbing here: device buffers + the collective."""
import torch
import torch.distributed as dist


def _align(x, a=256):
    return (x + a - 1) // a * a


def pack_local_shard(ctx, device):
    """Copy the five device arrays of the finished local parse into one contiguous uint8 tensor."""
    v = ctx.shard_view()
    sizes = v.nbytes()
    offs, tot = [], 0
    for b in sizes:
        offs.append(tot); tot = _align(tot + b)
    buf = torch.empty(max(tot, 256), dtype=torch.uint8, device=device)
    for off, b, ptr in zip(offs, sizes, (v.d_dict, v.d_ws, v.d_pid, v.d_ye, v.d_last)):
        ctx.device_copy(buf.data_ptr() + off, ptr, b)
    meta = torch.tensor([v.n, v.m, v.dwords, v.dsize, tot], dtype=torch.int64, device=device)
    return buf, meta


def allgather_shards(ctx, device, group=None):
    """Returns (views, keepalive): ShardView list for all ranks with device pointers into the receive buffers."""
    import pfbwt_hip
    world = dist.get_world_size(group)
    buf, meta = pack_local_shard(ctx, device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    maxb = int(max(int(m[4]) for m in metas))
    if buf.numel() < maxb:
        pad = torch.empty(maxb, dtype=torch.uint8, device=device); pad[:buf.numel()] = buf; buf = pad
    recv = [torch.empty(maxb, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(recv, buf[:maxb].contiguous(), group=group)
    views = []
    for r in range(world):
        n, m, dw, ds, _ = (int(x) for x in metas[r])
        v = pfbwt_hip.ShardView(); v.n, v.m, v.dwords, v.dsize = n, m, dw, ds
        off, ptrs = 0, []
        for b in v.nbytes():
            ptrs.append(recv[r].data_ptr() + off); off = _align(off + b)
        v.d_dict, v.d_ws, v.d_pid, v.d_ye, v.d_last = ptrs
        views.append(v)
    return views, recv


def sharded_build(ctx, feed_local, w, device, sa=True, rssa=False, group=None):
    """feed_local(ctx) feeds this rank's sequences.  Rank 0 returns (parse sizes, bwt sizes); other ranks None."""
    rank = dist.get_rank(group)
    if rank > 0:
        ctx.feed_left_context(w)
    feed_local(ctx)
    ctx.finalize()
    views, keep = allgather_shards(ctx, device, group)
    if rank != 0:
        return None
    sz = ctx.merge_shards(views)     # the local parse of rank 0 is consumed through its view in `keep`
    ctx.parse_bwt()
    b = ctx.bwt_build(sa=sa, rssa=rssa)
    del keep
    return sz, b
