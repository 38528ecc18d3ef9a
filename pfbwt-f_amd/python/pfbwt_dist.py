"""Sharded parse across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI; "gloo" in the CPU tests).  Every rank parses a run of whole sequences on its own GPU, the per-rank phrase
dictionaries and phrase-id sequences travel in ONE all-gather (padded to the largest rank: all_gather needs equal sizes, this is
the all-gather-v of the design); every rank then merges them, sorts the merged dictionary and the parse, and emits
its own slice of the BWT/SA rows (see sharded_build).
torch is plumbing here: device buffers + the collective."""
import torch
import torch.distributed as dist


def _align(x, a=256):
    return (x + a - 1) // a * a


def pack_local_shard(ctx, device):
    """Copy the dictionary, the word starts and the phrase ids of the finished local parse into one contiguous uint8 tensor
    (phrase ends and last bytes do not travel: every phrase is a dictionary word, pfp_merge_shards derives them -- S-32G at 8 ranks:
    ~290 MB per rank instead of 657 MB)."""
    v = ctx.shard_view()
    sizes = v.nbytes(compact=True)
    offs, tot = [], 0
    for b in sizes:
        offs.append(tot); tot = _align(tot + b)
    buf = torch.empty(max(tot, 256), dtype=torch.uint8, device=device)
    for off, b, ptr in zip(offs, sizes, (v.d_dict, v.d_ws, v.d_pid)):
        ctx.device_copy(buf.data_ptr() + off, ptr, b)
    meta = torch.tensor([v.n, v.m, v.dwords, v.dsize, tot, v.left_context], dtype=torch.int64, device=device)
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
    if dist.get_backend(group) == "nccl":
        # one receive buffer, no staging copies inside the collective (the payloads are GBs on a 1000-haplotype collection)
        flat = torch.empty(world * maxb, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(flat, buf[:maxb], group=group)
        recv = [flat[r * maxb:(r + 1) * maxb] for r in range(world)]
    else:
        recv = [torch.empty(maxb, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(recv, buf[:maxb].contiguous(), group=group)
    if torch.device(device).type == "cuda":
        # the collective runs on torch's / RCCL's stream, the engine reads the receive buffers on its own HIP stream:
        # the host waits for the collective before the pointers are handed over
        torch.cuda.synchronize(device)
    views = []
    for r in range(world):
        n, m, dw, ds, _, lc = (int(x) for x in metas[r])
        v = pfbwt_hip.ShardView(); v.n, v.m, v.dwords, v.dsize, v.left_context = n, m, dw, ds, lc
        off, ptrs = 0, []
        for b in v.nbytes(compact=True):
            ptrs.append(recv[r].data_ptr() + off); off = _align(off + b)
        v.d_dict, v.d_ws, v.d_pid = ptrs
        v.d_ye = v.d_last = None
        views.append(v)
    return views, recv


def allgather_docs(local_docs, n_local, group=None):
    """--print-docs through the sharded build (pfparser.hpp:321-325): local_docs = [(name, start inside this rank's own text)],
    n_local = bytes of text this rank fed (records + their w 'A's, WITHOUT the left context).  Returns the (name, start) list
    of the whole collection, the same on every rank -- what `.docs` holds after a single parse."""
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, (int(n_local), [(str(nm), int(st)) for nm, st in local_docs]), group=group)
    out, off = [], 0
    for n_r, docs in parts:
        out += [(nm, st + off) for nm, st in docs]; off += n_r
    return out


def sharded_build(ctx, feed_local, w, device, sa=True, rssa=False, group=None):
    """SPMD build over the ranks of `group`.
    1. every rank parses its own run of whole sequences (rank r > 0 with the w 'A's of the previous shard as context);
    2. ONE all-gather moves every rank's dictionary + parse to every rank;
    3. every rank merges them (pfp_merge_shards) and suffix-sorts the merged dictionary and the parse -- redundant but
       identical work (≈ constant for a haplotype panel), which saves broadcasting ≈ 15 B per dictionary byte;
    4. every rank emits its own slice of the output rows (pfp_bwt_build_slice): .bwt / .sa stay distributed in HBM.
    feed_local(ctx) feeds this rank's sequences.  Returns (parse sizes, bwt sizes of the slice, first row, rows)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    err = None
    try:
        if rank > 0:
            ctx.feed_left_context(w)
        feed_local(ctx)
        ctx.finalize(shard=True)        # no dictionary sort / ranks on a shard: the merge makes them for the union (a shard whose
                                        # parse is ONE phrase -- a short record without a trigger window -- is folded into its seams)
    except Exception as e:      # an invalid character, a shard that is too small, ...: every rank must learn of it BEFORE the
        err = e                  # collective, or the healthy ranks would wait in the all-gather for ever
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if err is not None:
        raise err
    if int(ok.item()) == 0:
        raise RuntimeError("sharded_build: another rank failed to parse its shard")
    views, keep = allgather_shards(ctx, device, group)
    sz = ctx.merge_shards(views)     # the local parse is consumed through its copy in `keep`
    del keep, views
    ctx.parse_bwt()
    b, begin, rows = ctx.bwt_build_slice(rank, world, sa=sa, rssa=rssa)
    return sz, b, begin, rows
