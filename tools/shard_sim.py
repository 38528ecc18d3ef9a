#!/usr/bin/env python3
"""Development aid: the N-rank sharded build (pfbwt_dist.sharded_build) replayed on ONE GPU, rank by rank, to time the
stages a rank runs and to check the sliced -r output against the single-context build.

  rank r:  parse haplotypes [r*H/N, (r+1)*H/N)            -> t_parse[r]
           (all-gather: replaced by device copies of every rank's packed shard)
           merge_shards + parse_bwt + bwt_build_slice(r)   -> t_merge, t_pbwt, t_slice[r]

--check builds the same collection in one context and compares: BWT slices, ssa / esa concatenated over the ranks."""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import torch
import pfbwt_hip, pfbwt_dist

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32_000_000); ap.add_argument("--H", type=int, default=1000); ap.add_argument("--seed", type=int, default=1000)
ap.add_argument("--ranks", type=int, default=8); ap.add_argument("--slices", type=str, default="0", help="comma list of ranks whose slice is emitted, or 'all'")
ap.add_argument("--check", action="store_true"); ap.add_argument("--sa", action="store_true", help="-s (full SA per slice) instead of -r")
a = ap.parse_args()
assert a.H % a.ranks == 0
lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
w, p = 10, 100
dev = torch.device("cuda", 0)
Hl = a.H // a.ranks
from concurrent.futures import ThreadPoolExecutor
big = np.empty((a.H, a.L), np.uint8)
with ThreadPoolExecutor(max_workers=16) as ex:
    list(ex.map(lambda h: lib.pfp_synth_haplotype(a.seed, a.L, h, 0, 0, 0, 0, big[h].ctypes.data_as(C.c_void_p)), range(a.H)))
print("synth done", flush=True)


def sync():
    torch.cuda.synchronize()


# --- every rank's local parse, packed like the all-gather payload
packed, metas, t_parse = [], [], []
for r in range(a.ranks):
    ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=True, sai=True)
    for rep in range(2 if r == 0 else 1):       # rank 0 twice: the first run pays the one-time allocations
        t0 = time.time()
        if r > 0:
            ctx.feed_left_context(w)
        for h in range(r * Hl, (r + 1) * Hl):
            ctx.feed(big[h], True)
            if os.environ.get("SHARD_SIM_TRACE"): sync(); print("  rank %d rep %d fed %d" % (r, rep, h), flush=True)
        t1 = time.time(); sz = ctx.finalize(shard=True); sync(); t2 = time.time()
        if os.environ.get("SHARD_SIM_TRACE"): print("  rank %d rep %d finalized" % (r, rep), flush=True)
    buf, meta = pfbwt_dist.pack_local_shard(ctx, dev); sync(); t3 = time.time()
    packed.append(buf); metas.append(meta); t_parse.append(t2 - t1)
    print("rank %d: local n=%d m=%d dwords=%d dsize=%d | feed(H2D) %.2fs parse %.3fs pack %.3fs payload %.1f MB"
          % (r, sz.n, sz.m, sz.dwords, sz.dsize, t1 - t0, t2 - t1, t3 - t2, buf.numel() / 1e6), flush=True)
    ctx.close()
views = []
for r in range(a.ranks):
    n, m, dw, ds, _, lc = (int(x) for x in metas[r])
    v = pfbwt_hip.ShardView(); v.n, v.m, v.dwords, v.dsize, v.left_context = n, m, dw, ds, lc
    off, ptrs = 0, []
    for b in v.nbytes(compact=True):      # what pack_local_shard sends: dictionary, word starts, phrase ids (the merge derives ends / last bytes)
        ptrs.append(packed[r].data_ptr() + off); off = pfbwt_dist._align(off + b)
    v.d_dict, v.d_ws, v.d_pid = ptrs
    v.d_ye = v.d_last = None
    views.append(v)

# --- what every rank does after the all-gather
want = list(range(a.ranks)) if a.slices == "all" else [int(x) for x in a.slices.split(",")]
ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=True, sai=True)
parts = {}
for it, r in enumerate(want):
    if it == len(want) - 1:
        ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.time(); sz = ctx.merge_shards(views); sync(); t1 = time.time()
    if it == len(want) - 1:
        rowsp = sorted(ctx.profile(), key=lambda x: -x["ms"]); tot = sum(x["ms"] for x in rowsp)
        print("  merge_shards kernels (%.1f ms):" % tot, ", ".join("%s %.1f" % (x["kernel"], x["ms"]) for x in rowsp[:10]), flush=True)
        ctx.profile_reset()
    ctx.parse_bwt(); sync(); t2 = time.time()
    if it == len(want) - 1:
        rowsp = sorted(ctx.profile(), key=lambda x: -x["ms"]); tot = sum(x["ms"] for x in rowsp)
        print("  parse_bwt kernels (%.1f ms):" % tot, ", ".join("%s %.1f" % (x["kernel"], x["ms"]) for x in rowsp[:10]), flush=True)
        ctx.profile_reset()
    b, beg, rows = ctx.bwt_build_slice(r, a.ranks, sa=a.sa, rssa=not a.sa); sync(); t3 = time.time()
    print("slice %d/%d: merged n=%d m=%d dwords=%d dsize=%d | merge %.3fs pbwt %.3fs slice-bwt %.3fs | rows [%d,+%d) runs %d  -> rank time %.3fs + parse %.3fs = %.3fs (%.2f Gbases/s at N=%d, all-gather excluded)"
          % (r, a.ranks, sz.n, sz.m, sz.dwords, sz.dsize, t1 - t0, t2 - t1, t3 - t2, beg, rows, b.r, t3 - t0, max(t_parse), t3 - t0 + max(t_parse),
             sz.n / (t3 - t0 + max(t_parse)) / 1e9, a.ranks), flush=True)
    if a.check:
        parts[r] = ctx.bwt_get()
rowsp = sorted(ctx.profile(), key=lambda x: -x["ms"]); tot = sum(x["ms"] for x in rowsp)
for x in rowsp[:8]:
    print("  %-20s %6d launches %10.2f ms %5.1f%%" % (x["kernel"], x["launches"], x["ms"], 100 * x["ms"] / tot), flush=True)
ctx.close()
if a.check:
    assert want == list(range(a.ranks)), "--check needs --slices all"
    one = pfbwt_hip.PfpContext(w=w, p=p, u64=True, sai=True)
    for h in range(a.H):
        one.feed(big[h], True)
    one.finalize(); one.parse_bwt(); b1 = one.bwt_build(sa=a.sa, rssa=not a.sa); ref = one.bwt_get(); one.close()
    for k in (("bwt", "sa") if a.sa else ("bwt", "ssa", "esa")):
        got = np.concatenate([parts[r][k] for r in range(a.ranks)])
        assert got.shape == ref[k].shape and np.array_equal(got, ref[k]), k
    print("check OK: slices of %d ranks concatenate to the single-context output (%s), r=%d" % (a.ranks, "bwt, sa" if a.sa else "bwt, ssa, esa", b1.r), flush=True)
