#!/usr/bin/env python3
"""Development aid: BWT + run-length sampled SA (-r) of a very large synthetic collection (north-star shape:
H haplotypes x L bases) on one GPU, checked by properties that need no full SA:
  * byte histogram of the BWT == histogram of the text (+ one 0x00),
  * r == number of positions where the BWT byte changes,
  * for every run-start / run-end sample (row, sa): T[sa-1] == BWT[row]; sample rows are exactly the run boundaries,
  * the SA values of the samples are distinct."""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import pfbwt_hip
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32_000_000); ap.add_argument("--H", type=int, default=1000); ap.add_argument("--seed", type=int, default=1000)
ap.add_argument("--reps", type=int, default=1); ap.add_argument("--no-check", action="store_true")
a = ap.parse_args()
lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
w = 10
t0 = time.time()
T = np.empty(a.H * (a.L + w), np.uint8)       # the text itself: haplotype h, then w 'A's
for h in range(a.H):
    o = h * (a.L + w)
    lib.pfp_synth_haplotype(a.seed, a.L, h, 0, 0, 0, 0, T[o:o + a.L].ctypes.data_as(C.c_void_p)); T[o + a.L:o + a.L + w] = ord("A")
    if h % 100 == 0: print("synth %d/%d %.0fs" % (h, a.H, time.time() - t0), flush=True)
ctx = pfbwt_hip.PfpContext(w=w, p=100, u64=True, sai=True)
for rep in range(a.reps):
    if rep == a.reps - 1: ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.time()
    for h in range(a.H):
        o = h * (a.L + w); ctx.feed(T[o:o + a.L], True)
    t1 = time.time(); sz = ctx.finalize(); t2 = time.time(); print("parsed %.2fs" % (t2 - t1), flush=True)
    ctx.parse_bwt(); t3 = time.time(); print("parse-bwt %.2fs" % (t3 - t2), flush=True)
    b = ctx.bwt_build(sa=False, rssa=True); t4 = time.time()
    print("rep %d: n=%d m=%d dwords=%d dsize=%d r=%d n/r=%.1f | feed %.2fs parse %.3fs pbwt %.3fs bwt %.3fs -> %.3f Gbases/s (device stages)"
          % (rep, sz.n, sz.m, sz.dwords, sz.dsize, b.r, sz.n / b.r, t1 - t0, t2 - t1, t3 - t2, t4 - t3, sz.n / (t4 - t1) / 1e9), flush=True)
rows = sorted(ctx.profile(), key=lambda r: -r["ms"]); tot = sum(r["ms"] for r in rows)
for r in rows[:14]: print("  %-20s %6d launches %10.2f ms %5.1f%%" % (r["kernel"], r["launches"], r["ms"], 100 * r["ms"] / tot), flush=True)
if a.no_check: ctx.close(); sys.exit(0)
out = ctx.bwt_get(); ctx.close()
n = sz.n; assert n == T.size
bwt = out["bwt"]; ssa = out["ssa"].reshape(-1, 2); esa = out["esa"].reshape(-1, 2)
CH = 1 << 28
hb = np.zeros(256, np.int64); ht = np.zeros(256, np.int64); starts = []
for s in range(0, n + 1, CH):                      # chunked: numpy promotes uint8 -> int64 inside bincount
    e = min(n + 1, s + CH); seg = bwt[s:e]
    hb += np.bincount(seg, minlength=256)
    if s < n: ht += np.bincount(T[s:min(n, e)], minlength=256)
    prev = bwt[s - 1] if s else 0
    d = np.flatnonzero(np.concatenate(([seg[0] != prev], seg[1:] != seg[:-1]))) + s; starts.append(d)
ht[0] += 1
assert np.array_equal(hb, ht), "BWT is not a permutation of the text"
print("histogram ok", flush=True)
starts = np.concatenate(starts)
assert starts.size == b.r == ssa.shape[0] == esa.shape[0], (starts.size, b.r)
assert np.array_equal(ssa[:, 0].astype(np.int64), starts) and np.array_equal(esa[:, 0].astype(np.int64), np.concatenate((starts[1:] - 1, [n])))
for nm, arr in (("ssa", ssa), ("esa", esa)):
    row, sa = arr[:, 0].astype(np.int64), arr[:, 1].astype(np.int64)
    assert (sa >= 0).all() and (sa <= n).all()
    mk = sa > 0
    assert np.array_equal(T[sa[mk] - 1], bwt[row[mk]]), nm + ": T[sa-1] != BWT[row]"
    assert (bwt[row[~mk]] == 0).all()
    assert np.unique(sa).size == sa.size, nm + ": SA samples not distinct"
assert ssa[0, 0] == 0 and ssa[0, 1] == n
print("properties OK: histogram, r == #run starts, sample rows == run boundaries, T[sa-1]==BWT[row] for all %d + %d samples" % (ssa.shape[0], esa.shape[0]), flush=True)
