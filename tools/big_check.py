#!/usr/bin/env python3
"""Development aid: build BWT+SA of a large synthetic text on the GPU and check size-independent properties
(SA is a permutation of 0..n, T[SA[i]-1] == BWT[i], exactly one 0x00 byte, sampled adjacent suffixes ordered)."""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import pfbwt_hip
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=1_000_000_000); ap.add_argument("--H", type=int, default=1); ap.add_argument("--seed", type=int, default=38)
ap.add_argument("--u64", action="store_true"); ap.add_argument("--reps", type=int, default=1); ap.add_argument("--nrun", type=int, nargs=4, default=[0, 0, 0, 0])
a = ap.parse_args()
lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
w = 10
t0 = time.time(); seqs = []
for h in range(a.H):
    s = np.empty(a.L, np.uint8); lib.pfp_synth_haplotype(a.seed, a.L, h, *a.nrun, s.ctypes.data_as(C.c_void_p)); seqs.append(s)
print("synth %.1fs" % (time.time() - t0), flush=True)
ctx = pfbwt_hip.PfpContext(w=w, p=100, u64=a.u64, sai=True)
for rep in range(a.reps):
    if rep == a.reps - 1 and a.reps > 1:
        ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.time()
    for s in seqs: ctx.feed(s, True)
    t1 = time.time(); sz = ctx.finalize(); t2 = time.time(); ctx.parse_bwt(); t3 = time.time(); b = ctx.bwt_build(sa=True, rssa=False); t4 = time.time()
    print("rep %d: n=%d m=%d dwords=%d dsize=%d r=%d | feed %.2fs parse %.3fs pbwt %.3fs bwt %.3fs -> %.3f Gbases/s" % (rep, sz.n, sz.m, sz.dwords, sz.dsize, b.r, t1 - t0, t2 - t1, t3 - t2, t4 - t3, sz.n / (t4 - t1) / 1e9), flush=True)
if a.reps > 1:
    rows = sorted(ctx.profile(), key=lambda r: -r["ms"]); tot = sum(r["ms"] for r in rows)
    for r in rows[:14]:
        print("  %-20s %6d launches %10.2f ms %5.1f%%" % (r["kernel"], r["launches"], r["ms"], 100 * r["ms"] / tot), flush=True)
out = ctx.bwt_get(); ctx.close()
n = sz.n
T = np.concatenate([np.concatenate([s, np.full(w, ord("A"), np.uint8)]) for s in seqs]); del seqs
sa = out["sa"]; bwt = out["bwt"]
assert int(sa[0]) == n
seen = np.zeros(n + 1, np.bool_); seen[sa] = True
assert seen.all(), "SA is not a permutation"; del seen
mk = sa > 0
idx = sa[mk].astype(np.int64) - 1
assert np.array_equal(T[idx], bwt[mk]), "T[SA-1] != BWT"; del idx
assert (bwt[~mk] == 0).all() and int((~mk).sum()) == 1
rng = np.random.default_rng(1)
Tz = T
for i in rng.integers(1, n + 1, 3000):
    x, y = int(sa[i - 1]), int(sa[i])
    la, lb = bytes(Tz[x:x + 3000]), bytes(Tz[y:y + 3000])
    assert la < lb or (len(la) == 3000 and la == lb) or (la == lb[:len(la)]), (i, x, y)
print("properties OK: permutation, T[SA-1]==BWT, one EOS byte, 3000 sampled neighbours ordered; r=%d" % b.r, flush=True)
