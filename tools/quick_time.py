#!/usr/bin/env python3
"""Development aid: time the engine stage by stage on a synthetic input and print the kernel table."""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import pfbwt_hip

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=5_000_000); ap.add_argument("--H", type=int, default=1)
ap.add_argument("--seed", type=int, default=22); ap.add_argument("--nrun", type=int, nargs=4, default=[0, 0, 0, 0])
ap.add_argument("--reps", type=int, default=2); ap.add_argument("--u64", action="store_true")
a = ap.parse_args()
lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
seqs = []
for h in range(a.H):
    s = np.empty(a.L, np.uint8); lib.pfp_synth_haplotype(a.seed, a.L, h, *a.nrun, s.ctypes.data_as(C.c_void_p)); seqs.append(s)
ctx = pfbwt_hip.PfpContext(w=10, p=100, u64=a.u64, sai=True)
for rep in range(a.reps):
    if rep == a.reps - 1:
        ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.time()
    for s in seqs: ctx.feed(s, True)
    t1 = time.time(); sz = ctx.finalize(); t2 = time.time(); ctx.parse_bwt(); t3 = time.time(); b = ctx.bwt_build(sa=True, rssa=False); t4 = time.time()
    print("rep %d: n=%d m=%d dwords=%d dsize=%d r=%d | feed %.1f ms  parse %.1f ms  pbwt %.1f ms  bwt %.1f ms  total(no feed) %.1f ms -> %.3f Gbases/s"
          % (rep, sz.n, sz.m, sz.dwords, sz.dsize, b.r, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t1), sz.n / (t4 - t1) / 1e9), flush=True)
rows = sorted(ctx.profile(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows)
print("kernel                launches      ms    %%   GB/s(alg)")
for r in rows:
    print("%-20s %8d %8.3f %5.1f %8.1f" % (r["kernel"], r["launches"], r["ms"], 100 * r["ms"] / tot, r["bytes"] / max(r["ms"], 1e-9) / 1e6))
print("sum kernel ms %.3f" % tot)
