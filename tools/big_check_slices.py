#!/usr/bin/env python3
"""Full-size cross-check that needs no oracle: the S-32G collection (or --H / --L of it) is built
once in ONE context (-r: .bwt, .ssa, .esa left in HBM) and once as N slices (pfp_bwt_build_slice, the multi-GPU emission);
the position-weighted checksums (pfp_debug_checksum: additive over pieces) of the single outputs must equal the sums of
the slices' checksums, and r the sum of the slices' run counts.  Nothing leaves the device."""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import torch
import pfbwt_hip
from bench import synth_seqs
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32_000_000); ap.add_argument("--H", type=int, default=1000); ap.add_argument("--seed", type=int, default=1000)
ap.add_argument("--slices", type=int, default=8)
a = ap.parse_args()
w, U = 10, 8
t0 = time.time()
seqs = synth_seqs(a.L, a.H, a.seed, (0, 0, 0, 0))
d_all = torch.from_numpy(np.ascontiguousarray(seqs[0].base if (a.H > 16 and seqs[0].base is not None) else np.stack(seqs))).to("cuda")
del seqs
print("synth + upload %.0fs" % (time.time() - t0), flush=True)
ctx = pfbwt_hip.PfpContext(w=w, p=100, u64=True, sai=True)
L = ctx.L
L.pfp_debug_checksum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64 * 2)]


def cks(ptr, nbytes, off):
    out = (C.c_uint64 * 2)()
    rc = L.pfp_debug_checksum(ctx.h, C.c_void_p(ptr), nbytes, off, C.byref(out))
    assert rc == 0, rc
    return np.array([out[0], out[1]], np.uint64)


ctx.feed_device_batch(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))
sz = ctx.finalize(); ctx.parse_bwt()
t0 = time.time(); b = ctx.bwt_build(sa=False, rssa=True); t1 = time.time()
pb, _, ps, pe = ctx.bwt_device_ptrs()
single = {"bwt": cks(pb, b.nout, 0), "ssa": cks(ps, 2 * U * b.r, 0), "esa": cks(pe, 2 * U * b.r, 0)}
print("single context: n=%d r=%d  %.3fs  checksums %s" % (sz.n, b.r, t1 - t0, {k: [hex(int(x)) for x in v] for k, v in single.items()}), flush=True)
acc = {k: np.zeros(2, np.uint64) for k in single}
rsum = 0; soff = 0; eoff = 0; rows_seen = 0
with np.errstate(over="ignore"):
    for sl in range(a.slices):
        t0 = time.time()
        bs, beg, rows = ctx.bwt_build_slice(sl, a.slices, sa=False, rssa=True)
        dt = time.time() - t0
        assert beg == rows_seen, (beg, rows_seen)
        pb, _, ps, pe = ctx.bwt_device_ptrs()
        acc["bwt"] += cks(pb, rows, beg)
        acc["ssa"] += cks(ps, 2 * U * bs.r, soff); soff += 2 * U * bs.r
        acc["esa"] += cks(pe, 2 * U * ctx.esa_pairs, eoff); eoff += 2 * U * ctx.esa_pairs
        rsum += bs.r; rows_seen += rows
        print("slice %d/%d: rows [%d, %d) runs starting here %d, esa pairs %d  %.3fs" % (sl, a.slices, beg, beg + rows, bs.r, ctx.esa_pairs, dt), flush=True)
assert rows_seen == b.nout and rsum == b.r and soff == eoff == 2 * U * b.r, (rows_seen, rsum, soff, eoff)
for k in single:
    assert np.array_equal(acc[k], single[k]), (k, acc[k], single[k])
print("OK: .bwt / .ssa / .esa of the single-context build == concatenation of %d slices (checksums of %d + 2 x %d bytes), r = %d" % (a.slices, b.nout, 2 * U * b.r, b.r), flush=True)
ctx.close()
