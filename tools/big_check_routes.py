#!/usr/bin/env python3
"""Full-size cross-check of the routes that round 3 added against the routes they replace (validated at full size in round 2):
a bench workload is built with the default switches (round 4: recursive suffix sort of the parse, rows fed by view) and again with the old routes forced -- the parse sorted by prefix doubling (parse_rec = 0), rows copied in, row-wise emission of the special rows
(emit_group_rows = 0), two-gather slot fields (no_slot_records = 1), rank-based dictionary sort (dict_text_rounds = 0), two parse
symbols in the initial key, run round always, emission windows of 2^30 rows -- and the position-weighted device checksums (pfp_debug_checksum) of every output
(.bwt, .sa if the workload has one, .ssa, .esa) and r must be equal.  Nothing leaves the device; no oracle is involved: small
inputs are compared with the oracle by the test suite under the same switches.
usage: python tools/big_check_routes.py [--workload S-32G | S-3G | S-50G | S-chr22]"""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import torch
import bench, pfbwt_hip

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="S-32G")
a = ap.parse_args()
Lb, H, seed, nruns, w, p, u64 = bench.WORKLOADS[a.workload]
want_sa, want_rssa = bench.outputs_of(a.workload)
U = 8 if u64 else 4
h_all = torch.empty((H, Lb), dtype=torch.uint8, pin_memory=True)
bench.synth_seqs(Lb, H, seed, nruns, out=h_all.numpy())
d_all = h_all.to("cuda"); del h_all
OLD = dict(emit_group_rows=0, no_slot_records=1, dict_text_rounds=0, int_key_symbols=2, force_run_round=1, emit_chunk_rows=1 << 30, parse_rec=0, dict_rec=0, dedup_variant=0, dedup_period=-1)      # parse_rec=0, dict_rec=0 (round 4): parse and dictionary suffix-sorted by prefix doubling, as in rounds 1-3


def build(switches):
    ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=0)
    ctx.debug_set(**switches)
    lib = ctx.L
    lib.pfp_debug_checksum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64 * 2)]
    t0 = time.time()
    if switches: ctx.feed_device_batch(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))      # the old routes: rows copied in
    else: ctx.feed_device_view(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))               # default: rows read in place (round 4)
    sz = ctx.finalize(); ctx.parse_bwt(); b = ctx.bwt_build(sa=want_sa, rssa=want_rssa)
    torch.cuda.synchronize(); dt = time.time() - t0
    pb, psa, ps, pe = ctx.bwt_device_ptrs()
    out = {}
    for name, ptr, nbytes in (("bwt", pb, b.nout), ("sa", psa if want_sa else None, U * b.nout), ("ssa", ps if want_rssa else None, 2 * U * b.r), ("esa", pe if want_rssa else None, 2 * U * b.r)):
        if ptr:
            o = (C.c_uint64 * 2)()
            assert lib.pfp_debug_checksum(ctx.h, C.c_void_p(ptr), nbytes, 0, C.byref(o)) == 0
            out[name] = (int(o[0]), int(o[1]))
    res = (sz.n, sz.m, sz.dwords, sz.dsize, b.nout, b.r, out)
    ctx.close()
    return res, dt


new, t_new = build({})
print("default switches : n=%d m=%d dwords=%d dsize=%d rows=%d r=%d  %.3fs  %s" % (*new[:6], t_new, {k: [hex(x) for x in v] for k, v in new[6].items()}), flush=True)
old, t_old = build(OLD)
print("round-2 routes   : n=%d m=%d dwords=%d dsize=%d rows=%d r=%d  %.3fs  %s" % (*old[:6], t_old, {k: [hex(x) for x in v] for k, v in old[6].items()}), flush=True)
assert new == old, "outputs differ between the routes"
print("OK: %s -- every output of the default build == the build with %s (checksums over %d rows, r = %d)" % (a.workload, OLD, new[4], new[5]), flush=True)
