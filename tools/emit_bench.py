#!/usr/bin/env python3
"""Emission stage only of a bench workload (parse and parse-BWT run once), kernel by kernel from the library's HIP-event
profile: for timing experiments on the emission kernels.  usage: python tools/emit_bench.py [--workload S-32G] [--set key=value ...] [--lib other.so]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import torch
import bench, pfbwt_hip

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="S-32G"); ap.add_argument("--set", nargs="*", default=[], help="key=value tunables"); ap.add_argument("--lib", default=None)
a = ap.parse_args()
L, H, seed, nruns, w, p, u64 = bench.WORKLOADS[a.workload]
want_sa, want_rssa = bench.outputs_of(a.workload)
h_all = torch.empty((H, L), dtype=torch.uint8, pin_memory=True)
bench.synth_seqs(L, H, seed, nruns, out=h_all.numpy())
d_all = h_all.to("cuda"); del h_all
ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=0, **({"lib": a.lib} if a.lib else {}))
for kv in a.set:
    k, v = kv.split("="); ctx.debug_set(**{k: int(v)})
ctx.feed_device_batch(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))
ctx.finalize(); ctx.parse_bwt()
print("parsed", flush=True)
for ab in (0,):
    for rep in range(3):
        ctx.profile_enable(True); ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.bwt_build(sa=want_sa, rssa=want_rssa)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rows = sorted(ctx.profile(), key=lambda r: -r["ms"])
        if rep:
            print("emission %.1f ms; " % (1e3 * dt) + ", ".join("%s %.2f" % (r["kernel"], r["ms"]) for r in rows[:9]), flush=True)
