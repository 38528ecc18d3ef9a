#!/usr/bin/env python3
"""Parse stage only (feed + finalize) of a bench workload, kernel by kernel from the library's own HIP-event profile.
For experiments on the trigger scan / de-duplication kernels: nothing behind the parse runs, so a build that computes
wrong occurrence counts on purpose (timing experiments) cannot reach the emission.
usage: python tools/parse_bench.py [--workload S-32G] [--reps 2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import torch
import bench, pfbwt_hip

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="S-32G"); ap.add_argument("--reps", type=int, default=2); ap.add_argument("--variants", type=int, nargs="*", default=None, help="k_dedup_insert<VAR> variants, one rep each, stages timed inside the kernel (PFP_TEST_HOOKS=1)")
a = ap.parse_args()
L, H, seed, nruns, w, p, u64 = bench.WORKLOADS[a.workload]
h_all = torch.empty((H, L), dtype=torch.uint8, pin_memory=True)
bench.synth_seqs(L, H, seed, nruns, out=h_all.numpy())
d_all = h_all.to("cuda")
ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=0)
reps = a.reps + 1 if a.variants is None else len(a.variants) + 1
for rep in range(reps):
    if a.variants is not None: v = a.variants[max(rep - 1, 0)]; ctx.debug_set(dedup_variant=v % 10, dedup_phases=(v // 10) % 10, dedup_period=-1 if v >= 1000 else 0)      # 1x: with the in-kernel stage timing; 1xxx: text order
    ctx.profile_enable(True); ctx.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.feed_device_batch(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))
    try:
        ctx.finalize()
    except pfbwt_hip.PfpError as e:      # timing experiments (PFP_EXP) leave the table unusable on purpose
        print("finalize:", e)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows = sorted(ctx.profile(), key=lambda r: -r["ms"])
    if rep:
        print("rep %d%s: parse %.1f ms; " % (rep, "" if a.variants is None else " variant %d" % a.variants[rep - 1], 1e3 * dt) + ", ".join("%s %.1f" % (r["kernel"], r["ms"]) for r in rows[:8]), flush=True)
    ctx.reset()
