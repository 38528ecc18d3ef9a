#!/usr/bin/env python3
"""Development aid: time of the fused feed + trigger scan (pfp_parse_feed_device_view) for aligned / unaligned row geometries."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python")); sys.path.insert(0, ROOT)
import pfbwt_hip, bench
H = int(sys.argv[1]) if len(sys.argv) > 1 else 250
L0 = 32_000_000
rows = torch.empty((H, L0), dtype=torch.uint8, pin_memory=True)
bench.synth_seqs(L0, H, 1000, (0, 0, 0, 0), out=rows.numpy())
d = rows.to("cuda")
for (length, what) in ((L0, "rows of 32 000 000 (+ 10 pad: every row shifts by 10 bytes)"), (L0 - 10, "rows of 31 999 990 (+ 10 pad = a multiple of 16: aligned)")):
    for mode in ("view", "batch"):
        ctx = pfbwt_hip.PfpContext(w=10, p=100, u64=True, sai=True)
        for rep in range(2):
            if rep == 1: ctx.profile_enable(True); ctx.profile_reset()
            if mode == "view": ctx.feed_device_view(d.data_ptr(), H, length, L0)
            else: ctx.feed_device_batch(d.data_ptr(), H, length, L0)
            sz = ctx.finalize(shard=True)
        prof = {r["kernel"]: r["ms"] for r in ctx.profile()}
        print("%s | %s: trigger_scan %.2f ms, misc (feed copy) %.2f ms, n=%d" % (what, mode, prof.get("trigger_scan", 0), prof.get("misc", 0), sz.n), flush=True)
        ctx.close()
