#!/usr/bin/env python3
"""Development aid: pfp_bwt_get_expanded (.bwt from its runs into host memory) -- threads only vs threads + copy engine, thread counts."""
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
ctx = pfbwt_hip.PfpContext(w=10, p=100, u64=True, sai=True)
ctx.feed_device_view(d.data_ptr(), H, L0, L0)
sz = ctx.finalize(); ctx.parse_bwt(); b = ctx.bwt_build(sa=False, rssa=True)
ssa, esa = ctx.samples_get()
n1 = sz.n + 1
print("n=%d r=%d cpus=%d" % (sz.n, b.r, len(os.sched_getaffinity(0))), flush=True)
ref = None
for pinned in (True, False):
    hb = np.empty(n1, np.uint8); hb.fill(0)
    if pinned: assert ctx.L.pfp_host_register(hb.ctypes.data, hb.size) == 0
    for dma in (0, 1):
        for th in (0, 8, 16, 32):
            ctx.debug_set(expand_dma=dma)
            best = 1e9
            for rep in range(2):
                t0 = time.perf_counter(); ctx.bwt_get_expanded(hb.ctypes.data, ssa, threads=th); best = min(best, time.perf_counter() - t0)
            s = int(hb[::4097].astype(np.uint64).sum())
            if ref is None: ref = s
            print("pinned=%d dma=%d threads=%2d: %.1f ms  %.1f GB/s  %s" % (pinned, dma, th, 1e3 * best, n1 / best / 1e9, "ok" if s == ref else "MISMATCH"), flush=True)
    if pinned: ctx.L.pfp_host_unregister(hb.ctypes.data)
# plain DMA of the rows for comparison
hb = np.empty(n1, np.uint8); hb.fill(0); assert ctx.L.pfp_host_register(hb.ctypes.data, hb.size) == 0
t0 = time.perf_counter(); ctx.bwt_get(out={"bwt": hb, "ssa": ssa, "esa": esa}); dt = time.perf_counter() - t0
print("pfp_bwt_get (all rows + samples over the link): %.1f ms" % (1e3 * dt))
