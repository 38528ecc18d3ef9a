#!/usr/bin/env python3
"""Measurement aid: the file reader (pfp_parse_feed_fasta_file) in a FRESH process versus warm, with and without helper threads
that fault host memory in at the same time (what bench.py's end-to-end child does for its output buffers).
  python tools/ingest_bench.py [--H 250] [--L 32000000]"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
ap = argparse.ArgumentParser(); ap.add_argument("--H", type=int, default=250); ap.add_argument("--L", type=int, default=32_000_000)
ap.add_argument("--child", default=""); ap.add_argument("--prefault-threads", type=int, default=0); ap.add_argument("--image", default="")
a = ap.parse_args()
if a.child:
    t00 = time.perf_counter()
    import threading
    import numpy as np
    import pfbwt_hip
    th = None
    if a.prefault_threads:
        def pf():
            from concurrent.futures import ThreadPoolExecutor
            buf = np.empty(os.path.getsize(a.image), np.uint8); step = 1 << 28
            with ThreadPoolExecutor(max_workers=a.prefault_threads) as ex:
                list(ex.map(lambda i: buf[i:i + step].fill(0), range(0, buf.size, step)))
        th = threading.Thread(target=pf); th.start()
    ctx = pfbwt_hip.PfpContext(w=10, p=100, u64=True, sai=True)
    out = []
    for it in range(3):
        t0 = time.perf_counter(); info = ctx.feed_fasta_file(a.image); t1 = time.perf_counter()
        out.append({"ingest_ms": round(1e3 * (t1 - t0), 1), "GBps": round(info.raw_bytes / (t1 - t0) / 1e9, 1), "reader_wait_ms": round(info.read_wait_ms, 1)})
        ctx.reset()
        if th is not None and it == 0:
            th.join()
    print(json.dumps({"variant": a.child, "startup_ms": round(1e3 * (time.perf_counter() - t00 - sum(o["ingest_ms"] for o in out) / 1e3), 1), "runs": out}), flush=True)
    sys.exit(0)
import numpy as np
import bench
rows = bench.synth_seqs(a.L, a.H, 1000, (0, 0, 0, 0))
img = "/dev/shm/ingest_bench_%d.fa" % os.getpid()
n = bench.write_fasta_image(img, rows); del rows
print("image %d bytes" % n, flush=True)
for variant, k in (("no helper threads", 0), ("6 helper threads faulting memory in", 6), ("16 helper threads", 16), ("no helper threads, again", 0)):
    time.sleep(4.0)
    pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", variant, "--prefault-threads", str(k), "--image", img], capture_output=True, text=True)
    print(pr.stdout.strip().splitlines()[-1] if pr.returncode == 0 and pr.stdout.strip() else "FAILED " + pr.stderr[-500:], flush=True)
os.remove(img)
