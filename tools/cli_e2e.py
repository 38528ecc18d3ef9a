#!/usr/bin/env python3
"""Measurement aid: the COMMAND LINE end to end -- FASTA file in, .bwt / .ssa / .esa (or .sa) files out, one cold process --
on a memory-resident file system, next to a check of what it wrote.
  python tools/cli_e2e.py [--H 1000] [--L 32000000] [--check-H 40] [--dir /dev/shm]
1. check: a collection of --check-H haplotypes goes through pfbwt-f_amd/bin/pfbwt-f64 -r AND through the Python binding
   (pfp_parse_feed per record + pfp_bwt_get); the files must be byte-identical (the CLI reads raw FASTA blocks and strips them on
   the device, and writes its outputs from the device through page-locked blocks: both paths are new in round 3).
2. timing: the full collection through the CLI, wall time of the process and its TASK lines."""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import numpy as np
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--H", type=int, default=1000); ap.add_argument("--L", type=int, default=32_000_000); ap.add_argument("--check-H", type=int, default=40)
ap.add_argument("--dir", default="/dev/shm"); ap.add_argument("--flags", default="-r")
a = ap.parse_args()
exe = os.path.join(ROOT, "pfbwt-f_amd", "bin", "pfbwt-f64")
flags = a.flags.split()


def run_cli(fa, out):
    t0 = time.perf_counter()
    pr = subprocess.run([exe] + flags + ["-w", "10", "-p", "100", "-o", out, fa], capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if pr.returncode != 0:
        raise SystemExit("CLI failed: " + pr.stderr[-2000:])
    return wall, [l for l in pr.stderr.splitlines() if l.startswith(("TASK", "n:", "r:", "read "))]


# 1. check
if a.check_H:
    import pfbwt_hip
    rows = bench.synth_seqs(a.L, a.check_H, 1000, (0, 0, 0, 0))
    fa = os.path.join(a.dir, "cli_check.fa"); out = os.path.join(a.dir, "cli_check")
    bench.write_fasta_image(fa, rows)
    run_cli(fa, out)
    c = pfbwt_hip.PfpContext(w=10, p=100, u64=True, sai=True)
    for r in rows:
        c.feed(r, True)
    c.finalize(); c.parse_bwt(); c.bwt_build(sa="-s" in flags, rssa="-r" in flags)
    o = c.bwt_get(); c.close()
    for k in ("bwt", "sa", "ssa", "esa"):
        if o.get(k) is None:
            continue
        got = np.fromfile(out + "." + k, dtype=o[k].dtype)
        assert got.size == o[k].size and np.array_equal(got, o[k]), "CLI file .%s differs from the binding's output" % k
    print("check ok: CLI files == binding outputs on %d x %d bases (%s)" % (a.check_H, a.L, " ".join(flags)), flush=True)
    for f in os.listdir(a.dir):
        if f.startswith("cli_check"):
            os.remove(os.path.join(a.dir, f))
    del rows, o

# 2. timing at full size
if a.H:
    rows = bench.synth_seqs(a.L, a.H, 1000, (0, 0, 0, 0))
    fa = os.path.join(a.dir, "cli_full.fa"); out = os.path.join(a.dir, "cli_full")
    nbytes = bench.write_fasta_image(fa, rows)
    del rows
    time.sleep(3.0)
    wall, lines = run_cli(fa, out)
    n = a.H * (a.L + 10)
    print("CLI %s on %d x %d bases (%d FASTA bytes, %s): process wall %.2f s = %.2f Gbases/s" % (" ".join(flags), a.H, a.L, nbytes, a.dir, wall, n / wall / 1e9), flush=True)
    for l in lines:
        print("   " + l, flush=True)
    sizes = {e: os.path.getsize(out + "." + e) for e in ("bwt", "ssa", "esa", "sa", "dict", "parse", "ilist", "bwsai", "bwlast") if os.path.exists(out + "." + e)}
    print("   files:", sizes, flush=True)
    for f in os.listdir(a.dir):
        if f.startswith("cli_full"):
            os.remove(os.path.join(a.dir, f))
