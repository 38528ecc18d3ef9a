#!/usr/bin/env python3
"""One-off stress (not part of the test suite): the seeded random differential sweep of tests/pfp_testlib.random_cases for a
range of seeds, engine (real library, or PFBWT_HIP_LIB / --emu) against the oracle, optionally under forced-route
environments (each environment in a child process, the switches are read once per process).
usage: python tools/stress_random.py --seeds 1000 1040 [--count 25] [--emu] [--envs]"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
ENVS = [{}, {"PFP_CLASS_SORT_MAXRANGE": "40"}, {"PFP_SORT_NO_TABLE": "1", "PFP_FORCE_WIDE_ROWS": "1"}, {"PFP_SORT_K": "1", "PFP_EMIT_CHUNK_ROWS": "777", "PFP_SAMPLE_CAP": "40"},
        {"PFP_BIG_GROUP_MEMBERS": "1", "PFP_DEDUP_TABLE_LOG2": "5", "PFP_NO_TRIGGER_TABLE": "1"}, {"PFP_NO_RUNAWARE": "1", "PFP_EMIT_CHUNK_ROWS": "3000", "PFP_FILL_SUBS": "1"}]
ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, nargs=2, default=[1000, 1010]); ap.add_argument("--count", type=int, default=25)
ap.add_argument("--emu", action="store_true"); ap.add_argument("--envs", action="store_true"); ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if a.envs and not a.child:
    for e in ENVS:
        env = dict(os.environ); env.update(e)
        t0 = time.time()
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--seeds", str(a.seeds[0]), str(a.seeds[1]), "--count", str(a.count), "--child"] + (["--emu"] if a.emu else []), env=env, capture_output=True, text=True)
        print("%s: rc=%d %.0fs %s" % (e or "default", pr.returncode, time.time() - t0, pr.stdout.strip().splitlines()[-1] if pr.stdout.strip() else pr.stderr[-800:]), flush=True)
        if pr.returncode: sys.exit(1)
    sys.exit(0)
import pfbwt_hip
from pfp_testlib import EMU_SO, check_random
factory = (lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw)) if a.emu else (lambda **kw: pfbwt_hip.PfpContext(**kw))
n = 0
for seed in range(a.seeds[0], a.seeds[1]):
    check_random(factory, seed, a.count); n += a.count
print("ok: %d random cases, seeds [%d, %d)" % (n, a.seeds[0], a.seeds[1]))
