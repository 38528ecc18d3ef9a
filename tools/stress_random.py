#!/usr/bin/env python3
"""One-off stress (not part of the test suite): the seeded random differential sweep of tests/pfp_testlib.random_cases for a
range of seeds, engine (real library, or PFBWT_HIP_LIB / --emu) against the oracle, optionally under forced-route
environments (each environment in a child process, the switches are read once per process).
usage: python tools/stress_random.py --seeds 1000 1040 [--count 25] [--emu] [--envs]"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
ENVS = [{}, {"PFP_CLASS_SORT_MAXRANGE": "40"}, {"PFP_SORT_NO_TABLE": "1", "PFP_FORCE_WIDE_ROWS": "1"}, {"PFP_SORT_K": "1", "PFP_EMIT_CHUNK_ROWS": "777", "PFP_SAMPLE_CAP": "40"},
        {"PFP_BIG_GROUP_MEMBERS": "1", "PFP_DEDUP_TABLE_LOG2": "5", "PFP_NO_TRIGGER_TABLE": "1"}, {"PFP_NO_RUNAWARE": "1", "PFP_EMIT_CHUNK_ROWS": "3000", "PFP_FILL_SUBS": "1"},
        {"PFP_EMIT_GROUP_ROWS": "0", "PFP_NO_SLOT_RECORDS": "1", "PFP_DICT_TEXT_ROUNDS": "0", "PFP_INT_KEY_SYMBOLS": "2", "PFP_FORCE_RUN_ROUND": "1"},      # round 2's routes: row-wise special rows, two-gather slots, rank-based dictionary sort, two-symbol parse keys, run round always
        {"PFP_EMIT_GROUP_ROWS": "24", "PFP_DICT_TEXT_ROUNDS": "1", "PFP_EMIT_CHUNK_ROWS": "2000"},   # tiny batches (most groups through the LDS radix sort or left to the row-wise kernel), text rounds forced
        {"PFP_PARSE_REC": "1"},                                                                      # round 4: the parse suffix-sorted through its level-2 prefix-free parse (recsort.h), whatever its size
        {"PFP_PARSE_REC": "1", "PFP_PARSE_REC_P2": "3", "PFP_PARSE_REC_TILE_ROWS": "30", "PFP_PARSE_REC_DEPTH": "2", "PFP_EMIT_CHUNK_ROWS": "5000"},
        {"PFP_DICT_REC": "1"},                                                                       # round 4: the dictionary suffix-sorted through its own level-2 parse (dictrec.h), whatever its size
        {"PFP_DICT_REC": "1", "PFP_DICT_REC_P2": "5", "PFP_PARSE_REC_TILE_ROWS": "20", "PFP_PARSE_REC": "1"},
        {"PFP_DEDUP_VARIANT": "0", "PFP_DEDUP_PERIOD": "-1"}, {"PFP_DEDUP_VARIANT": "1", "PFP_DEDUP_PERIOD": "3", "PFP_DEDUP_CHUNK": "2"}]   # short level-2 phrases, assembly batches of 20 rows (larger classes through the global sort), both recursive sorts   # two levels, assembly batches of 30 rows (larger classes through the global sort)
ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, nargs=2, default=[1000, 1010]); ap.add_argument("--count", type=int, default=25)
ap.add_argument("--emu", action="store_true"); ap.add_argument("--envs", action="store_true"); ap.add_argument("--child", action="store_true")
ap.add_argument("--medium", type=int, default=0, help="instead of the small cases: this many synthetic panels of 0.5-10 Mbase (bench generator: variant sites with skewed allele frequencies, optional N runs)")
a = ap.parse_args()
if a.envs and not a.child:
    for e in ENVS:
        env = dict(os.environ); env.update(e); env["PFP_TEST_HOOKS"] = "1"
        t0 = time.time()
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--seeds", str(a.seeds[0]), str(a.seeds[1]), "--count", str(a.count), "--medium", str(a.medium), "--child"] + (["--emu"] if a.emu else []), env=env, capture_output=True, text=True)
        print("%s: rc=%d %.0fs %s" % (e or "default", pr.returncode, time.time() - t0, pr.stdout.strip().splitlines()[-1] if pr.stdout.strip() else pr.stderr[-800:]), flush=True)
        if pr.returncode: sys.exit(1)
    sys.exit(0)
import pfbwt_hip
from pfp_testlib import EMU_SO, check_random
factory = (lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw)) if a.emu else (lambda **kw: pfbwt_hip.PfpContext(**kw))
if a.medium:
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from pfp_testlib import compare, engine_run, oracle_run
    rng = np.random.default_rng(a.seeds[0])
    for ci in range(a.medium):
        H = int(rng.integers(2, 120)); L = int(rng.integers(20_000, 400_000)); L = min(L, 10_000_000 // H)
        w = int(rng.choice([4, 6, 8, 10, 10, 10, 12])); p = int(rng.choice([7, 20, 50, 100, 100])); U = int(rng.choice([4, 8]))
        nruns = (int(rng.integers(0, L // 2)), int(rng.integers(0, 5000)), 0, 0) if rng.random() < 0.4 else (0, 0, 0, 0)
        seqs = [bytes(x) for x in bench.synth_seqs(L, H, int(rng.integers(1, 1 << 30)), nruns)]
        sa, rssa = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        ref = oracle_run(seqs, w=w, p=p, U=U)
        res = engine_run(factory, seqs, w, p, U, sa=sa, rssa=rssa)
        names = ["dict", "occ", "parse", "last", "sai", "bwlast", "ilist", "bwsai", "bwt"] + (["sa"] if sa else []) + (["ssa", "esa"] if rssa else [])
        bad = compare(res, ref, U, names=tuple(names))
        assert bad == [] and res["r"] == ref["r"], (ci, H, L, w, p, U, nruns, bad)
    print("ok: %d medium panels (0.5-10 Mbase), seed %d" % (a.medium, a.seeds[0]))
    sys.exit(0)
n = 0
for seed in range(a.seeds[0], a.seeds[1]):
    check_random(factory, seed, a.count); n += a.count
print("ok: %d random cases, seeds [%d, %d)" % (n, a.seeds[0], a.seeds[1]))
