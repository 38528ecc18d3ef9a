"""GPU parity of the instantiations the S-32G headline is timed on: 64-bit row counters
(k_emit<u64,u64>, k_samples_tile<u64,u64>), emission in windows of rows, the one-pass sample route and its exact
two-pass fallback, the sort route for groups with many members -- forced on inputs small enough for the oracle through
the same environment switches bench.py's parity leg uses, on the PRODUCT library (hipcc, gfx950), in -s, -r and
sliced mode.  Semantics under test: include/pfbwt.hpp:96-194, src/pfbwt-f.cpp:298-328."""
import os
import subprocess
import sys
import pytest
from pfp_testlib import ROOT

pytestmark = pytest.mark.gpu

ROUTE_CODE = r'''
import sys, os
sys.path.insert(0, sys.argv[1] + "/tests")
from pfp_testlib import *
from test_gpu_parity import synth
import numpy as np
import pfbwt_hip
lib = pfbwt_hip.load_library()
assert lib.pfp_backend().decode() == "hip-gfx950"
F = lambda **kw: pfbwt_hip.PfpContext(**kw)
cases = [(synth(31, 120000, 6), 10, 100),           # a panel: multi-member groups, several windows at the forced chunk sizes
         (synth(32, 40000, 12), 4, 11)]             # short phrases, many words per group (sort route when forced)
for ci, (seqs, w, p) in enumerate(cases):
    for U in (8, 4):
        ref = oracle_run(seqs, w=w, p=p, U=U)
        for sa, rssa in ((True, True), (False, True), (True, False), (False, False)):
            res = engine_run(F, seqs, w, p, U, sa=sa, rssa=rssa)
            names = ["bwt"] + (["sa"] if sa else []) + (["ssa", "esa"] if rssa else [])
            bad = compare(res, ref, U, names=tuple(names))
            assert bad == [] and res["r"] == ref["r"], (ci, U, sa, rssa, bad)
        # streamed build (pfp_bwt_build_stream): every window's rows leave for host memory while the next window is emitted; a window
        # re-writes the last row of its predecessor, so the copies are shifted by one row (ADVICE r3) -- with the forced window sizes of
        # these environments (1 000 ... 100 000 rows) groups straddle most window boundaries
        for sa, rssa in ((False, True), (True, False), (False, False)):
            c = F(w=w, p=p, u64=(U == 8), sai=True)
            for s in seqs: c.feed(s, True)
            c.finalize(); c.parse_bwt()
            hb = np.full(ref["n"] + 1, 0xEE, np.uint8); hs = np.full(ref["n"] + 1, 0xEEEEEEEE, np.uint64 if U == 8 else np.uint32)
            b = c.bwt_build_stream(hb.ctypes.data, hs.ctypes.data if sa else None, rssa=rssa)
            assert np.array_equal(hb, ref["bwt"]) and b.r == ref["r"], (ci, U, sa, rssa, "streamed bwt")
            if sa: assert np.array_equal(hs.astype(np.uint64), ref["sa"]), (ci, U, "streamed sa")
            if rssa:
                ssa, esa = c.samples_get()
                assert np.array_equal(ssa.astype(np.uint64), ref["ssa"]) and np.array_equal(esa.astype(np.uint64), ref["esa"]), (ci, U, sa, "streamed samples")
            c.close()
        # sliced mode (multi-GPU emission): concatenated slices == the single-context output
        for ns, sa in ((3, True), (5, False)):
            parts = {"bwt": [], "sa": [], "ssa": [], "esa": []}; r = 0
            for sl in range(ns):
                c = F(w=w, p=p, u64=(U == 8), sai=True)
                for s in seqs: c.feed(s, True)
                c.finalize(); c.parse_bwt()
                b, beg, rows = c.bwt_build_slice(sl, ns, sa=sa, rssa=True)
                o = c.bwt_get(); c.close()
                assert beg == sum(len(x) for x in parts["bwt"]) and rows == len(o["bwt"])
                for k in parts:
                    if o.get(k) is not None: parts[k].append(o[k])
                r += b.r
            res = {k: np.concatenate(v) for k, v in parts.items() if v}; res["r"] = r
            bad = compare(res, ref, U, names=("bwt", "sa", "ssa", "esa") if sa else ("bwt", "ssa", "esa"))
            assert bad == [], (ci, U, ns, sa, bad)
print("routes ok")
'''

ENVS = [
    {"PFP_FORCE_WIDE_ROWS": "1"},
    {"PFP_FORCE_WIDE_ROWS": "1", "PFP_EMIT_CHUNK_ROWS": "100000"},
    {"PFP_FORCE_WIDE_ROWS": "1", "PFP_EMIT_CHUNK_ROWS": "1000"},
    {"PFP_EMIT_CHUNK_ROWS": "7777", "PFP_SAMPLE_CAP": "40"},
    {"PFP_BIG_GROUP_MEMBERS": "1"},
    {"PFP_BIG_GROUP_MEMBERS": "2", "PFP_FORCE_WIDE_ROWS": "1", "PFP_EMIT_CHUNK_ROWS": "50000"},
    {"PFP_DEDUP_TABLE_LOG2": "6", "PFP_NO_TRIGGER_TABLE": "1"},      # the phrase table overflows and is rebuilt; the trigger test by hashing every window
    {"PFP_NO_RUNAWARE": "1", "PFP_EMIT_CHUNK_ROWS": "30000"},       # -r with every row enumerated (the route a full SA takes)
    {"PFP_EMIT_GROUP_ROWS": "0", "PFP_EMIT_CHUNK_ROWS": "50000", "PFP_NO_SLOT_RECORDS": "1"},
    {"PFP_DICT_TEXT_ROUNDS": "0", "PFP_INT_KEY_SYMBOLS": "2", "PFP_FORCE_RUN_ROUND": "1"},                                  # dictionary suffix sort: rank-based rounds only
    {"PFP_DICT_TEXT_ROUNDS": "1", "PFP_CLASS_SORT_MAXRANGE": "150"},   # text rounds forced (given up on repetitive inputs: second sort), large classes through the global sort   # special rows: all through the row-wise kernel (the group-stationary kernel off)
    {"PFP_EMIT_GROUP_ROWS": "40", "PFP_EMIT_CHUNK_ROWS": "20000", "PFP_FORCE_WIDE_ROWS": "1"},   # batches of at most 40 rows: most groups are left to the row-wise kernel, the rest goes through LDS
    {"PFP_PARSE_REC": "1"},                                                                      # suffix sort of the parse through its own level-2 prefix-free parse (recsort.h), the route S-32G takes by itself
    {"PFP_PARSE_REC": "1", "PFP_PARSE_REC_P2": "3", "PFP_PARSE_REC_TILE_ROWS": "40", "PFP_PARSE_REC_DEPTH": "2", "PFP_FORCE_WIDE_ROWS": "1"},   # two levels, assembly batches of 40 rows, larger classes through the global sort
    {"PFP_DICT_REC": "1"},                                                                       # suffix sort of the dictionary through its own level-2 parse (dictrec.h), the route S-32G takes by itself
    {"PFP_DICT_REC": "1", "PFP_DICT_REC_P2": "5", "PFP_PARSE_REC_TILE_ROWS": "40", "PFP_PARSE_REC": "1", "PFP_EMIT_CHUNK_ROWS": "50000"},      # short level-2 phrases, small assembly batches, both recursive sorts
    {"PFP_DICT_REC": "0", "PFP_PARSE_REC": "0", "PFP_DEDUP_VARIANT": "0", "PFP_DEDUP_PERIOD": "-1"},   # the routes of rounds 1-3 (every lane reads its own representative, workgroups in text order)
    {"PFP_DEDUP_VARIANT": "1", "PFP_DEDUP_PERIOD": "3", "PFP_DEDUP_CHUNK": "2"},                                            # text de-duplication: workgroups visit the text as a matrix of 3 loci per sequence, columns of 2 per XCD
    {"PFP_DEDUP_PERIOD": "1", "PFP_DEDUP_TABLE_LOG2": "5", "PFP_DEDUP_VARIANT": "0"},             # one locus per sequence; a first table that overflows
    {"PFP_DEDUP_VARIANT": "1", "PFP_DEDUP_TABLE_LOG2": "5"},                                      # the cooperative kernel on both tables (new entries of a wave counted together)
]


@pytest.mark.parametrize("env", ENVS, ids=lambda e: ",".join("%s=%s" % (k.replace("PFP_", ""), v) for k, v in e.items()))
def test_forced_emission_routes_gpu(env):
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"      # the engine reads PFP_* switches only under PFP_TEST_HOOKS=1
    pr = subprocess.run([sys.executable, "-c", ROUTE_CODE, ROOT], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "routes ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]
