"""The recursive suffix sort of the parse (pfbwt-f_amd/csrc/recsort.h: one level of prefix-free parsing of the parse itself, the
data-parallel restatement of SACA-K's recursion, gsa/gsacak.c:1397-1526) against the oracle's SA-IS (oracle/pfp_oracle.c, itself
checked against the reference's sacak in tests/test_oracle_golden.py), through the sacak_int drop-in (gsa/gsacak.h:88) and
through the whole pipeline.  Forced routes on small inputs: the route itself (PFP_PARSE_REC=1), other moduli, assembly batches of
5 / 40 rows (classes with more rows take the global sort), two and three levels, a phrase table that overflows (-> prefix
doubling), inputs whose level-2 phrases are too long (-> prefix doubling).  CPU: tests/emu with poisoned memory; GPU: the product
library, plus one parse long enough to take the route by itself."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest
from pfp_testlib import EMU_SO, ROOT

CODE = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1] + "/tests")
from pfp_testlib import oracle, EMU_SO
import pfbwt_hip
lib = EMU_SO if sys.argv[2] == "emu" else None
if lib is None: assert pfbwt_hip.load_library().pfp_backend().decode() == "hip-gfx950"
scale = int(sys.argv[3])
def check(s, k, tag):
    SA, rounds = pfbwt_hip.sacak_int(s, k, lib=lib)
    want = np.zeros(len(s), np.uint64)
    assert oracle().orc_sais_int(s.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), len(s), k) == 0
    assert np.array_equal(SA.astype(np.uint64), want), (tag, len(s), k)
rng = np.random.default_rng(int(sys.argv[4]))
def panel(L, H, sigma, mut):
    base = rng.integers(1, sigma, L).astype(np.uint32); rows = []
    for h in range(H):
        r = base.copy(); m = rng.random(L) < mut; r[m] = rng.integers(1, sigma, int(m.sum())); rows.append(r)
    return np.concatenate(rows + [np.zeros(1, np.uint32)])
for (L, H, sigma, mut) in ((50, 4, 20, 0.05), (300, 20, 50, 0.02), (100 * scale, 30, 300, 0.01), (40, 100, 9, 0.03), (200 * scale, 8, 5, 0.01)):
    check(panel(L, H, sigma, mut), sigma, "panel")
for n, k in ((10, 3), (1000, 5), (500 * scale, 300), (700 * scale, 3), (3000, 2)):
    s = rng.integers(1, k, n).astype(np.uint32); s[-1] = 0; check(s, k, "random")
check(np.concatenate([np.tile(np.array([3, 1, 2], np.uint32), 500), [0]]).astype(np.uint32), 4, "periodic")
check(np.concatenate([np.full(900, 7, np.uint32), rng.integers(1, 9, 300).astype(np.uint32), np.full(1500, 7, np.uint32), [0]]).astype(np.uint32), 9, "runs")
print("recsort ok")
'''

ENVS = [
    {"PFP_PARSE_REC": "1"},
    {"PFP_PARSE_REC": "1", "PFP_PARSE_REC_P2": "2"},
    {"PFP_PARSE_REC": "1", "PFP_PARSE_REC_P2": "7", "PFP_PARSE_REC_TILE_ROWS": "5"},
    {"PFP_PARSE_REC": "1", "PFP_PARSE_REC_P2": "3", "PFP_PARSE_REC_TILE_ROWS": "40", "PFP_PARSE_REC_DEPTH": "3"},
    {"PFP_PARSE_REC": "1", "PFP_PARSE_REC_TABLE_LOG2": "5"},          # the level-2 phrase table overflows: prefix doubling
    {"PFP_PARSE_REC": "0"},                                           # the route of rounds 1-3
]
IDS = lambda e: ",".join("%s=%s" % (k.replace("PFP_PARSE_", ""), v) for k, v in e.items())


def run_cases(kind, env, scale, seed=7):
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"
    if kind == "emu":
        e["PFP_EMU_POISON"] = "1"      # fresh device memory holds garbage, as on the card
    pr = subprocess.run([sys.executable, "-c", CODE, ROOT, kind, str(scale), str(seed)], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "recsort ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]


@pytest.mark.parametrize("env", ENVS, ids=IDS)
def test_recursive_parse_sort_emu(env):
    subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu"], check=True, stdout=subprocess.DEVNULL)
    run_cases("emu", env, 3)


PIPE_CODE = r'''
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
from pfp_testlib import *
import pfbwt_hip
lib = EMU_SO if sys.argv[2] == "emu" else None
F = lambda **kw: pfbwt_hip.PfpContext(lib=lib, **kw)
for name in sys.argv[3:]:
    man, recs = golden_case(name)
    seqs = [s for _, s in recs]
    for U in (8, 4):
        ref = oracle_run(seqs, w=man["w"], p=man["p"], U=U)
        res = engine_run(F, seqs, man["w"], man["p"], U)
        assert compare(res, ref, U) == [], (name, U)
        mf = man["files"]["u%d" % (U * 8)]
        for k, img in images(res, U).items():
            assert sha(img) == mf[k]["sha256"], (name, k)
print("pipeline ok")
'''


DICT_ENVS = [{"PFP_DICT_REC": "1"}, {"PFP_DICT_REC": "1", "PFP_DICT_REC_P2": "3", "PFP_PARSE_REC_TILE_ROWS": "30"}, {"PFP_DICT_REC": "1", "PFP_DICT_REC_P2": "64", "PFP_PARSE_REC": "1"},
             {"PFP_DICT_REC": "1", "PFP_PARSE_REC_TABLE_LOG2": "4"}]      # (the last: the phrase table overflows -> the old dictionary sorter)


@pytest.mark.parametrize("env", DICT_ENVS, ids=lambda e: ",".join("%s=%s" % (k.replace("PFP_", ""), v) for k, v in e.items()))
def test_pipeline_with_recursive_dictionary_sort_emu(env):
    """whole build with the dictionary suffix-sorted through its own level-2 parse (dictrec.h: D2 by the dictionary sorter, P2 by the integer sorter,
    tie classes, assembly): every array == oracle == the reference's file digests; other moduli, small assembly batches, a table that overflows"""
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"; e["PFP_EMU_POISON"] = "1"
    pr = subprocess.run([sys.executable, "-c", PIPE_CODE, ROOT, "emu", "edge", "w4p7", "mult_chroms_fa"], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "pipeline ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", DICT_ENVS, ids=lambda e: ",".join("%s=%s" % (k.replace("PFP_", ""), v) for k, v in e.items()))
def test_pipeline_with_recursive_dictionary_sort_gpu(env):
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"
    pr = subprocess.run([sys.executable, "-c", PIPE_CODE, ROOT, "gpu", "edge", "mult_chroms_fa", "w4p7", "single_chrom", "mult_chroms", "panel8"], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "pipeline ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]


@pytest.mark.parametrize("env", ENVS[:1] + ENVS[3:4], ids=IDS)
def test_pipeline_with_recursive_parse_sort_emu(env):
    """whole build (parse -> parse BWT through the recursive sort -> BWT + SA + samples) == oracle == the reference's file digests"""
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"; e["PFP_EMU_POISON"] = "1"
    pr = subprocess.run([sys.executable, "-c", PIPE_CODE, ROOT, "emu", "mult_chroms_fa", "w4p7"], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "pipeline ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", ENVS, ids=IDS)
def test_recursive_parse_sort_gpu(env):
    run_cases("gpu", env, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("env", ENVS[:1] + ENVS[3:4], ids=IDS)
def test_pipeline_with_recursive_parse_sort_gpu(env):
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"
    pr = subprocess.run([sys.executable, "-c", PIPE_CODE, ROOT, "gpu", "mult_chroms_fa", "w4p7", "single_chrom", "mult_chroms", "panel8"], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "pipeline ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]


@pytest.mark.gpu
def test_recursive_parse_sort_default_route_gpu(gpu_ctx_factory):
    """a string long and repetitive enough to take the route with the default switches (3 M symbols, 1500 distinct): SA == SA-IS,
    and the log says which route ran"""
    code = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1] + "/tests")
from pfp_testlib import oracle
import pfbwt_hip
rng = np.random.default_rng(11)
base = rng.integers(1, 1500, 30000).astype(np.uint32); rows = []
for h in range(100):
    r = base.copy(); m = rng.random(30000) < 0.004; r[m] = rng.integers(1, 1500, int(m.sum())); rows.append(r)
s = np.concatenate(rows + [np.zeros(1, np.uint32)])
SA, rounds = pfbwt_hip.sacak_int(s, 1500)
want = np.zeros(len(s), np.uint64)
assert oracle().orc_sais_int(s.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), len(s), 1500) == 0
assert np.array_equal(SA.astype(np.uint64), want)
print("default ok")
'''
    e = dict(os.environ); e["PFP_VERBOSE"] = "1"; e.pop("PFP_TEST_HOOKS", None)
    pr = subprocess.run([sys.executable, "-c", code, ROOT], env=e, capture_output=True, text=True, timeout=1500)
    assert pr.returncode == 0 and "default ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]
    assert "recursive parse sort (depth 0): N=3000001" in pr.stderr and "assembled:" in pr.stderr, pr.stderr[-2000:]
