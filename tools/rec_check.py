import os, sys, ctypes as C, time
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, "tests")); sys.path.insert(0, os.path.join(R_, "pfbwt-f_amd", "python"))
import pfbwt_hip
from pfp_testlib import oracle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "build", "libpfbwt_emu.so")
lib = EMU if len(sys.argv) < 2 or sys.argv[1] == "emu" else None
def check(s, k, tag):
    SA, rounds = pfbwt_hip.sacak_int(s, k, lib=lib)
    want = np.zeros(len(s), np.uint64)
    assert oracle().orc_sais_int(s.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), len(s), k) == 0
    ok = np.array_equal(SA.astype(np.uint64), want)
    print(tag, "n=%d k=%d rounds=%d %s" % (len(s), k, rounds, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        bad = np.flatnonzero(SA.astype(np.uint64) != want)
        print("  first bad rows", bad[:10], SA[bad[:10]], want[bad[:10]])
        sys.exit(1)
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
def panel(L, H, sigma, mut):
    base = rng.integers(1, sigma, L).astype(np.uint32)
    rows = []
    for h in range(H):
        r = base.copy(); m = rng.random(L) < mut; r[m] = rng.integers(1, sigma, m.sum()); rows.append(r)
    s = np.concatenate(rows + [np.zeros(1, np.uint32)]); return s
for (L, H, sigma, mut) in ((50, 4, 20, 0.05), (300, 20, 50, 0.02), (1000, 30, 300, 0.01), (40, 100, 9, 0.03), (2000, 8, 5, 0.01)):
    s = panel(L, H, sigma, mut); check(s, sigma, "panel L=%d H=%d" % (L, H))
for n, k in ((10, 3), (1000, 5), (5000, 300), (7000, 3), (3000, 2)):
    s = rng.integers(1, k, n).astype(np.uint32); s[-1] = 0; check(s, k, "random")
# periodic / runs
s = np.concatenate([np.tile(np.array([3, 1, 2], np.uint32), 500), [0]]).astype(np.uint32); check(s, 4, "periodic")
s = np.concatenate([np.full(900, 7, np.uint32), rng.integers(1, 9, 300).astype(np.uint32), np.full(500, 7, np.uint32), [0]]).astype(np.uint32); check(s, 9, "runs")
