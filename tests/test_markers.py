"""Marker-array post-pass (include/marker_array.hpp:138-174, src/mps_to_ma.cpp).
* oracle/marker_oracle.py is pinned to the reference's own goldens tests/data/{single_chrom,mult_chroms}.{sa,markers}
  (kept as reference_golden.*.gz; fixtures + derivation: tests/golden/make_marker_golden.py);
* the engine (pfp_marker_array, fused with the build or on a given suffix array) must equal the oracle on those fixtures and
  on seeded marker-position streams with multi-marker lists, repeated lists, empty lists and untouched text;
* the mps_to_ma command line reproduces the reference tool's files."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import numpy as np
import pytest
from pfp_testlib import EMU_SO, GOLDEN, ROOT, engine_run, golden_case, oracle_run

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import marker_oracle as mo


def fixture(case):
    d = os.path.join(GOLDEN, case)
    sa = np.frombuffer(gzip.open(os.path.join(d, "reference_golden.sa.u64.gz")).read(), "<u8")
    txt = gzip.open(os.path.join(d, "reference_golden.markers.gz")).read().decode()
    mps = np.fromfile(os.path.join(d, "markers.mps"), "<u8")
    return sa, txt, mps, json.load(open(os.path.join(d, "markers.json")))


@pytest.mark.parametrize("case", ["single_chrom", "mult_chroms"])
def test_marker_oracle_reproduces_reference_goldens(case):
    sa, txt, mps, meta = fixture(case)
    assert np.array_equal(mo.mps_from_golden(sa, txt), mps)
    ma = mo.marker_array(mps, sa)
    assert mo.readable(ma) == txt                          # the reference's own golden file, byte for byte
    assert hashlib.sha256(ma.astype("<u8").tobytes()).hexdigest() == meta["ma_sha256"]
    # marker packing, include/marker.hpp (tests/marker_test.cpp checks the same fields)
    m = mo.create_marker(123456789, 3, 77)
    assert mo.marker_fields(m) == (77, 123456789, 3)


def seeded_mps(n, seed, multi=True):
    """intervals over [0, n): one to three markers per list, some lists repeated in later intervals, one empty list"""
    rng = np.random.default_rng(seed)
    starts, ends, lists, p = [], [], [], int(rng.integers(0, 50))
    pool = [tuple(mo.create_marker(int(rng.integers(1, 1 << 30)), int(rng.integers(0, 4)), int(rng.integers(0, 300))) for _ in range(int(rng.integers(1, 4 if multi else 2)))) for _ in range(40)]
    while p < n - 30:
        ln = int(rng.integers(1, 25))
        starts.append(p); ends.append(p + ln - 1); lists.append(pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.9 else ())
        p += ln + (0 if rng.random() < 0.3 else int(rng.integers(1, 400)))      # adjacent intervals too
    return mo.mps_build(starts, ends, lists)


def engine_check(factory):
    for case in ("single_chrom", "mult_chroms"):
        sa, txt, mps, meta = fixture(case)
        man, recs = golden_case(case)
        # fused: the suffix array never leaves the device
        ctx = factory(w=man["w"], p=man["p"], u64=True, sai=True)
        for _, s in recs:
            ctx.feed(s, True)
        ctx.finalize(); ctx.parse_bwt(); ctx.bwt_build(sa=True, rssa=False)
        ma = ctx.marker_array(mps)
        assert hashlib.sha256(ma.astype("<u8").tobytes()).hexdigest() == meta["ma_sha256"] and mo.readable(ma) == txt
        assert np.array_equal(ctx.marker_array(seeded_mps(sa.size, 5)), mo.marker_array(seeded_mps(sa.size, 5), sa))
        ctx.close()
        # stand-alone, from a suffix array in host memory (what src/mps_to_ma.cpp does), both uint_t widths
        for u64 in (True, False):
            ctx = factory(w=10, p=100, u64=u64, sai=True)
            for seed in (1, 2):
                m2 = seeded_mps(sa.size, seed, multi=seed == 1)
                assert np.array_equal(ctx.marker_array(m2, sa=sa), mo.marker_array(m2, sa))
            assert ctx.marker_array(np.zeros(0, np.uint64), sa=sa).size == 0          # no records: an empty marker array
            ctx.close()
    import pfbwt_hip
    ctx = factory(w=10, p=100, u64=True)
    bad = mo.mps_build([10, 5], [20, 8], [(1,), (2,)])                                 # intervals out of order
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.marker_array(bad, sa=np.arange(100, dtype=np.uint64))
    assert e.value.status == pfbwt_hip.E_CORRUPT
    ctx.close()


def cli_check(exe, tmp):
    """mps_to_ma <mps> <sa> -o out (src/mps_to_ma.cpp:19-51): out == the oracle's stream; '-' reads the suffix array from stdin"""
    for case in ("single_chrom", "mult_chroms"):
        sa, txt, mps, meta = fixture(case)
        sap = os.path.join(tmp, case + ".sa"); sa.astype("<u8").tofile(sap)
        out = os.path.join(tmp, case + ".ma")
        pr = subprocess.run([exe, "-o", out, os.path.join(GOLDEN, case, "markers.mps"), sap], capture_output=True, text=True)
        assert pr.returncode == 0, pr.stderr[-2000:]
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == meta["ma_sha256"]
        with open(sap, "rb") as f:
            pr = subprocess.run([exe, "-o", out + "2", os.path.join(GOLDEN, case, "markers.mps"), "-"], stdin=f, capture_output=True, text=True)
        assert pr.returncode == 0 and open(out + "2", "rb").read() == open(out, "rb").read()


def test_marker_array_emu(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu-host"], check=True, stdout=subprocess.DEVNULL)
    import pfbwt_hip
    engine_check(lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw))
    cli_check(os.path.join(ROOT, "tests", "emu", "build", "mps_to_ma-emu"), str(tmp_path))


@pytest.mark.gpu
def test_marker_array_gpu(gpu_ctx_factory, tmp_path):
    engine_check(gpu_ctx_factory)
    cli_check(os.path.join(ROOT, "pfbwt-f_amd", "bin", "mps_to_ma"), str(tmp_path))
