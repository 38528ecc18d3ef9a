"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI of
include/pfbwt_hip.h, against (1) the committed fixtures generated from the reference and (2) the CPU
oracle on the same seeded inputs; plus size-independent properties at larger sizes.  Bar: bit-exact."""
import ctypes as C
import gzip
import os
import numpy as np
import pytest
from pfp_testlib import GOLDEN, ROOT, compare, engine_run, golden_case, golden_cases, images, oracle_run, sha

pytestmark = pytest.mark.gpu


def synth(seed, L, H, nruns=(0, 0, 0, 0)):
    lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
    lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
    out = []
    for h in range(H):
        a = np.empty(L, np.uint8)
        lib.pfp_synth_haplotype(seed, L, h, *nruns, a.ctypes.data_as(C.c_void_p))
        out.append(a.tobytes())
    return out


@pytest.mark.parametrize("name", golden_cases())
@pytest.mark.parametrize("U", [4, 8])
def test_engine_matches_reference_fixtures(gpu_ctx_factory, name, U):
    man, recs = golden_case(name)
    res = engine_run(gpu_ctx_factory, [s for _, s in recs], man["w"], man["p"], U)
    for k in ("n", "m", "dwords", "dsize", "r"):
        assert int(res[k]) == int(man[k]), k
    mf = man["files"]["u%d" % (U * 8)]
    for k, img in images(res, U).items():
        assert len(img) == mf[k]["size"], (name, k)
        assert sha(img) == mf[k]["sha256"], (name, k)


@pytest.mark.parametrize("name", ["single_chrom", "mult_chroms"])
def test_engine_matches_reference_own_goldens(gpu_ctx_factory, name):
    man, recs = golden_case(name)
    res = engine_run(gpu_ctx_factory, [s for _, s in recs], 10, 100, 8)
    d = os.path.join(GOLDEN, name)
    assert np.array_equal(res["bwt"], np.frombuffer(gzip.open(os.path.join(d, "reference_golden.bwt.gz")).read(), np.uint8))
    assert np.array_equal(res["sa"], np.frombuffer(gzip.open(os.path.join(d, "reference_golden.sa.u64.gz")).read(), "<u8"))


CASES = [
    # (seed, L, H, w, p, nruns, U)
    (1, 20000, 1, 10, 100, (0, 0, 0, 0), 4),
    (2, 30000, 4, 10, 100, (0, 0, 0, 0), 8),
    (3, 50000, 3, 4, 7, (0, 0, 0, 0), 4),          # many short phrases, big multi-word groups
    (4, 60000, 2, 10, 100, (5000, 9000, 40000, 300), 4),   # N runs -> giant phrases, long LCPs
    (5, 200000, 10, 10, 100, (0, 0, 0, 0), 4),     # panel
    (6, 20000, 2, 32, 16, (0, 0, 0, 0), 8),        # w = 32: hash.hpp:26 mask quirk -> k-mer is always 0; p | wang_hash(0): every position triggers
    (7, 40000, 2, 1, 3, (0, 0, 0, 0), 4),          # w = 1
    (8, 4099, 1, 10, 100, (0, 0, 0, 0), 4),        # ragged size
]


@pytest.mark.parametrize("seed,L,H,w,p,nruns,U", CASES)
def test_engine_matches_oracle_on_seeded_inputs(gpu_ctx_factory, seed, L, H, w, p, nruns, U):
    seqs = synth(seed, L, H, nruns)
    ref = oracle_run(seqs, w=w, p=p, U=U)
    res = engine_run(gpu_ctx_factory, seqs, w, p, U)
    assert compare(res, ref, U) == []


def test_engine_lowercase_ntoa_and_errors(gpu_ctx_factory):
    import pfbwt_hip
    s = synth(9, 30000, 1)[0]
    low = s.lower()
    a = engine_run(gpu_ctx_factory, [low], 10, 100, 4)
    b = oracle_run([s], w=10, p=100, U=4)
    assert compare(a, b, 4) == []
    iupac = bytearray(s); iupac[100] = ord("R"); iupac[20000] = ord("Y")
    ctx = gpu_ctx_factory(w=10, p=100, u64=False)
    ctx.feed(bytes(iupac))
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.finalize()
    assert e.value.status == pfbwt_hip.E_INVALID_CHAR and e.value.pos == 100 and e.value.ch == ord("R")
    ctx.close()
    a = engine_run(gpu_ctx_factory, [bytes(iupac)], 10, 100, 4, non_acgt_to_a=True)
    b = oracle_run([bytes(iupac)], w=10, p=100, U=4, non_acgt_to_a=True)
    assert compare(a, b, 4) == []
    ctx = gpu_ctx_factory(w=10, p=100)
    ctx.feed(b"ACGTACGTAAAA")
    ctx.finalize()
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.parse_bwt()
    assert e.value.status == pfbwt_hip.E_ONE_WORD
    ctx.close()


def test_engine_pfbwt_only_path(gpu_ctx_factory):
    seqs = synth(11, 80000, 5)
    ref = oracle_run(seqs, w=10, p=100, U=8)
    ctx = gpu_ctx_factory(w=10, p=100, u64=True)
    ctx.bwt_load(ref["dict"], ref["occ"], ref["bwlast"], ref["ilist"], ref["bwsai"], n_hint=ref["n"])
    b = ctx.bwt_build(sa=True, rssa=True)
    out = ctx.bwt_get()
    ctx.close()
    assert b.nout == ref["n"] + 1 and b.r == ref["r"]
    for k in ("bwt", "sa", "ssa", "esa"):
        assert np.array_equal(out[k].astype(np.uint64), ref[k]), k


def test_feed_device_view_gpu(gpu_ctx_factory):
    """the fused feed + trigger scan on the MI355X (rows read in place by pfp_parse_finalize), cases of tests/test_emu_pipeline.py"""
    from test_emu_pipeline import check_feed_device_view

    def to_dev(rows, length):      # device memory without another runtime in the process: the (raw, never finalized) text of a second context
        owner = gpu_ctx_factory(w=3, p=100)
        for h in range(rows.shape[0]):
            owner.feed(bytes(rows[h, :length]), True)
        ptr, n = C.c_void_p(), C.c_uint64()
        owner._check(owner.L.pfp_text_view(owner.h, C.byref(ptr), C.byref(n)))
        assert n.value == rows.shape[0] * (length + 3)
        return ptr.value, length + 3, owner
    check_feed_device_view(gpu_ctx_factory, to_dev=to_dev)


def test_sacak_int_dropin(gpu_ctx_factory):
    import pfbwt_hip
    from pfp_testlib import oracle
    rng = np.random.default_rng(3)
    for n, k in ((1000, 5), (50000, 300), (70000, 3)):
        s = rng.integers(1, k, n).astype(np.uint32); s[-1] = 0
        SA, rounds = pfbwt_hip.sacak_int(s, k)
        want = np.zeros(n, np.uint64)
        assert oracle().orc_sais_int(s.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), n, k) == 0
        assert np.array_equal(SA.astype(np.uint64), want) and rounds >= 1


def test_gsacak_dropin(gpu_ctx_factory):
    """pfp_gsacak_u32/u64 on the MI355X: SA, LCP and DA of dictionary images == the oracle's restatement of gsacak (which the
    CPU suite pins to the reference's own gsacak build), incl. a dictionary with a long run of N"""
    from test_emu_pipeline import _gsacak_check
    _gsacak_check(None, ["edge", "w4p7", "mult_chroms_fa", "panel8"])


def test_full_size_properties(gpu_ctx_factory):
    """S-50M-like panel at a size the oracle is too slow for in a unit test: check properties that do
    not need it -- SA is a permutation of 0..n, T[SA[i]-1] == BWT[i], run samples consistent."""
    seqs = synth(12345, 2_000_000, 5)
    w = 10
    res = engine_run(gpu_ctx_factory, seqs, w, 100, 4)
    n = res["n"]
    T = np.frombuffer(b"".join(s + b"A" * w for s in seqs), np.uint8)
    assert n == T.size
    sa = res["sa"].astype(np.int64); bwt = res["bwt"]
    assert sa[0] == n
    seen = np.zeros(n + 1, bool); seen[sa] = True
    assert seen.all()
    mk = sa > 0
    assert np.array_equal(T[sa[mk] - 1], bwt[mk]) and (bwt[~mk] == 0).all() and (~mk).sum() == 1
    # suffixes adjacent in SA are ordered: spot-check 2000 random adjacent pairs
    rng = np.random.default_rng(1)
    Tz = np.concatenate([T, np.zeros(1, np.uint8)])
    for i in rng.integers(1, n + 1, 2000):
        a, b = sa[i - 1], sa[i]
        la = bytes(Tz[a:a + 4000]); lb = bytes(Tz[b:b + 4000])
        assert la < lb or (len(la) == 4000 and la == lb)
    starts = np.flatnonzero(np.concatenate(([True], bwt[1:] != bwt[:-1])))
    assert res["r"] == starts.size
    assert np.array_equal(res["ssa"].reshape(-1, 2)[:, 0].astype(np.int64), starts)
    assert np.array_equal(res["ssa"].reshape(-1, 2)[:, 1].astype(np.int64), sa[starts])


def test_chr22_full_size_vs_oracle(gpu_ctx_factory):
    """configs[1] of BASELINE.json at its full size inside the driver's own test run: S-chr22 (50.8 Mbase, two runs of N, -s, 32-bit
    uint_t): sha256 of .bwt / .sa == oracle/pfbwt_oracle (about ten seconds of CPU); then -r on the same text: .ssa / .esa ==
    the oracle's, and the device-side adjacent-row order check over all run samples"""
    import hashlib, sys
    sys.path.insert(0, ROOT)
    import bench
    L, H, seed, nruns, w, p, u64 = bench.WORKLOADS["S-chr22"]
    seqs = bench.synth_seqs(L, H, seed, nruns)
    _, dig = bench.cpu_baseline(seqs, w, p, u64, mode=("-s", "-r"))
    c = gpu_ctx_factory(w=w, p=p, u64=u64, sai=True)
    for s in seqs:
        c.feed(s, True)
    c.finalize(); c.parse_bwt(); c.bwt_build(sa=True, rssa=True)
    o = c.bwt_get()
    for k in ("bwt", "sa", "ssa", "esa"):
        assert hashlib.sha256(o[k].tobytes()).hexdigest() == dig[k], k
    chk = c.check_sample_order()
    assert chk["pairs"] == len(o["ssa"]) // 2 - 1 and chk["order_violations"] == 0 and chk["rows_not_adjacent"] == 0, chk
    c.close()


def test_medium_panel_default_routes_vs_oracle(gpu_ctx_factory):
    """220 haplotypes x 1 Mbase with the DEFAULT switches against the oracle: the shape of the headline collection at a size the
    oracle finishes in seconds.  Groups of hundreds of rows with dozens of members reach the LDS-sorting group kernel
    (k_emit_groups_large) un-forced, the parse (2.2 M phrases) takes the recursive suffix sort (recsort.h) by itself; -r outputs
    (.bwt .ssa .esa) and, in a second build, the full .sa"""
    import hashlib, sys
    sys.path.insert(0, ROOT)
    import bench
    seqs = bench.synth_seqs(1_000_000, 220, 1000, (0, 0, 0, 0))
    _, dig = bench.cpu_baseline(seqs, 10, 100, True, mode=("-s", "-r"))
    c = gpu_ctx_factory(w=10, p=100, u64=True, sai=True)
    for sa, names in ((False, ("bwt", "ssa", "esa")), (True, ("bwt", "sa", "ssa", "esa"))):
        c.reset()
        # first build: every switch at its default (220 sequences: the wave-cooperative de-duplication kernel by itself); second build: the text visited as
        # 220 sequences x 39 loci, one set of locus columns per XCD -- the order S-32G takes by itself (its sequences are long enough), here next to the oracle
        if sa: c.debug_set(dedup_period=39)
        for s in seqs:
            c.feed(s, True)
        sz = c.finalize(); c.parse_bwt(); c.bwt_build(sa=sa, rssa=True)
        assert sz.m > (1 << 21)                                  # long enough for the recursive parse sort with the default switches
        o = c.bwt_get()
        for k in names:
            assert hashlib.sha256(o[k].tobytes()).hexdigest() == dig[k], (sa, k)
        chk = c.check_sample_order()
        assert chk["order_violations"] == 0 and chk["rows_not_adjacent"] == 0, chk
        if sa:      # the device-side property checks the S-3G test rests on, here next to the oracle's verdict on the same outputs
            cs = c.check_sa()
            assert cs["rows"] == sz.n + 1 and cs["out_of_range"] == 0 and cs["duplicates"] == 0 and cs["bwt_mismatches"] == 0 and cs["eos_bytes"] == 1, cs
            smp = c.check_samples()
            assert smp["row_errors"] == 0 and smp["value_errors"] == 0, smp
    c.close()


def test_s3g_full_size(gpu_ctx_factory):
    """configs[2] of BASELINE.json at its full size inside the driver's own test run: S-3G (3.1 Gbase, 35 Mbp of N in two runs, -s -r,
    64-bit uint_t).  The oracle cannot sort 3.1 G suffixes inside a unit test, so: (1) size-independent properties of the whole
    output on the device -- SA is a permutation of [0, n], BWT[row] == T[SA[row] - 1], one EOS byte (pfp_debug_check_sa); all
    adjacent run-boundary rows ordered, suffix against suffix on the text (pfp_debug_check_sample_order: r - 1 pairs, r = 0.74 n);
    every run sample names a row where the BWT byte changes and carries that row's SA value (pfp_debug_check_samples); (2) sha256 of
    .bwt .sa .ssa .esa == oracle/pfbwt_oracle on the first 100 Mbase of the same sequence."""
    import hashlib, sys
    sys.path.insert(0, ROOT)
    import bench
    L, H, seed, nruns, w, p, u64 = bench.WORKLOADS["S-3G"]
    seqs = bench.synth_seqs(L, H, seed, nruns)
    c = gpu_ctx_factory(w=w, p=p, u64=u64, sai=True)
    for s in seqs:
        c.feed(s, True)
    sz = c.finalize(); c.parse_bwt(); b = c.bwt_build(sa=True, rssa=True)
    assert sz.n == L + w and b.nout == sz.n + 1
    cs = c.check_sa()
    assert cs["rows"] == sz.n + 1 and cs["out_of_range"] == 0 and cs["duplicates"] == 0 and cs["bwt_mismatches"] == 0 and cs["eos_bytes"] == 1, cs
    chk = c.check_sample_order()
    assert chk["pairs"] == b.r - 1 and chk["order_violations"] == 0 and chk["rows_not_adjacent"] == 0, chk
    smp = c.check_samples()
    assert smp["runs"] == b.r and smp["row_errors"] == 0 and smp["value_errors"] == 0, smp
    c.close()
    # (2) the first 100 Mbase against the oracle
    sub = [seqs[0][:100_000_000]]
    _, dig = bench.cpu_baseline(sub, w, p, u64, mode=("-s", "-r"))
    c = gpu_ctx_factory(w=w, p=p, u64=u64, sai=True)
    c.feed(sub[0], True)
    c.finalize(); c.parse_bwt(); c.bwt_build(sa=True, rssa=True)
    o = c.bwt_get()
    for k in ("bwt", "sa", "ssa", "esa"):
        assert hashlib.sha256(o[k].tobytes()).hexdigest() == dig[k], k
    c.close()


def test_text_range_grows_and_moves(gpu_ctx_factory):
    """A text fed record by record without an announced size outgrows its first address range (64 MiB) and is MOVED to a larger
    one (ensure_text, csrc/pfbwt_hip.hip): eight records of 9 Mbase, a panel of one sequence with variants; -r outputs == the oracle's.
    (The move path is not reachable by the small inputs of the other tests.)"""
    import hashlib, sys
    sys.path.insert(0, ROOT)
    import bench
    seqs = bench.synth_seqs(9_000_000, 8, 4242, (0, 0, 0, 0))
    _, dig = bench.cpu_baseline(seqs, 10, 100, True, mode=("-r",))
    c = gpu_ctx_factory(w=10, p=100, u64=True, sai=True)
    c.feed(b"ACGT" * 5, True)          # a first feed of a few bytes: the range starts at its minimum size
    c.reset()
    for s in seqs:
        c.feed(s, True)
    c.finalize(); c.parse_bwt(); c.bwt_build(sa=False, rssa=True)
    o = c.bwt_get()
    for k in ("bwt", "ssa", "esa"):
        assert hashlib.sha256(o[k].tobytes()).hexdigest() == dig[k], k
    chk = c.check_sample_order()
    assert chk["order_violations"] == 0 and chk["rows_not_adjacent"] == 0, chk
    c.close()


def test_sample_order_check_detects_disorder(gpu_ctx_factory):
    """the device-side order check is not vacuous: the same samples checked against a DIFFERENT text (same length) report violations"""
    seqs = synth(77, 300_000, 3)
    c = gpu_ctx_factory(w=10, p=100, u64=True, sai=True)
    for s in seqs:
        c.feed(s, True)
    c.finalize(); c.parse_bwt(); c.bwt_build(sa=False, rssa=True)
    good = c.check_sample_order()
    assert good["order_violations"] == 0 and good["rows_not_adjacent"] == 0 and good["pairs"] > 1000
    # overwrite the resident text with another one of the same length (device copy through the engine's own entry point)
    import ctypes as C
    other = np.frombuffer(b"".join(s + b"A" * 10 for s in synth(78, 300_000, 3)), np.uint8).copy()
    c2 = gpu_ctx_factory(w=10, p=100, u64=True, sai=True)      # the other text reaches the device through a second context
    for s in synth(78, 300_000, 3):
        c2.feed(s, True)
    dt, dn, st, sn = C.c_void_p(0), C.c_uint64(0), C.c_void_p(0), C.c_uint64(0)
    c._check(c.L.pfp_text_view(c.h, C.byref(dt), C.byref(dn)))
    c2._check(c2.L.pfp_text_view(c2.h, C.byref(st), C.byref(sn)))
    assert dn.value == other.size == sn.value
    c.device_copy(dt.value, st.value, other.size)
    bad = c.check_sample_order()
    assert bad["order_violations"] > good["pairs"] // 10, bad
    c.close(); c2.close()


def test_engine_ragged_inputs(gpu_ctx_factory):
    """empty records, records shorter than w, hundreds of tiny records, one-word parses"""
    from pfp_testlib import check_ragged
    check_ragged(gpu_ctx_factory)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606])
def test_random_differential_gpu(gpu_ctx_factory, seed):
    """the seeded random sweep of tests/pfp_testlib.random_cases on the card (seeds 101..303 also run on the emulator)"""
    from pfp_testlib import check_random
    check_random(gpu_ctx_factory, seed, 25)


CLASS_SORT_CODE = r'''
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
from pfp_testlib import *
from test_gpu_parity import synth
import pfbwt_hip
seqs = synth(77, 50000, 40)
ref = oracle_run(seqs, w=10, p=20, U=8)
res = engine_run(lambda **kw: pfbwt_hip.PfpContext(**kw), seqs, 10, 20, 8, sa=True, rssa=True)
bad = compare(res, ref, 8)
assert bad == [], bad
print("class sort variant ok")
'''


@pytest.mark.parametrize("env", [{"PFP_CLASS_SORT_MAXRANGE": "1160"}, {}, {"PFP_CLASS_SORT_MAXRANGE": "40"},
                                 {"PFP_SORT_K": "1"}, {"PFP_SORT_K": "1", "PFP_CLASS_SORT_MAXRANGE": "40"}, {"PFP_SORT_NO_TABLE": "1"}])
def test_doubling_round_sort_routes(env):
    """the sort of a refinement round: classes sorted inside LDS tiles by the fused round kernel (default: by three further
    ranks per round, read from the table built in text order; PFP_SORT_NO_TABLE: by following the chains; PFP_SORT_K=1:
    plain doubling), classes too large for a tile collected and radix-sorted (forced by a smaller range limit: some of
    the pairs, nearly all of them) -- every route must give the oracle's arrays"""
    import subprocess, sys
    e = dict(os.environ); e.update(env); e["PFP_VERBOSE"] = "1"; e["PFP_TEST_HOOKS"] = "1"
    pr = subprocess.run([sys.executable, "-c", CLASS_SORT_CODE, ROOT], env=e, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0 and "class sort variant ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]
    lines = [l for l in pr.stderr.splitlines() if "class sort:" in l]
    if env.get("PFP_CLASS_SORT_MAXRANGE"):       # the large-class route really ran
        assert lines, "no round reported classes too large for a tile"
    else:
        assert not lines, lines[:5]
    k3 = [l for l in pr.stderr.splitlines() if "K=3" in l]
    assert bool(k3) == (env.get("PFP_SORT_K") != "1"), pr.stderr[-2000:]
    assert any("(table)" in l for l in k3) == (not env.get("PFP_SORT_K") and not env.get("PFP_SORT_NO_TABLE")), k3[:5]
