"""CPU tests of the host orchestration and kernel logic through tests/emu (a fiber interpreter of the
HIP execution model -- a debugging harness for a container without a GPU, NOT a backend: the product
never loads it and no parity claim rests on it; the parity tests proper are tests/test_gpu_parity.py)."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
from pfp_testlib import EMU_SO, ROOT, compare, engine_run, golden_case, images, oracle_run, sha


@pytest.fixture(scope="module")
def emu_factory():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu"], check=True, stdout=subprocess.DEVNULL)
    import pfbwt_hip
    assert pfbwt_hip.load_library(EMU_SO).pfp_backend().decode() == "cpu-emu-TEST-ONLY"
    return lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw)


@pytest.mark.parametrize("name,U", [("edge", 8), ("edge", 4), ("w4p7", 8), ("mult_chroms_fa", 4)])
def test_emu_pipeline_matches_oracle_and_reference_digests(emu_factory, name, U):
    man, recs = golden_case(name)
    seqs = [s for _, s in recs]
    ref = oracle_run(seqs, w=man["w"], p=man["p"], U=U)
    res = engine_run(emu_factory, seqs, man["w"], man["p"], U)
    assert compare(res, ref, U) == []
    mf = man["files"]["u%d" % (U * 8)]
    for k, img in images(res, U).items():
        assert sha(img) == mf[k]["sha256"], k


def test_emu_long_phrases(emu_factory):
    """runs of N longer than the long-phrase threshold (2048): chunked fingerprints, chunked pair verification, the run
    round of the suffix sorter; the same long phrase twice, and one of a different length"""
    rng = np.random.default_rng(3)
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8))
    a = rnd(3000) + b"N" * 40001 + rnd(2500)
    seqs = [a, a, rnd(1200) + b"N" * 20000 + rnd(900)]
    ref = oracle_run(seqs, w=10, p=100, U=4)
    res = engine_run(emu_factory, seqs, 10, 100, 4)
    assert compare(res, ref, 4) == []


def test_emu_feed_device_batch(emu_factory):
    """pfp_parse_feed_device_batch == one pfp_parse_feed_device per record (the emulator's "device" memory is host memory)"""
    rng = np.random.default_rng(8)
    base = rng.choice(list(b"ACGT"), 3000).astype(np.uint8)
    haps = np.stack([base.copy() for _ in range(5)])
    for h in range(1, 5):
        haps[h, rng.integers(0, 3000, 25)] = rng.choice(list(b"ACGT"), 25)
    pad = np.zeros((5, 3100), np.uint8); pad[:, :3000] = haps            # stride > len
    a = emu_factory(w=6, p=13, sai=True); b = emu_factory(w=6, p=13, sai=True)
    a.feed_device_batch(pad.ctypes.data, 5, 3000, 3100)
    for h in range(5):
        b.feed_device(haps[h].ctypes.data, 3000, True)
    sa_, sb = a.finalize(), b.finalize()
    assert (sa_.n, sa_.m, sa_.dwords, sa_.dsize) == (sb.n, sb.m, sb.dwords, sb.dsize) == (5 * 3006, sb.m, sb.dwords, sb.dsize)
    pa, pb = a.parse_get(), b.parse_get()
    assert all(np.array_equal(pa[k], pb[k]) for k in pa)
    a.close(); b.close()


def check_feed_device_view(factory, to_dev=None):
    """pfp_parse_feed_device_view (the rows stay where they are, the trigger scan of finalize reads them in place and writes the text)
    == pfp_parse_feed_device_batch, for rows shorter than a load, rows that are no multiple of 16, lower case / N / IUPAC, the
    hash-per-window scan (w > 10: the rows are materialised first), a view followed by another feed or a text view (materialised),
    and the error position of an invalid character"""
    import pfbwt_hip
    rng = np.random.default_rng(12)
    keep = []
    def dev(a, length=None):      # -> (device pointer, stride)
        if to_dev is None: return a.ctypes.data, a.shape[1]
        ptr, stride, owner = to_dev(a, a.shape[1] if length is None else length); keep.append(owner); return ptr, stride
    for (count, length, stride, w, p, alphabet, ntoa) in ((5, 3000, 3100, 6, 13, b"ACGT", False), (40, 7, 7, 4, 3, b"ACGT", False), (3, 16384 + 5, 16400, 10, 100, b"ACGTacgtNn", False),
                                                          (9, 1000, 1024, 12, 50, b"ACGT", False), (4, 2500, 2500, 8, 20, b"ACGTRYKM", True), (1, 50000, 50000, 10, 100, b"ACGT", False)):
        rows = np.zeros((count, stride), np.uint8)
        base = rng.choice(list(alphabet), length).astype(np.uint8)
        for h in range(count):
            rows[h, :length] = base
            rows[h, rng.integers(0, length, max(1, length // 100))] = rng.choice(list(alphabet), max(1, length // 100))
        dp, stride = dev(rows, length)
        res = []
        for mode in ("view", "batch", "view+feed", "view+textview"):
            c = factory(w=w, p=p, sai=True, non_acgt_to_a=ntoa)
            if mode == "batch": c.feed_device_batch(dp, count, length, stride)
            else: c.feed_device_view(dp, count, length, stride)
            if mode == "view+feed": c.feed(bytes(base[:50]), True)
            if mode == "view+textview":
                tv = (C.c_void_p(), C.c_uint64()); c._check(c.L.pfp_text_view(c.h, C.byref(tv[0]), C.byref(tv[1]))); assert tv[1].value == count * (length + w)
            sz = c.finalize(); c.parse_bwt(); b = c.bwt_build(sa=True, rssa=True)
            out = c.bwt_get(); out.update(c.parse_get()); out["sizes"] = (sz.n, sz.m, sz.dwords, sz.dsize, b.r)
            res.append(out); c.close()
        for k in res[0]:
            if k == "sizes": assert res[0][k] == res[1][k] == res[3][k], (count, length, k)
            else: assert np.array_equal(res[0][k], res[1][k]) and np.array_equal(res[0][k], res[3][k]), (count, length, k)
        assert res[2]["sizes"][0] == res[0]["sizes"][0] + min(50, length) + w
    # an invalid character inside a viewed row: the reference's error, position and byte (hash.hpp:31)
    rows = np.frombuffer(b"ACGT" * 500, np.uint8).copy().reshape(2, 1000); rows[1, 321] = ord("R")
    dp, st = dev(rows)
    c = factory(w=10, p=100); c.feed_device_view(dp, 2, 1000, st)
    with pytest.raises(pfbwt_hip.PfpError) as e:
        c.finalize()
    assert e.value.status == -2 and e.value.pos == 1010 + 321 and e.value.ch == ord("R")
    c.close()
    c = factory(w=10, p=100); c.feed(b"ACGT" * 10, True)      # a view must be the whole text
    with pytest.raises(pfbwt_hip.PfpError) as e:
        c.feed_device_view(dp, 2, 1000, st)
    assert e.value.status == -7
    c.close()


def test_emu_feed_device_view(emu_factory):
    check_feed_device_view(emu_factory)


def test_emu_pfbwt_only_path(emu_factory):
    """--pfbwt-only: stage 2 from the on-disk arrays alone (pfp_bwt_load)."""
    man, recs = golden_case("w4p7")
    ref = oracle_run([s for _, s in recs], w=man["w"], p=man["p"], U=8)
    ctx = emu_factory(w=man["w"], p=man["p"], u64=True)
    ctx.bwt_load(ref["dict"], ref["occ"], ref["bwlast"], ref["ilist"], ref["bwsai"], n_hint=ref["n"])
    b = ctx.bwt_build(sa=True, rssa=True)
    out = ctx.bwt_get()
    ctx.close()
    assert b.nout == ref["n"] + 1 and b.r == ref["r"]
    for k in ("bwt", "sa", "ssa", "esa"):
        assert np.array_equal(out[k].astype(np.uint64), ref[k]), k


def test_emu_error_paths(emu_factory):
    import pfbwt_hip
    ctx = emu_factory(w=4, p=5)
    ctx.feed(b"ACGTRACGTACGTACGTACGT")
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.finalize()
    assert e.value.status == pfbwt_hip.E_INVALID_CHAR and e.value.pos == 4 and e.value.ch == ord("R")
    ctx.close()
    with pytest.raises(pfbwt_hip.PfpError):
        emu_factory(w=33, p=100)
    ctx = emu_factory(w=10, p=100)
    ctx.feed(b"ACGTACGTAAAA")                     # no trigger: one phrase
    ctx.finalize()
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.parse_bwt()
    assert e.value.status == pfbwt_hip.E_ONE_WORD
    ctx.close()


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_emu_random_differential(emu_factory, seed):
    """seeded random collections (panels with indels, unrelated records, N runs / homopolymers / tandem repeats, IUPAC and
    lower case), random w, p, uint_t width and output combination: every array equals the oracle's"""
    from pfp_testlib import check_random
    check_random(emu_factory, seed, 25)


def test_emu_workspace_sweep(emu_factory):
    """explicit workspace sizes from far too small to ample: every build either equals the oracle or reports PFP_E_NOMEM at a
    stage boundary (the optional allocations -- the table of the three-rank rounds -- must give way, not fail the build)"""
    import pfbwt_hip
    rng = np.random.default_rng(5)
    base = rng.choice(list(b"ACGT"), 60000).astype(np.uint8)
    seqs = []
    for h in range(6):
        b = base.copy()
        for _ in range(200): b[int(rng.integers(0, b.size))] = rng.choice(list(b"ACGT"))
        seqs.append(bytes(b))
    ref = oracle_run(seqs, w=10, p=20, U=8)
    outcomes = []
    for ws in (1 << 20, 10 << 20, 14 << 20, 18 << 20, 24 << 20, 64 << 20):
        c = emu_factory(w=10, p=20, u64=True, sai=True, workspace_bytes=ws)
        try:
            for s in seqs: c.feed(s, True)
            c.finalize(); c.parse_bwt(); b = c.bwt_build(sa=True, rssa=True)
            res = c.bwt_get(); res["r"] = b.r
            assert compare(res, ref, 8, names=("bwt", "sa", "ssa", "esa")) == [] and b.r == ref["r"], ws
            outcomes.append("ok")
        except pfbwt_hip.PfpError as e:
            assert e.status == pfbwt_hip.E_NOMEM, (ws, e)
            outcomes.append("nomem")
        c.close()
    assert outcomes[0] == "nomem" and outcomes[-1] == "ok" and outcomes == sorted(outcomes, key=lambda o: o == "ok"), outcomes


def test_emu_ragged_inputs(emu_factory):
    from pfp_testlib import check_ragged
    check_ragged(emu_factory)


VARIANT_CODE = r'''
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
from pfp_testlib import *
import pfbwt_hip
F = lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw)
import os, random
man, recs = golden_case("w4p7")
cases = [([s for _, s in recs], man["w"], man["p"])]
if "PFP_BIG_GROUP_MEMBERS" in os.environ or "PFP_CLASS_SORT_MIN" in os.environ or "PFP_EMIT_GROUP_ROWS" in os.environ:       # a small "panel": many words share long suffixes -> groups with many members
    rng = random.Random(5); base = [rng.choice("ACGT") for _ in range(1500)]; haps = []
    for h in range(7):
        b = list(base)
        for _ in range(40): b[rng.randrange(len(b))] = rng.choice("ACGT")
        haps.append("".join(b).encode())
    cases.append((haps, 4, 11))
for ci, (seqs, w, p) in enumerate(cases):
    for U in ((8, 4) if ci == 0 else (8,)):
        ref = oracle_run(seqs, w=w, p=p, U=U)
        for sa, rssa in (((True, True), (False, True), (True, False), (False, False)) if (U == 8 and ci == 0) else ((False, True),)):
            res = engine_run(F, seqs, w, p, U, sa=sa, rssa=rssa)
            names = ["bwt"] + (["sa"] if sa else []) + (["ssa", "esa"] if rssa else [])
            bad = compare(res, ref, U, names=tuple(names))
            assert bad == [] and res["r"] == ref["r"], (U, sa, rssa, bad)
        if U == 8:      # streamed build: the windows' rows reach the host buffers shifted by one row (pfp_bwt_build_stream)
            import numpy as np
            for sa, rssa in ((False, True), (True, False)):
                c = F(w=w, p=p, u64=True, sai=True)
                for s_ in seqs: c.feed(s_, True)
                c.finalize(); c.parse_bwt()
                hb = np.full(ref["n"] + 1, 0xEE, np.uint8); hs = np.full(ref["n"] + 1, 0xEEEEEEEE, np.uint64)
                b = c.bwt_build_stream(hb.ctypes.data, hs.ctypes.data if sa else None, rssa=rssa)
                assert np.array_equal(hb, ref["bwt"]) and b.r == ref["r"] and (not sa or np.array_equal(hs, ref["sa"])), (ci, sa, rssa, "streamed")
                c.close()
print("variant ok")
'''


@pytest.mark.parametrize("env", [{"PFP_FORCE_WIDE_ROWS": "1"}, {"PFP_FORCE_WIDE_ROWS": "1", "PFP_EMIT_CHUNK_ROWS": "1000"},
                                 {"PFP_EMIT_CHUNK_ROWS": "777", "PFP_SAMPLE_CAP": "40"},
                                 {"PFP_EMIT_GROUP_ROWS": "0", "PFP_NO_SLOT_RECORDS": "1"}, {"PFP_DICT_TEXT_ROUNDS": "0", "PFP_INT_KEY_SYMBOLS": "2", "PFP_FORCE_RUN_ROUND": "1"}, {"PFP_DICT_TEXT_ROUNDS": "1", "PFP_CLASS_SORT_MIN": "1", "PFP_CLASS_SORT_MAXRANGE": "150"}, {"PFP_EMIT_GROUP_ROWS": "4096", "PFP_EMIT_CHUNK_ROWS": "1500"}, {"PFP_EMIT_GROUP_ROWS": "12", "PFP_EMIT_CHUNK_ROWS": "900", "PFP_FORCE_WIDE_ROWS": "1"},
                                 {"PFP_DEDUP_TABLE_LOG2": "4", "PFP_NO_TRIGGER_TABLE": "1", "PFP_NO_RUNAWARE": "1"},
                                 {"PFP_BIG_GROUP_MEMBERS": "1", "PFP_CLASS_SORT_MIN": "1"},
                                 {"PFP_BIG_GROUP_MEMBERS": "2", "PFP_EMIT_CHUNK_ROWS": "5000", "PFP_CLASS_SORT_MIN": "1", "PFP_CLASS_SORT_MAXRANGE": "150"},
                                 {"PFP_CLASS_SORT_MIN": "1", "PFP_SORT_K": "1", "PFP_CLASS_SORT_MAXRANGE": "150"}, {"PFP_CLASS_SORT_MIN": "1", "PFP_SORT_NO_TABLE": "1"},
                                 {"PFP_DICT_REC": "1", "PFP_BIG_GROUP_MEMBERS": "2"}, {"PFP_DICT_REC": "1", "PFP_DICT_REC_P2": "5", "PFP_PARSE_REC_TILE_ROWS": "20", "PFP_PARSE_REC": "1", "PFP_EMIT_GROUP_ROWS": "12"},
                                 {"PFP_DEDUP_VARIANT": "0", "PFP_DEDUP_PERIOD": "-1"}, {"PFP_DEDUP_VARIANT": "1", "PFP_DEDUP_PERIOD": "2", "PFP_DEDUP_CHUNK": "1"}, {"PFP_DEDUP_PERIOD": "3", "PFP_DEDUP_TABLE_LOG2": "5", "PFP_DEDUP_VARIANT": "0"}])
def test_emu_wide_rows_and_chunked_emission(emu_factory, env):
    """The code paths taken by texts of 2^32 bases and more (64-bit row counters, emission in windows of rows, run
    samples in two passes), the sort route for groups of equal suffixes with many members and the LDS class sort of the
    doubling rounds, forced on small inputs: every output combination must still equal the oracle."""
    import sys
    from pfp_testlib import ROOT
    e = dict(os.environ); e.update(env); e["PFP_TEST_HOOKS"] = "1"
    pr = subprocess.run([sys.executable, "-c", VARIANT_CODE, ROOT], env=e, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0 and "variant ok" in pr.stdout, pr.stdout[-1500:] + pr.stderr[-3000:]


def test_emu_load_rejects_inconsistent_files(emu_factory):
    """--pfbwt-only with a truncated .ilist, a wrong .occ, an unterminated .dict or an ilist entry past the last row:
    PFP_E_CORRUPT at load time (these used to become out-of-bounds device reads in the emission)"""
    import pfbwt_hip
    man, recs = golden_case("w4p7")
    ref = oracle_run([s for _, s in recs], w=man["w"], p=man["p"], U=8)
    good = dict(dict_=ref["dict"], occ=ref["occ"], bwlast=ref["bwlast"], ilist=ref["ilist"], bwsai=ref["bwsai"], n_hint=ref["n"])

    def expect_corrupt(**chg):
        a = dict(good); a.update(chg)
        ctx = emu_factory(w=man["w"], p=man["p"], u64=True)
        with pytest.raises(pfbwt_hip.PfpError) as e:
            ctx.bwt_load(**a)
        assert e.value.status == pfbwt_hip.E_CORRUPT, e.value
        ctx.close()

    expect_corrupt(ilist=ref["ilist"][:-3])                               # truncated file
    expect_corrupt(bwsai=ref["bwsai"][:10])
    occ = ref["occ"].copy(); occ[3] += 5
    expect_corrupt(occ=occ)                                               # sum(occ) + 1 != rows
    il = ref["ilist"].copy(); il[7] = len(il) + 1000
    expect_corrupt(ilist=il)                                              # entry past the last row
    d = ref["dict"].copy(); d[-1] = 65
    expect_corrupt(dict_=d)                                               # no EndOfDict
    expect_corrupt(n_hint=5)                                              # more rows than text positions
    # a wrong .n: caught by the "exactly n + 1 rows" check of the emission
    ctx = emu_factory(w=man["w"], p=man["p"], u64=True)
    a = dict(good); a["n_hint"] = ref["n"] + 17
    ctx.bwt_load(**a)
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.bwt_build(sa=True, rssa=False)
    assert e.value.status == pfbwt_hip.E_CORRUPT
    ctx.close()


def test_emu_failed_stage_leaves_context_usable(emu_factory):
    """a stage that fails restores the workspace marks.  (1) PFP_E_NOMEM from bwt_build(sa=True) in a small
    workspace, then the cheaper request in the SAME context succeeds and is bit-exact; (2) a rejected text
    (invalid character) followed by pfp_reset and a good text."""
    import pfbwt_hip
    man, recs = golden_case("w4p7")
    seqs = [s for _, s in recs]
    ref = oracle_run(seqs, w=man["w"], p=man["p"], U=8)

    def attempt(ws, sa_first):
        """outcome of [bwt_build(sa=True)] + bwt_build(sa=False) in one context with `ws` bytes of workspace; stage 2 alone
        (--pfbwt-only: the parse stage has its own, larger, minimum and would hide the window between the two requests)"""
        ctx = emu_factory(w=man["w"], p=man["p"], u64=True, workspace_bytes=ws)
        try:
            try:
                ctx.bwt_load(ref["dict"], ref["occ"], ref["bwlast"], ref["ilist"], ref["bwsai"], n_hint=ref["n"])
                if sa_first:
                    try:
                        ctx.bwt_build(sa=True, rssa=True)
                        return "sa fits"
                    except pfbwt_hip.PfpError as e:
                        assert e.status == pfbwt_hip.E_NOMEM
                b = ctx.bwt_build(sa=False, rssa=False)
            except pfbwt_hip.PfpError as e:
                assert e.status == pfbwt_hip.E_NOMEM
                return "nomem"
            assert np.array_equal(ctx.bwt_get()["bwt"], ref["bwt"]) and b.r == ref["r"]
            return "bwt ok"
        finally:
            ctx.close()

    lo, hi = 50_000, 16_000_000                    # smallest workspace in which the BWT-only build succeeds (fresh context)
    assert attempt(hi, False) == "bwt ok"
    while hi - lo > 4096:
        mid = (lo + hi) // 2
        if attempt(mid, False) == "bwt ok": hi = mid
        else: lo = mid
    # in that workspace the full-SA request does not fit; the cheaper request after the failure must behave like in a fresh context
    # ("sa fits": since round 3 a build with a full SA drops the per-slot arrays before it samples, and can need less than the run-aware
    # BWT-only build; the recovery path is then exercised by the next, smaller, workspaces)
    assert attempt(hi, True) in ("bwt ok", "sa fits"), "PFP_E_NOMEM from bwt_build(sa=True) left the workspace unusable for the cheaper request"
    for ws in (hi - 8192, hi - 65536, hi // 2):      # the full-SA request fails or not, the cheaper one fails: never a crash, never a wrong answer
        assert attempt(ws, True) in ("bwt ok", "sa fits", "nomem")
    ctx = emu_factory(w=man["w"], p=man["p"], u64=True)
    ctx.feed(b"ACGTRRACGT" * 30, True)
    with pytest.raises(pfbwt_hip.PfpError) as e:
        ctx.finalize()
    assert e.value.status == pfbwt_hip.E_INVALID_CHAR
    ctx.reset()
    for s in seqs:
        ctx.feed(s, True)
    sz = ctx.finalize(); ctx.parse_bwt(); ctx.bwt_build(sa=True, rssa=True)
    assert sz.n == ref["n"] and np.array_equal(ctx.bwt_get()["sa"], ref["sa"])
    ctx.close()


def _gsacak_check(lib, names, nruns_case=True):
    """pfp_gsacak_u32/u64 (gsa/gsacak.h:86-96 drop-in): SA, LCP (stops at the separator) and DA of a dictionary image ==
    the reference's gsacak() (oracle/_ref/libgsacak64.so when it is there) and == the oracle's restatement of it"""
    import ctypes as C
    import pfbwt_hip
    from pfp_testlib import ROOT, oracle
    so = os.path.join(ROOT, "oracle", "_ref", "libgsacak64.so")
    G = C.CDLL(so) if os.path.exists(so) else None
    if G is not None:
        G.gsacak.argtypes = [C.c_void_p] * 4 + [C.c_uint64]
    O = oracle(); O.orc_gsa_lcp.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    dicts = []
    for name in names:
        man, recs = golden_case(name)
        dicts.append(oracle_run([s for _, s in recs], w=man["w"], p=man["p"], U=8)["dict"].copy())
    if nruns_case:     # a long run of N inside a phrase (the run-skipping branch of the LCP kernel) and a phrase that recurs
        rng = np.random.default_rng(4)
        rnd = lambda k: bytes(rng.choice(list(b"ACGT"), k).astype(np.uint8))
        a = rnd(700) + b"N" * 5000 + rnd(300)
        dicts.append(oracle_run([a, rnd(100) + a[200:], a], w=6, p=13, U=8)["dict"].copy())
    for d in dicts:
        n = d.size; dwords = int((d == 1).sum())
        a = np.zeros(n, np.uint64); b = np.zeros(n, np.uint64)
        assert O.orc_gsa_lcp(d.ctypes.data_as(C.c_void_p), n, dwords, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
        for u64 in (True, False):
            sa, lcp, da, r = pfbwt_hip.gsacak(d, lcp=True, da=True, u64=u64, lib=lib)
            assert r >= 1 and np.array_equal(sa.astype(np.uint64), a) and np.array_equal(lcp.astype(np.uint64), b)
            assert np.array_equal(da.astype(np.int64), np.cumsum(np.concatenate(([0], d[:-1] == 1)))[sa.astype(np.int64)])
            if G is not None and u64:
                SA = np.zeros(n, np.uint64); LCP = np.zeros(n, np.int64); DA = np.zeros(n, np.int64)
                G.gsacak(d.ctypes.data_as(C.c_void_p), SA.ctypes.data_as(C.c_void_p), LCP.ctypes.data_as(C.c_void_p), DA.ctypes.data_as(C.c_void_p), n)
                assert np.array_equal(sa, SA) and np.array_equal(lcp, LCP) and np.array_equal(da, DA)
        sa2, l2, d2, _ = pfbwt_hip.gsacak(d, lib=lib)                      # SA only
        assert l2 is None and d2 is None and np.array_equal(sa2.astype(np.uint64), a)
    # any byte alphabet (another caller of gsa/gsacak.h:86-96): protein-like strings, all 254 other byte values, runs of one byte
    rng = np.random.default_rng(9)
    others = []
    for alpha, nstr, mx in ((list(b"ACDEFGHIKLMNPQRSTVWY"), 40, 300), (list(range(2, 256)), 25, 500), (list(b"ab"), 30, 200)):
        parts = []
        for _ in range(nstr):
            t = rng.choice(alpha, int(rng.integers(1, mx))).astype(np.uint8)
            if rng.random() < 0.3:
                t[: t.size // 2] = t[0]                                    # a long run of one byte
            parts += [t, np.array([1], np.uint8)]
        parts += parts[:4]                                                 # repeated strings: byte-identical suffixes in position order
        others.append(np.concatenate(parts + [np.array([0], np.uint8)]))
    for d in others:
        n = d.size
        sa, lcp, da, r = pfbwt_hip.gsacak(d, lcp=True, da=True, u64=True, lib=lib)
        assert r >= 1
        if G is not None:
            SA = np.zeros(n, np.uint64); LCP = np.zeros(n, np.int64); DA = np.zeros(n, np.int64)
            G.gsacak(d.ctypes.data_as(C.c_void_p), SA.ctypes.data_as(C.c_void_p), LCP.ctypes.data_as(C.c_void_p), DA.ctypes.data_as(C.c_void_p), n)
            assert np.array_equal(sa, SA) and np.array_equal(lcp, LCP) and np.array_equal(da, DA)
        else:      # (GPU box: no reference build) suffixes in order up to their separators, ties by position
            key = lambda x: (bytes(d[x:]).split(b"\x01")[0] + b"\x01" if 1 in d[x:] else bytes(d[x:]), x)
            idx = sa.astype(np.int64)
            step = max(1, n // 400)
            assert all(key(int(idx[i])) < key(int(idx[i + 1])) for i in range(0, n - 1, step))
            assert sorted(idx.tolist()) == list(range(n))
    bad = dicts[0].copy(); bad[5] = 0
    with pytest.raises(pfbwt_hip.PfpError):
        pfbwt_hip.gsacak(bad, lib=lib)                                    # a second terminator: -1


def test_emu_gsacak_dropin(emu_factory):
    _gsacak_check(EMU_SO, ["edge", "w4p7"])
