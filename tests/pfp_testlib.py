"""Shared test helpers: oracle bindings (ctypes over oracle/liboracle.so), FASTA reading, file images.

TEST INFRASTRUCTURE.  The oracle is the checker, never the thing under test in the gpu tests.
"""
import ctypes as C
import gzip
import hashlib
import json
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))

ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
EMU_SO = os.path.join(ROOT, "tests", "emu", "build", "libpfbwt_emu.so")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def build_oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_SO


class OrcParse(C.Structure):
    _fields_ = [("n", C.c_uint64), ("m", C.c_uint64), ("dwords", C.c_uint64), ("dsize", C.c_uint64),
                ("text", C.POINTER(C.c_uint8)), ("dict", C.POINTER(C.c_uint8)), ("occ", C.POINTER(C.c_uint64)),
                ("parse", C.POINTER(C.c_uint32)), ("last", C.POINTER(C.c_uint8)), ("sai", C.POINTER(C.c_uint64)),
                ("bwlast", C.POINTER(C.c_uint8)), ("ilist", C.POINTER(C.c_uint64)), ("bwsai", C.POINTER(C.c_uint64)),
                ("err", C.c_int), ("err_pos", C.c_uint64), ("err_char", C.c_int)]


_orc = None


def oracle():
    global _orc
    if _orc is None:
        L = C.CDLL(build_oracle())
        L.orc_wang_hash.restype = C.c_uint64
        L.orc_wang_hash.argtypes = [C.c_uint64]
        L.orc_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint, C.POINTER(OrcParse)]
        L.orc_parse_bwt.argtypes = [C.POINTER(OrcParse)]
        L.orc_parse_free.argtypes = [C.POINTER(OrcParse)]
        L.orc_bwt.restype = C.c_int64
        L.orc_bwt.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                              C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_outfn.restype = C.c_uint64
        L.orc_outfn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sais_int.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        _orc = L
    return _orc


def _np(ptr, n, dt):
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dt, copy=True) if n else np.zeros(0, dt)


def fasta_records(path):
    recs, cur = [], None
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if line.startswith(">"):
                cur = [line[1:].split()[0] if line[1:].split() else "", []]
                recs.append(cur)
            elif cur is not None:
                cur[1].append(line)
    return [(n, "".join(s).encode()) for n, s in recs]


def oracle_run(seqs, w=10, p=100, U=8, non_acgt_to_a=False, want_sa=True):
    """Full CPU restatement.  Returns dict of numpy arrays (index-valued arrays as uint64)."""
    L = oracle()
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8) if seqs else np.zeros(0, np.uint8)
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    ps = OrcParse()
    rc = L.orc_parse(cat.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), len(seqs), w, p, 1 if non_acgt_to_a else 0, C.byref(ps))
    if rc == 1:
        return {"err": "invalid_char", "err_pos": ps.err_pos, "err_char": ps.err_char}
    assert rc == 0, rc
    out = {"n": ps.n, "m": ps.m, "dwords": ps.dwords, "dsize": ps.dsize,
           "text": _np(ps.text, ps.n, np.uint8), "dict": _np(ps.dict, ps.dsize, np.uint8), "occ": _np(ps.occ, ps.dwords, np.uint64),
           "parse": _np(ps.parse, ps.m, np.uint32), "last": _np(ps.last, ps.m, np.uint8), "sai": _np(ps.sai, ps.m, np.uint64)}
    rc = L.orc_parse_bwt(C.byref(ps))
    if rc == 2:
        out["err"] = "one_word"
        L.orc_parse_free(C.byref(ps))
        return out
    nr = ps.m + 1
    out.update(bwlast=_np(ps.bwlast, nr, np.uint8), ilist=_np(ps.ilist, nr, np.uint64), bwsai=_np(ps.bwsai, nr, np.uint64))
    nout = ps.n + 1
    bwt = np.zeros(nout, np.uint8)
    sa_raw = np.zeros(nout, np.uint64)
    easy, hard = C.c_uint64(0), C.c_uint64(0)
    got = L.orc_bwt(ps.dict, ps.dsize, ps.occ, ps.dwords, ps.bwlast, ps.ilist, ps.bwsai, nr, w, U,
                    bwt.ctypes.data_as(C.c_void_p), sa_raw.ctypes.data_as(C.c_void_p) if want_sa else None, C.byref(easy), C.byref(hard))
    assert got == nout, (got, nout)
    sa = np.zeros(nout, np.uint64); ssa = np.zeros(2 * nout, np.uint64); esa = np.zeros(2 * nout, np.uint64)
    r = L.orc_outfn(bwt.ctypes.data_as(C.c_void_p), sa_raw.ctypes.data_as(C.c_void_p) if want_sa else None, nout, ps.n, U,
                    sa.ctypes.data_as(C.c_void_p), ssa.ctypes.data_as(C.c_void_p), esa.ctypes.data_as(C.c_void_p))
    out.update(bwt=bwt, sa=sa, ssa=ssa[:2 * r].copy(), esa=esa[:2 * r].copy(), r=int(r), easy=easy.value, hard=hard.value)
    L.orc_parse_free(C.byref(ps))
    return out


FILE_KINDS = {"dict": "u8", "occ": "U", "parse": "u32", "bwlast": "u8", "ilist": "U", "bwsai": "U", "bwt": "u8", "sa": "U", "ssa": "U", "esa": "U"}


def file_image(arr, kind, U):
    """Bytes of the on-disk file (include/pfbwt_io.hpp:44-82) for an array."""
    if kind == "u8":
        return np.asarray(arr, np.uint8).tobytes()
    if kind == "u32":
        return np.asarray(arr, "<u4").tobytes()
    return np.asarray(arr).astype("<u4" if U == 4 else "<u8").tobytes()


def images(res, U, names=None):
    out = {}
    for k, kind in FILE_KINDS.items():
        if (names is None or k in names) and res.get(k) is not None:
            out[k] = file_image(res[k], kind, U)
    if "n" in res and (names is None or "n" in names):
        out["n"] = ("%d\n" % res["n"]).encode()
    return out


def golden_cases():
    return sorted(d for d in os.listdir(GOLDEN) if os.path.isdir(os.path.join(GOLDEN, d)))


def golden_case(name):
    d = os.path.join(GOLDEN, name)
    man = json.load(open(os.path.join(d, "manifest.json")))
    fa = os.path.join(d, "input.fa")
    if not os.path.exists(fa):
        fa += ".gz"
    return man, fasta_records(fa)


def engine_run(ctx_factory, seqs, w, p, U, non_acgt_to_a=False, sa=True, rssa=True):
    """Run the HIP engine (through the C ABI) end to end and collect every array."""
    ctx = ctx_factory(w=w, p=p, u64=(U == 8), non_acgt_to_a=non_acgt_to_a, sai=True)
    try:
        for s in seqs:
            ctx.feed(s, True)
        sz = ctx.finalize()
        res = {"n": sz.n, "m": sz.m, "dwords": sz.dwords, "dsize": sz.dsize}
        res.update(ctx.parse_get())
        ctx.parse_bwt()
        res.update(ctx.parse_bwt_get())
        b = ctx.bwt_build(sa=sa, rssa=rssa)
        res.update(ctx.bwt_get())
        res["r"] = b.r
        res["stage_ms"] = ctx.stage_ms()
        return res
    finally:
        ctx.close()


def compare(res, ref, U, names=("dict", "occ", "parse", "last", "sai", "bwlast", "ilist", "bwsai", "bwt", "sa", "ssa", "esa")):
    """Bit-exact comparison of engine arrays with oracle arrays; returns list of mismatching names."""
    bad = []
    for k in ("n", "m", "dwords", "dsize", "r"):
        if k in res and k in ref and int(res[k]) != int(ref[k]):
            bad.append("%s: %d != %d" % (k, res[k], ref[k]))
    for k in names:
        a, b = res.get(k), ref.get(k)
        if a is None or b is None:
            continue
        a = np.asarray(a).astype(np.uint64); b = np.asarray(b).astype(np.uint64)
        if U == 4:
            b = b & np.uint64(0xFFFFFFFF)
        if a.shape != b.shape:
            bad.append("%s: shape %s != %s" % (k, a.shape, b.shape))
        elif not np.array_equal(a, b):
            i = int(np.flatnonzero(a != b)[0])
            bad.append("%s: first diff at %d: got %d want %d (%d diffs)" % (k, i, a[i], b[i], int((a != b).sum())))
    return bad


def ragged_cases(seed=11):
    """Edge inputs (empty records, records shorter than w, hundreds of tiny records, one short record, only A)."""
    rng = np.random.default_rng(seed)
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8))
    return {
        "empty_records": [b"", rnd(300), b"", b"", rnd(50), b""],
        "short_records": [rnd(1), rnd(3), rnd(9), rnd(10), rnd(11), rnd(500)],
        "many_tiny": [rnd(int(rng.integers(0, 25))) for _ in range(400)],
        "single_short": [rnd(37)],
        "all_A": [b"A" * 2000, b"A" * 100],
        "only_pad_first": [b"", rnd(200)],
    }


def check_ragged(factory):
    """Every ragged case, two (w, p) settings: engine == oracle, or both report 'only one dict word'."""
    import pfbwt_hip
    for name, seqs in ragged_cases().items():
        for w, p in ((10, 100), (4, 5)):
            ref = oracle_run(seqs, w=w, p=p, U=4)
            if ref.get("err") == "one_word":
                try:
                    engine_run(factory, seqs, w, p, 4)
                    raise AssertionError("%s: engine accepted a one-word parse" % name)
                except pfbwt_hip.PfpError as e:
                    assert e.status == pfbwt_hip.E_ONE_WORD, (name, e)
                continue
            bad = compare(engine_run(factory, seqs, w, p, 4), ref, 4)
            assert bad == [], (name, w, p, bad)


def random_cases(seed, count):
    """Seeded random collections for differential testing: panels of mutated copies (the repetitive case), unrelated
    records, runs of N and of one base, IUPAC / lower-case letters (with non_acgt_to_a), every w from 1 to 12 and 32,
    small and large p, both uint_t widths, every output combination."""
    rng = np.random.default_rng(seed)
    out = []
    for c in range(count):
        w = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 12, 32])); p = int(rng.choice([2, 3, 5, 7, 11, 20, 50, 100]))
        kind = int(rng.integers(0, 4))
        base = rng.choice(list(b"ACGT"), int(rng.integers(200, 4000))).astype(np.uint8)
        seqs = []
        if kind == 0:      # a panel: copies of one sequence with point changes, a few insertions / deletions
            for h in range(int(rng.integers(2, 12))):
                b = base.copy()
                for _ in range(int(rng.integers(0, 12))): b[int(rng.integers(0, b.size))] = rng.choice(list(b"ACGT"))
                if rng.random() < 0.5: k = int(rng.integers(0, b.size)); b = np.concatenate([b[:k], rng.choice(list(b"ACGT"), int(rng.integers(1, 30))).astype(np.uint8), b[k:]])
                if rng.random() < 0.5: k = int(rng.integers(0, b.size - 40)); b = np.concatenate([b[:k], b[k + int(rng.integers(1, 40)):]])
                seqs.append(bytes(b))
        elif kind == 1:    # unrelated records of very different lengths
            seqs = [bytes(rng.choice(list(b"ACGT"), int(n)).astype(np.uint8)) for n in rng.integers(0, 1500, int(rng.integers(1, 9)))]
            if sum(len(x) for x in seqs) < 50: seqs.append(bytes(base))
        elif kind == 2:    # runs: N stretches, homopolymers, a short tandem repeat
            b = base.copy(); k = int(rng.integers(0, b.size // 2)); b[k:k + int(rng.integers(20, 600))] = ord("N")
            seqs = [bytes(b), b"A" * int(rng.integers(20, 300)) + bytes(base[:100]) + b"T" * int(rng.integers(20, 300)), b"ACG" * int(rng.integers(10, 200))]
        else:              # letters outside ACGTN, lower case (valid with non_acgt_to_a)
            b = base.copy()
            for _ in range(20): b[int(rng.integers(0, b.size))] = rng.choice(list(b"RYKMacgtn"))
            seqs = [bytes(b), bytes(base[::2])]
        out.append(dict(seqs=seqs, w=w, p=p, U=int(rng.choice([4, 8])), non_acgt_to_a=(kind == 3), sa=bool(rng.integers(0, 2)), rssa=bool(rng.integers(0, 2))))
    return out


def check_random(factory, seed, count):
    """engine == oracle on every array of every random case (or both refuse a one-word parse)"""
    import pfbwt_hip
    for ci, c in enumerate(random_cases(seed, count)):
        ref = oracle_run(c["seqs"], w=c["w"], p=c["p"], U=c["U"], non_acgt_to_a=c["non_acgt_to_a"])
        tag = (seed, ci, c["w"], c["p"], c["U"], c["sa"], c["rssa"], [len(x) for x in c["seqs"]][:6])
        if ref.get("err") == "one_word":
            try:
                engine_run(factory, c["seqs"], c["w"], c["p"], c["U"], non_acgt_to_a=c["non_acgt_to_a"])
                raise AssertionError("engine accepted a one-word parse: %r" % (tag,))
            except pfbwt_hip.PfpError as e:
                assert e.status == pfbwt_hip.E_ONE_WORD, (tag, e)
            continue
        res = engine_run(factory, c["seqs"], c["w"], c["p"], c["U"], non_acgt_to_a=c["non_acgt_to_a"], sa=c["sa"], rssa=c["rssa"])
        names = ["dict", "occ", "parse", "last", "sai", "bwlast", "ilist", "bwsai", "bwt"] + (["sa"] if c["sa"] else []) + (["ssa", "esa"] if c["rssa"] else [])
        bad = compare(res, ref, c["U"], names=tuple(names))
        assert bad == [] and res["r"] == ref["r"], (tag, bad)
