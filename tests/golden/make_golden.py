#!/usr/bin/env python3
"""Generate tests/golden fixtures from the REFERENCE ITSELF (run in the build container only).

Inputs  : /root/reference/tests/data/{single_chrom,mult_chroms}.{bwt,sa}  (the reference's own goldens)
          /root/reference/tests/data/mult_chroms.fa
Tools   : oracle/_ref/merge_pfp{32,64}  (compiled from src/merge_pfp.cpp + include/pfparser.hpp ...)
          oracle/_ref/libgsacak{32,64}.so (compiled from gsa/gsacak.c; `sacak` gives SA/BWT of the text)
Outputs : tests/golden/<case>/{input.fa, manifest.json, [small files]}  -- data only, no reference source.

The .bwt/.sa goldens of the reference pin the end of the path; the text they describe is recovered
from them (T[SA[i]-1] = BWT[i]).  `.ssa/.esa` are derived from (.bwt,.sa) by the
rule of src/pfbwt-f.cpp:298-328 restated in `run_samples` below.
"""
import ctypes, gzip, hashlib, json, os, random, shutil, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
RB = os.path.join(ROOT, "oracle", "_ref")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


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
    return [(n, "".join(s)) for n, s in recs]


def text_of(recs, w):
    return "".join(s.upper() + "A" * w for _, s in recs).encode()


def ref_sa(text, U):
    lib = ctypes.CDLL(os.path.join(RB, "libgsacak%d.so" % (U * 8)))
    n = len(text)
    buf = ctypes.create_string_buffer(text + b"\0", n + 1)
    dt = np.uint32 if U == 4 else np.uint64
    SA = np.zeros(n + 1, dtype=dt)
    lib.sacak.restype = ctypes.c_int
    r = lib.sacak(buf, SA.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n + 1) if U == 8 else ctypes.c_uint32(n + 1))
    assert r >= 0
    return SA


def run_samples(bwt, sa, n):
    """src/pfbwt-f.cpp:298-328: run starts -> .ssa, run ends -> .esa (row, sa) pairs."""
    starts = np.flatnonzero(np.concatenate(([True], bwt[1:] != bwt[:-1])))
    if bwt[0] == 0:  # pbwtc starts at 0
        starts = starts[1:]
    ends = np.concatenate((starts[1:] - 1, [len(bwt) - 1]))
    ssa = np.stack([starts, sa[starts]], axis=1).astype(sa.dtype)
    esa = np.stack([ends, sa[ends]], axis=1).astype(sa.dtype)
    return len(starts), ssa.ravel(), esa.ravel()


def make_case(name, fasta_path, w, p, keep_files=False, extra=None):
    out = os.path.join(HERE, name)
    os.makedirs(out, exist_ok=True)
    shutil.copyfile(fasta_path, os.path.join(out, "input.fa"))
    recs = fasta_records(fasta_path)
    X = text_of(recs, w)
    n = len(X)
    man = {"case": name, "w": w, "p": p, "n": n, "nseq": len(recs), "files": {}}
    if extra:
        man.update(extra)
    for U in (4, 8):
        tmp = tempfile.mkdtemp()
        pref = os.path.join(tmp, "x")
        subprocess.run([os.path.join(RB, "merge_pfp%d" % (U * 8)), "-w", str(w), "-p", str(p), "-s", "--parse-bwt",
                        "--docs", "-o", pref, fasta_path], check=True, stderr=subprocess.DEVNULL, cwd=tmp)
        files = {}
        for ext in ("dict", "occ", "parse", "n", "docs", "bwlast", "ilist", "bwsai"):
            files[ext] = open(pref + "." + ext, "rb").read()
        SA = ref_sa(X, U)
        assert int(SA[0]) == n
        Xa = np.frombuffer(X, dtype=np.uint8)
        bwt = np.where(SA > 0, Xa[(SA.astype(np.int64) - 1) % max(n, 1)], 0).astype(np.uint8)
        r, ssa, esa = run_samples(bwt, SA, n)
        files["bwt"] = bwt.tobytes(); files["sa"] = SA.tobytes(); files["ssa"] = ssa.tobytes(); files["esa"] = esa.tobytes()
        man["files"]["u%d" % (U * 8)] = {k: {"sha256": sha(v), "size": len(v)} for k, v in files.items()}
        if U == 8:
            man["m"] = len(files["parse"]) // 4
            man["dwords"] = len(files["occ"]) // 8
            man["dsize"] = len(files["dict"])
            man["r"] = int(r)
            if keep_files:
                for k, v in files.items():
                    open(os.path.join(out, "u64." + k), "wb").write(v)
        shutil.rmtree(tmp)
    json.dump(man, open(os.path.join(out, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(name, {k: man[k] for k in ("n", "m", "dwords", "dsize", "r")})
    return man


def recover_text(name):
    """rebuild the text the reference's golden describes."""
    D = os.path.join(REF, "tests", "data")
    bwt = np.fromfile(os.path.join(D, name + ".bwt"), dtype=np.uint8)
    sa = np.array(open(os.path.join(D, name + ".sa")).read().split(), dtype=np.int64)
    n = int(sa[0])
    T = np.zeros(n, dtype=np.uint8)
    mk = sa > 0
    T[sa[mk] - 1] = bwt[mk]
    assert bytes(T[-10:]) == b"A" * 10
    return bytes(T[:-10]), bwt, sa


def main():
    tmp = tempfile.mkdtemp()
    # G1/G2: the reference's own .bwt/.sa goldens
    for name in ("single_chrom", "mult_chroms"):
        seq, bwt, sa = recover_text(name)
        fa = os.path.join(tmp, name + ".recon.fa")
        open(fa, "wb").write(b">recon\n" + seq + b"\n")
        man = make_case(name, fa, 10, 100, keep_files=False, extra={"source": "reference tests/data/%s.{bwt,sa}" % name})
        # the reference goldens themselves (data files held by the reference's tests)
        out = os.path.join(HERE, name)
        with gzip.open(os.path.join(out, "reference_golden.bwt.gz"), "wb", 9) as f:
            f.write(bwt.tobytes())
        with gzip.open(os.path.join(out, "reference_golden.sa.u64.gz"), "wb", 9) as f:
            f.write(sa.astype("<u8").tobytes())
        assert man["files"]["u64"]["bwt"]["sha256"] == sha(bwt.tobytes())
        assert man["files"]["u64"]["sa"]["sha256"] == sha(sa.astype("<u8").tobytes())
    # G2b: tests/data/mult_chroms.fa parsed directly (3 records, docs)
    make_case("mult_chroms_fa", os.path.join(REF, "tests", "data", "mult_chroms.fa"), 10, 100, keep_files=True)
    # G3: edge FASTA -- lowercase, N run, '-' char, two records, multi-line, p=20
    rng = random.Random(7)
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
    wrap = lambda s, k: "\n".join(s[i:i + k] for i in range(0, len(s), k))
    s1 = rnd(900) + "n" * 30 + "N" * 27 + rnd(400).lower() + "--" + rnd(400)
    s2 = rnd(1700)
    fa = os.path.join(tmp, "edge.fa")
    open(fa, "w").write(">e1 first record\n" + wrap(s1, 70) + "\n>e2\n" + wrap(s2, 61) + "\n")
    make_case("edge", fa, 10, 20, keep_files=True)
    # G3b: small window / modulus, forces many short phrases and many multi-word suffix groups
    fa = os.path.join(tmp, "w4.fa")
    base = rnd(3000)
    haps = []
    for h in range(6):
        b = list(base)
        for i in range(rng.randint(0, 40), len(b), rng.randint(50, 200)):
            b[i] = rng.choice("ACGT")
        haps.append("".join(b))
    open(fa, "w").write("".join(">h%d\n%s\n" % (i, wrap(h, 80)) for i, h in enumerate(haps)))
    make_case("w4p7", fa, 4, 7, keep_files=True)
    # G4: repetitive panel, 8 haplotypes x 250 kbase (digests only)
    fa = os.path.join(tmp, "panel.fa")
    base = rnd(250000)
    haps = []
    for h in range(8):
        b = list(base)
        for i in range(rng.randint(0, 300), len(b), rng.randint(300, 900)):
            b[i] = rng.choice("ACGT")
        haps.append("".join(b))
    with gzip.open(os.path.join(tmp, "panel.fa.gz"), "wt") as f:
        f.write("".join(">hap%d\n%s\n" % (i, wrap(h, 60000)) for i, h in enumerate(haps)))
    open(fa, "w").write("".join(">hap%d\n%s\n" % (i, wrap(h, 60000)) for i, h in enumerate(haps)))
    make_case("panel8", fa, 10, 100, keep_files=False)
    # store the panel input compressed (2 MB -> ~0.6 MB)
    pdir = os.path.join(HERE, "panel8")
    with open(os.path.join(pdir, "input.fa"), "rb") as fi, gzip.open(os.path.join(pdir, "input.fa.gz"), "wb", 9) as fo:
        fo.write(fi.read())
    os.remove(os.path.join(pdir, "input.fa"))
    # G5: wang_hash known answers -- computed by compiling hash.hpp's function
    kat_src = os.path.join(tmp, "kat.cpp")
    open(kat_src, "w").write('#include <cstdio>\n#include "hash.hpp"\nint main(){unsigned long long k[]={0ULL,1ULL,0xfffffULL,0x1b1b1ULL,0xffffffffffffffffULL,0x123456789abcdefULL};'
                             'for(auto x:k) printf("%llu %llu\\n",x,(unsigned long long)wang_hash(x));}\n')
    exe = os.path.join(tmp, "kat")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + REF + "/include", kat_src, os.path.join(RB, "utils.o"), "-o", exe], check=True)
    rows = [l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()]
    json.dump({"wang_hash": [[int(a), int(b)] for a, b in rows]}, open(os.path.join(HERE, "wang_hash_kat.json"), "w"), indent=1)
    print("wang_hash KATs:", rows[:2])
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
