"""The host-side mirror of the reference interface (pfbwt-f_amd/host: pfbwtf::PfParser<>, PrefixFreeBWT<>,
pfbwt_io, the pfbwt-f / pfbwt-f64 / merge_pfp command lines) checked file-by-file against the digests of
the files the REFERENCE wrote (tests/golden manifests).

* CPU (`not gpu`): binaries linked against tests/emu (debugging harness) -- exercises the host C++ only.
* GPU (`gpu`): the product binaries pfbwt-f_amd/bin/* linked against libpfbwt_hip.so.
"""
import gzip
import hashlib
import json
import os
import subprocess
import pytest
from pfp_testlib import GOLDEN, ROOT, fasta_records

PARSE_FILES = ("dict", "occ", "parse", "n", "docs", "bwlast", "ilist", "bwsai")
ALL_FILES = PARSE_FILES + ("bwt", "sa", "ssa", "esa")


def sha_f(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def bins(kind):
    if kind == "asan":
        subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "asan"], check=True, stdout=subprocess.DEVNULL)
        d = os.path.join(ROOT, "tests", "emu", "build")
        return {"pfbwt-f": os.path.join(d, "pfbwt-f-asan"), "pfbwt-f64": os.path.join(d, "pfbwt-f64-asan"), "merge_pfp": os.path.join(d, "merge_pfp-asan")}
    if kind == "emu":
        subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu-host"], check=True, stdout=subprocess.DEVNULL)
        d = os.path.join(ROOT, "tests", "emu", "build")
        return {"pfbwt-f": os.path.join(d, "pfbwt-f-emu"), "pfbwt-f64": os.path.join(d, "pfbwt-f64-emu"), "merge_pfp": os.path.join(d, "merge_pfp-emu"), "mps_to_ma": os.path.join(d, "mps_to_ma-emu")}
    d = os.path.join(ROOT, "pfbwt-f_amd", "bin")
    for b in ("pfbwt-f", "pfbwt-f64", "merge_pfp"):
        assert os.path.exists(os.path.join(d, b)), "product binaries missing: make -C pfbwt-f_amd"
    return {b: os.path.join(d, b) for b in ("pfbwt-f", "pfbwt-f64", "merge_pfp", "mps_to_ma")}


def manifest(name):
    return json.load(open(os.path.join(GOLDEN, name, "manifest.json")))


def input_fa(name, tmp):
    fa = os.path.join(GOLDEN, name, "input.fa")
    if os.path.exists(fa):
        return fa
    out = os.path.join(tmp, name + ".fa")
    open(out, "wb").write(gzip.open(fa + ".gz").read())
    return out


def run(cmd, check=True, **kw):
    pr = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert pr.returncode == 0 or not check, pr.stderr[-2000:]
    return pr


def check_cli(B, tmp, cases):
    for name, exe, U in cases:
        man = manifest(name)
        pref = os.path.join(tmp, "o_" + name + str(U))
        pr = run([B[exe], "-s", "-r", "--print-docs", "-w", str(man["w"]), "-p", str(man["p"]), "-o", pref, input_fa(name, tmp)])
        mf = man["files"]["u%d" % (U * 8)]
        for e in ALL_FILES:
            assert sha_f(pref + "." + e) == mf[e]["sha256"], (name, e)
        assert "n: %d" % man["n"] in pr.stderr and "r: %d" % man["r"] in pr.stderr and "TASK\tparsing input\t" in pr.stderr


def check_stages_and_merge(B, tmp, merge_from_files=True):
    man = manifest("edge"); mf = man["files"]["u64"]; fa = input_fa("edge", tmp)
    gz = os.path.join(tmp, "e.fa.gz")
    with open(fa, "rb") as fi, gzip.open(gz, "wb") as fo:
        fo.write(fi.read())
    po = os.path.join(tmp, "po")
    run([B["pfbwt-f64"], "--parse-only", "-s", "-w", "10", "-p", "20", "-o", po, gz])          # gz input, --parse-only
    assert not os.path.exists(po + ".bwt")
    run([B["pfbwt-f64"], "--pfbwt-only", "-s", "-r", "-w", "10", "-p", "20", "-o", po])       # resumes from the files (+ .n)
    for e in ("dict", "occ", "parse", "bwlast", "ilist", "bwsai", "bwt", "sa", "ssa", "esa"):
        assert sha_f(po + "." + e) == mf[e]["sha256"], e
    si = os.path.join(tmp, "si")
    with open(fa, "rb") as fi, open(si + ".stdout", "wb") as fo:                                    # stdin input, -c bwt
        pr = subprocess.run([B["pfbwt-f64"], "-s", "-w", "10", "-p", "20", "-o", si, "--stdout", "bwt"], stdin=fi, stdout=fo, stderr=subprocess.PIPE)
    assert pr.returncode == 0
    assert sha_f(si + ".stdout") == mf["bwt"]["sha256"] and sha_f(si + ".sa") == mf["sa"]["sha256"]
    # BWT only (no SA): same .bwt
    nb = os.path.join(tmp, "nb")
    run([B["pfbwt-f64"], "-w", "10", "-p", "20", "-o", nb, fa])
    assert sha_f(nb + ".bwt") == mf["bwt"]["sha256"] and not os.path.exists(nb + ".sa")
    # -r without -s: run samples only (generate_bwt_lcp takes the rows' SA values from the samples, no full SA on the host)
    ro = os.path.join(tmp, "ro")
    run([B["pfbwt-f64"], "-r", "-w", "10", "-p", "20", "-o", ro, fa])
    for e in ("bwt", "ssa", "esa"):
        assert sha_f(ro + "." + e) == mf[e]["sha256"], e
    assert not os.path.exists(ro + ".sa")
    # merge_pfp: three records parsed separately, merged == the single parse (tests/test_parser.cpp:188-234)
    mf = manifest("mult_chroms_fa")["files"]["u64"]
    parts = []
    for i, (nm, s) in enumerate(fasta_records(os.path.join(GOLDEN, "mult_chroms_fa", "input.fa"))):
        p = os.path.join(tmp, "part%d.fa" % i)
        open(p, "wb").write(b">" + nm.encode() + b"\n" + s + b"\n"); parts.append(p)
    mg = os.path.join(tmp, "merged")
    run([B["merge_pfp"], "-w", "10", "-p", "100", "-s", "--parse-bwt", "--docs", "-o", mg] + parts)
    for e in PARSE_FILES:
        assert sha_f(mg + "." + e) == mf[e]["sha256"], e
    if merge_from_files:
        for p in parts:                                                                          # now from saved .dict/.parse
            run([B["pfbwt-f64"], "--parse-only", "--print-docs", "-s", "-o", p, p])
        mg2 = os.path.join(tmp, "merged2")
        run([B["merge_pfp"], "-w", "10", "-p", "100", "-s", "--parse-bwt", "--docs", "-o", mg2] + parts)
        for e in PARSE_FILES:
            assert sha_f(mg2 + "." + e) == mf[e]["sha256"], e
    # merge_pfp with an operand shorter than a phrase (no trigger window: its parse is ONE phrase) in first, middle and last
    # position, as FASTA and as saved .dict/.parse, and a single operand: == the parse of the concatenated records
    import numpy as np
    rng = np.random.default_rng(5)
    recs = [bytes(rng.choice(list(b"ACGT"), k).astype(np.uint8)) for k in (3000, 30, 2000)]
    for order in ((0, 1, 2), (1, 0, 2), (0, 2, 1), (1,)):
        ops = []
        for k, i in enumerate(order):
            q = os.path.join(tmp, "sh%s_%d.fa" % ("".join(map(str, order)), k))
            open(q, "wb").write(b">r%d\n" % i + recs[i] + b"\n"); ops.append(q)
        whole = os.path.join(tmp, "whole%s.fa" % "".join(map(str, order)))
        open(whole, "wb").write(b"".join(open(q, "rb").read() for q in ops))
        run([B["pfbwt-f64"], "--parse-only", "-s", "--print-docs", "-o", whole, whole] if len(order) > 1 else [B["pfbwt-f64"], "--parse-only", "--print-docs", "-o", whole, whole], **({} if len(order) > 1 else {"check": False}))
        for saved in ((False, True) if merge_from_files else (False,)):
            if saved:
                for q in ops:
                    subprocess.run([B["pfbwt-f64"], "--parse-only", "--print-docs", "-s", "-o", q, q], capture_output=True)      # a one-phrase operand stops at "only one dict word" AFTER .dict/.parse are written
                    assert os.path.exists(q + ".dict") and os.path.exists(q + ".parse")
            mo = os.path.join(tmp, "mshort%s%d" % ("".join(map(str, order)), saved))
            run([B["merge_pfp"], "-w", "10", "-p", "100", "-s", "--docs", "-o", mo] + ops)
            for e in ("dict", "occ", "parse", "docs"):
                assert sha_f(mo + "." + e) == sha_f(whole + "." + e), (order, saved, e)
    # error behaviour: message and exit status of include/hash.hpp:31
    bad = os.path.join(tmp, "bad.fa")
    open(bad, "w").write(">x\nACGTACGTRACGTACGTACGTAAAACCCCGGGGTTTT\n")
    pr = subprocess.run([B["pfbwt-f64"], "-o", os.path.join(tmp, "bad"), bad], capture_output=True, text=True)
    assert pr.returncode == 1 and "error, invalid character 82/R -> 5" in pr.stderr
    pr = subprocess.run([B["pfbwt-f64"], "-w", "40", "-o", os.path.join(tmp, "bad"), fa], capture_output=True, text=True)
    assert pr.returncode == 1 and "window size w must be < 32!" in pr.stderr


def check_chained_recipe(B, tmp, case="mult_chroms"):
    """The command recipe of the reference's driver, vcf_to_bwt.py:118-131, 174-181, 236-285 (configs[4] of BASELINE.json), as ONE
    chain: every haplotype through `pfbwt-f64 --non-acgt-to-a --parse-only --print-docs -s` (one from stdin, as behind vcf_scan),
    `merge_pfp -w W -s --parse-bwt --docs -t T` over the saved parses, `pfbwt-f64 --pfbwt-only --print-docs -w W -m MOD --stdout sa
    -s -r` piped through `tee O.sa` into `mps_to_ma -o O.ma O.mps -`; .bwt and .sa are diffed with the reference's OWN goldens
    (tests/data/*.bwt, *.sa -- what tests/vcf_to_bwt_test.sh:23-37 diffs), the marker array with the stream that
    tests/test_markers.py pins to the reference's golden .markers.  The haplotypes are the golden text cut at its w-'A' pads
    (vcf_scan / consensus and merge_mps -- the VCF front end -- are out of scope: the merged marker positions are a fixture)."""
    import numpy as np
    man = manifest(case); w = man["w"]
    recs = fasta_records(input_fa(case, tmp))
    text = b"".join(s for _, s in recs) + b"A" * w                      # the golden text: every sequence + w 'A's
    hl = 10000 + w
    assert len(text) == man["n"] and len(text) % hl == 0
    haps = [text[i:i + hl - w] for i in range(0, len(text), hl)]
    assert all(text[i + hl - w:i + hl] == b"A" * w for i in range(0, len(text), hl))
    O = os.path.join(tmp, "chain_" + case)
    prefixes = []
    for h, seq in enumerate(haps):
        pre = "%s.h%d" % (O, h); prefixes.append(pre)
        fa = pre + ".fa"
        open(fa, "wb").write(b">hap%d\n" % h + seq + b"\n")
        cmd = [B["pfbwt-f64"], "--non-acgt-to-a", "--parse-only", "--print-docs", "-s", "-o", pre]
        if h % 2:                                                          # vcf_scan --stdout | pfbwt-f64 ... (no input name: stdin)
            with open(fa, "rb") as fi:
                pr = subprocess.run(cmd, stdin=fi, capture_output=True, text=True)
        else:
            pr = subprocess.run(cmd + [fa], capture_output=True, text=True)
        assert pr.returncode == 0, pr.stderr[-2000:]
        os.remove(fa)                                                      # merge_pfp must load the saved .dict / .parse
    run([B["merge_pfp"], "-w", str(w), "-s", "--parse-bwt", "--docs", "-o", O, "-t", "3"] + prefixes)
    mps = os.path.join(GOLDEN, case, "markers.mps")
    p1 = subprocess.Popen([B["pfbwt-f64"], "--pfbwt-only", "--print-docs", "-o", O, "-w", str(w), "-m", "100", "--stdout", "sa", "-s", "-r"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    p2 = subprocess.Popen(["tee", O + ".sa"], stdin=p1.stdout, stdout=subprocess.PIPE)
    p3 = subprocess.run([B["mps_to_ma"], "-o", O + ".ma", mps, "-"], stdin=p2.stdout, capture_output=True, text=True)
    p2.wait(); err1 = p1.stderr.read().decode(); p1.wait()
    assert p1.returncode == 0 and p2.returncode == 0 and p3.returncode == 0, err1[-1500:] + p3.stderr[-1500:]
    d = os.path.join(GOLDEN, case)
    assert open(O + ".bwt", "rb").read() == gzip.open(os.path.join(d, "reference_golden.bwt.gz")).read()
    assert open(O + ".sa", "rb").read() == gzip.open(os.path.join(d, "reference_golden.sa.u64.gz")).read()
    meta = json.load(open(os.path.join(d, "markers.json")))
    assert sha_f(O + ".ma") == meta["ma_sha256"]
    assert "n: %d" % man["n"] in err1 and "r: %d" % man["r"] in err1
    mf = man["files"]["u64"]
    for e in ("ssa", "esa"):
        assert sha_f(O + "." + e) == mf[e]["sha256"], e


def check_cli_gpus(B, tmp, cases):
    """--gpus N (extension): records sharded over N ranks -- here N contexts on device 0, the rehearsal of the multi-GPU protocol on
    one card -- must write the single-device files"""
    for name, exe, U, ng in cases:
        man = manifest(name); mf = man["files"]["u%d" % (U * 8)]
        pref = os.path.join(tmp, "g_%s%d_%d" % (name, U, ng))
        pr = run([B[exe], "-s", "-r", "--gpus", str(ng), "--devices", ",".join(["0"] * ng), "-w", str(man["w"]), "-p", str(man["p"]), "-o", pref, input_fa(name, tmp)])
        for e in ("bwt", "sa", "ssa", "esa"):
            assert sha_f(pref + "." + e) == mf[e]["sha256"], (name, ng, e)
        assert "n: %d" % man["n"] in pr.stderr and "r: %d" % man["r"] in pr.stderr
        pr = run([B[exe], "-r", "--gpus", str(ng), "--devices", ",".join(["0"] * ng), "-w", str(man["w"]), "-p", str(man["p"]), "-o", pref + "r", input_fa(name, tmp)])
        for e in ("bwt", "ssa", "esa"):
            assert sha_f(pref + "r." + e) == mf[e]["sha256"], (name, ng, e)


def test_cli_emu(tmp_path):
    B = bins("emu")
    check_cli(B, str(tmp_path), [("edge", "pfbwt-f64", 8), ("mult_chroms_fa", "pfbwt-f", 4)])
    check_stages_and_merge(B, str(tmp_path))


def test_chained_recipe_emu(tmp_path):
    check_chained_recipe(bins("emu"), str(tmp_path))


def test_cli_gpus_emu(tmp_path):
    check_cli_gpus(bins("emu"), str(tmp_path), [("mult_chroms_fa", "pfbwt-f64", 8, 3), ("mult_chroms_fa", "pfbwt-f", 4, 2), ("edge", "pfbwt-f64", 8, 2)])


def test_cli_asan_ubsan(tmp_path, monkeypatch):
    """the host mirror + command lines built with -fsanitize=address,undefined (`make asan`), engine = tests/emu: any
    sanitizer report makes the binary exit non-zero, which fails the run() helper.  Leak checking is on; the emulator's
    cached fiber stacks are the one suppressed allocation site."""
    sup = tmp_path / "lsan.supp"
    sup.write_text("leak:emu::run_block\n")
    monkeypatch.setenv("ASAN_OPTIONS", "detect_leaks=1:abort_on_error=0:exitcode=66")
    monkeypatch.setenv("LSAN_OPTIONS", "suppressions=%s:print_suppressions=0" % sup)
    monkeypatch.setenv("UBSAN_OPTIONS", "halt_on_error=1:print_stacktrace=1")
    B = bins("asan")
    check_cli(B, str(tmp_path), [("edge", "pfbwt-f64", 8)])
    check_stages_and_merge(B, str(tmp_path), merge_from_files=False)


@pytest.mark.gpu
def test_cli_gpu(tmp_path):
    B = bins("gpu")
    check_cli(B, str(tmp_path), [(n, exe, U) for n in ("edge", "w4p7", "mult_chroms_fa", "single_chrom", "mult_chroms", "panel8") for exe, U in (("pfbwt-f64", 8), ("pfbwt-f", 4))])
    check_stages_and_merge(B, str(tmp_path))


@pytest.mark.gpu
def test_cli_gpus_gpu(tmp_path):
    B = bins("gpu")
    check_cli_gpus(B, str(tmp_path), [("mult_chroms_fa", "pfbwt-f64", 8, 3), ("panel8", "pfbwt-f64", 8, 4), ("panel8", "pfbwt-f", 4, 2), ("edge", "pfbwt-f64", 8, 2)])
    man = manifest("panel8"); mf = man["files"]["u64"]      # one rank: the library talks to RCCL (world size 1)
    pref = os.path.join(str(tmp_path), "g1")
    run([B["pfbwt-f64"], "-s", "-r", "--gpus", "1", "-w", str(man["w"]), "-p", str(man["p"]), "-o", pref, input_fa("panel8", str(tmp_path))])
    for e in ("bwt", "sa", "ssa", "esa"):
        assert sha_f(pref + "." + e) == mf[e]["sha256"], e


@pytest.mark.gpu
def test_chained_recipe_gpu(tmp_path):
    B = bins("gpu")
    for case in ("mult_chroms", "single_chrom"):
        check_chained_recipe(B, str(tmp_path), case)


# ---- the drop-in claim itself: the reference's UNCHANGED src/pfbwt-f.cpp and src/merge_pfp.cpp on top of the mirror ----
REF = "/root/reference"
MIRROR_INC = ["-I" + os.path.join(ROOT, "pfbwt-f_amd", "host", "include"), "-I" + os.path.join(ROOT, "include")]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("src", ["pfbwt-f.cpp", "merge_pfp.cpp"])
@pytest.mark.parametrize("m64", [[], ["-DM64"]])
def test_reference_cli_sources_compile_against_mirror(src, m64):
    """g++ -fsyntax-only of the reference's own CLI sources with pfbwt-f_amd/host/include in place of the reference's
    include/ (INTEGRATION.md section A): global die / open_aux_file, PfParser::get_ntab, run_pfbwt's template-template
    parameters -- everything those files name must exist in the mirror"""
    pr = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w"] + m64 + MIRROR_INC + [os.path.join(REF, "src", src)], capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present (GPU box)")
def test_reference_cli_dropin_emu(tmp_path):
    """the same sources BUILT against the mirror and run (engine = tests/emu on the CPU): every file equals the digests of
    the files the reference itself wrote"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu"], check=True, stdout=subprocess.DEVNULL)
    d = os.path.join(ROOT, "tests", "emu", "build")
    B = {}
    for name, src, m64 in (("pfbwt-f", "pfbwt-f.cpp", []), ("pfbwt-f64", "pfbwt-f.cpp", ["-DM64"]), ("merge_pfp", "merge_pfp.cpp", ["-DM64"])):
        B[name] = os.path.join(d, "ref-" + name + "-emu")
        run(["g++", "-O1", "-std=c++17", "-w"] + m64 + MIRROR_INC + ["-o", B[name], os.path.join(REF, "src", src), "-L" + d, "-lpfbwt_emu", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    check_cli(B, str(tmp_path), [("edge", "pfbwt-f", 4)])
    check_stages_and_merge(B, str(tmp_path))


@pytest.mark.gpu
def test_reference_cli_dropin_gpu(tmp_path):
    """oracle/_ref/dropin-*: the reference's unchanged CLI sources compiled (in the build container, oracle/Makefile
    `dropin`) against the mirror and linked with libpfbwt_hip.so, run here on the MI355X"""
    d = os.path.join(ROOT, "oracle", "_ref")
    B = {"pfbwt-f": os.path.join(d, "dropin-pfbwt-f"), "pfbwt-f64": os.path.join(d, "dropin-pfbwt-f64"), "merge_pfp": os.path.join(d, "dropin-merge_pfp")}
    if not all(os.path.exists(p) for p in B.values()):
        pytest.skip("oracle/_ref/dropin-* not built (needs /root/reference at build time)")
    check_cli(B, str(tmp_path), [(n, exe, U) for n in ("edge", "w4p7", "mult_chroms_fa", "single_chrom", "panel8") for exe, U in (("pfbwt-f64", 8), ("pfbwt-f", 4))])
    check_stages_and_merge(B, str(tmp_path))
