"""FASTA ingest on the device (csrc/fasta.h, pfp_parse_feed_fasta): raw file bytes in, the same text as
kseq_read + PfParser::add_fasta (include/kseq.h:178-228, include/pfparser.hpp:300-337) build record by record.
The check is end to end: every array of the build over the raw bytes == the oracle's over the records a plain Python
restatement of kseq's record rules extracts (the BWT determines the text), plus the (header offset, record start) pairs.
* CPU: through tests/emu.  * GPU: the product library."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
from pfp_testlib import EMU_SO, ROOT, compare, oracle_run


def kseq_model(raw):
    """records of a FASTA byte string by kseq's rules: [(offset of the header's first byte, name, sequence)]"""
    a, b = raw.find(b">"), raw.find(b"@")
    start = min(x for x in (a, b) if x >= 0) if (a >= 0 or b >= 0) else -1
    if start < 0:
        return []
    recs, pos = [], start
    for i, ln in enumerate(raw[start:].split(b"\n")):
        if i == 0 or ln[:1] in (b">", b"@"):
            nm = ln[1:].split()
            recs.append([pos, nm[0] if nm else b"", bytearray()])
        else:
            recs[-1][2] += ln.replace(b"\r", b"")
        pos += len(ln) + 1
    return [(o, n, bytes(s)) for o, n, s in recs]


def random_fasta(rng, nrec, maxlen, preamble=True, crlf=False, at_headers=False, trailing_newline=True):
    out = bytearray()
    if preamble:
        out += b"# a comment line without header characters\n\nACGT not a record\n"
    eol = b"\r\n" if crlf else b"\n"
    for k in range(nrec):
        L = int(rng.integers(0, maxlen)) if rng.random() > 0.15 else 0
        hdr = (b"@" if (at_headers and rng.random() < 0.3) else b">") + b"rec%d" % k
        if rng.random() < 0.5:
            hdr += b" some comment with > and @ and + inside"
        out += hdr + eol
        seq = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        width = int(rng.choice([1, 7, 60, 61, 63, 64, 65, 70, 128, 1000, 20000, 10 ** 9]))
        for i in range(0, L, width):
            out += seq[i:i + width] + eol
            if rng.random() < 0.05:
                out += eol                                  # an empty line inside a record
    if not trailing_newline:
        while out[-1:] in (b"\n", b"\r"):
            out = out[:-1]
    return bytes(out)


def check_ingest(factory, raw, w, p, U, pieces, chunk_bytes=0, env_ok=True):
    recs = kseq_model(raw)
    seqs = [s for _, _, s in recs]
    c = factory(w=w, p=p, u64=(U == 8), sai=True)
    if chunk_bytes:
        c.debug_set(fasta_chunk_bytes=chunk_bytes)
    c.reserve(len(raw))
    cuts = sorted(set([0, len(raw)] + [int(x) for x in pieces]))
    got_off, got_pos = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        ro, tp = c.feed_fasta(raw[a:b], final=(b == len(raw)), records=True)
        got_off += [int(x) + a for x in ro]; got_pos += [int(x) for x in tp]
    assert got_off == [o for o, _, _ in recs], (got_off[:5], [o for o, _, _ in recs][:5])
    starts, acc = [], 0
    for s in seqs:
        starts.append(acc); acc += len(s) + w
    assert got_pos == starts
    if acc < 2 * w + 2:
        c.close(); return
    sz = c.finalize()
    assert sz.n == acc
    res = {"n": sz.n, "m": sz.m, "dwords": sz.dwords, "dsize": sz.dsize}
    res.update(c.parse_get())
    ref = oracle_run(seqs, w=w, p=p, U=U)
    if sz.m >= 2:
        c.parse_bwt(); res.update(c.parse_bwt_get())
        b = c.bwt_build(sa=True, rssa=True); res.update(c.bwt_get()); res["r"] = b.r
        assert compare(res, ref, U) == []
    else:
        assert compare(res, ref, U, names=("dict", "occ", "parse")) == []
    c.close()


def run_cases(factory, scale):
    rng = np.random.default_rng(41)
    # many records, every line width, CRLF, '@' headers, no trailing newline; fed whole, in random pieces, through tiny device chunks
    for k, kw in enumerate((dict(), dict(crlf=True), dict(at_headers=True, trailing_newline=False), dict(preamble=False))):
        raw = random_fasta(rng, 25, 6000 * scale, **kw)
        check_ingest(factory, raw, 10, 100, 8, [])
        check_ingest(factory, raw, 4, 7, 4, rng.integers(0, len(raw), 7), chunk_bytes=int(rng.choice([64, 200, 4096, 16384, 50000])))
    # cuts right behind newlines / inside header lines, one-byte pieces at the start
    raw = random_fasta(rng, 6, 3000, crlf=False)
    nl = [i + 1 for i, ch in enumerate(raw) if ch == 10][:40]
    check_ingest(factory, raw, 6, 11, 8, nl + [1, 2, 3], chunk_bytes=64)
    # one long single-line record (tiles without any newline), then short ones
    raw = b">big\n" + bytes(rng.choice(list(b"ACGT"), 70000 * scale).astype(np.uint8)) + b"\n>e1\n>e2\n\n>x\nACGTTTGACCA\n"
    check_ingest(factory, raw, 10, 100, 8, [16384, 16385, 32768 + 5], chunk_bytes=0)
    check_ingest(factory, raw, 10, 100, 8, [], chunk_bytes=16384)
    # nothing but a header; nothing at all; preamble only
    for raw in (b">only a header", b"", b"no header here\nACGT\n"):
        check_ingest(factory, raw, 10, 100, 8, [])


@pytest.fixture(scope="module")
def emu_factory():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu"], check=True, stdout=subprocess.DEVNULL)
    import pfbwt_hip
    return lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw)


def test_fasta_ingest_emu(emu_factory):
    run_cases(emu_factory, 1)


def check_file_ingest(factory, tmp, block_bytes):
    """pfp_parse_feed_fasta_file: plain (parallel pread), gzip and FASTQ inputs, two files in a row, docs"""
    import gzip
    rng = np.random.default_rng(7)
    raw1 = random_fasta(rng, 9, 5000, crlf=True)
    raw2 = random_fasta(rng, 5, 3000, preamble=False, trailing_newline=False)
    fq = b"".join(b"@q%d desc\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 80 + 13 * i).astype(np.uint8)), b"I" * (80 + 13 * i)) for i in range(12))
    files = {"a.fa": raw1, "b.fa.gz": gzip.compress(raw2), "c.fq": fq, "d.fq.gz": gzip.compress(fq)}
    for nm, data in files.items():
        open(os.path.join(tmp, nm), "wb").write(data)
    fq_recs = [(0, b"q%d" % i, fq.split(b"\n")[4 * i + 1]) for i in range(12)]
    for order in (("a.fa", "b.fa.gz"), ("b.fa.gz", "a.fa"), ("c.fq", "a.fa"), ("d.fq.gz",)):
        recs = []
        for nm in order:
            recs += fq_recs if nm.startswith(("c.", "d.")) else kseq_model(raw1 if nm == "a.fa" else raw2)
        seqs = [s for _, _, s in recs]
        c = factory(w=6, p=11, u64=True, sai=True)
        c.debug_set(ingest_block_bytes=block_bytes, fasta_chunk_bytes=block_bytes)
        names, starts, nrec = [], [], 0
        for nm in order:
            info = c.feed_fasta_file(os.path.join(tmp, nm), records=True)
            nrec += info.records
            cnt = C.c_uint64(0); c._check(c.L.pfp_parse_docs(c.h, C.byref(cnt)))
            for i in range(cnt.value):
                pn, ps = C.c_char_p(), C.c_uint64(0)
                c._check(c.L.pfp_parse_doc_get(c.h, i, C.byref(pn), C.byref(ps)))
                names.append(pn.value); starts.append(ps.value)
        assert nrec == len(recs) and names == [n for _, n, _ in recs], (order, names[:4])
        acc, want = 0, []
        for s_ in seqs:
            want.append(acc); acc += len(s_) + 6
        assert starts == want and c.text_length() == acc
        sz = c.finalize(); c.parse_bwt()
        hb = np.zeros(sz.n + 1, np.uint8)
        b = c.bwt_build_stream(hb.ctypes.data, None, rssa=True)
        ssa, esa = c.samples_get()
        ref = oracle_run(seqs, w=6, p=11, U=8)
        assert compare({"bwt": hb, "ssa": ssa, "esa": esa, "r": b.r, "n": sz.n}, ref, 8, names=("bwt", "ssa", "esa")) == [], order
        hb2 = np.full(sz.n + 1, 255, np.uint8)      # the same bytes from the run-length form (one byte per run + host threads)
        for th in (3, 0, 64, 1):      # shares of the bytes: runs are cut at the borders of the shares; 0 = one thread per CPU
            hb2[:] = 255
            c.bwt_get_expanded(hb2.ctypes.data, ssa if th != 64 else None, threads=th)
            assert np.array_equal(hb2, hb), th
        c.close()
    import pfbwt_hip
    c = factory(w=6, p=11, u64=True, sai=True)
    with pytest.raises(pfbwt_hip.PfpError) as e:
        c.feed_fasta_file(os.path.join(tmp, "does-not-exist.fa"))
    assert e.value.status == -9
    c.close()


def test_file_ingest_emu(emu_factory, tmp_path):
    for blk in (4096, 70000):
        check_file_ingest(emu_factory, str(tmp_path), blk)


def fastq_like_is_rejected(factory):
    import pfbwt_hip
    c = factory(w=10, p=100, u64=True, sai=True)
    with pytest.raises(pfbwt_hip.PfpError) as e:
        c.feed_fasta(b"@r1\nACGTACGTACGTACGTACGTAAAA\n+\nIIIIIIIIIIIIIIIIIIIIIIII\n", final=True)
    assert e.value.status == pfbwt_hip.E_ARG and e.value.ch == ord("+")
    c.reset()
    c.feed_fasta(b">r1\nACGTACGTACGTACGTACGTAAAA\n", final=True)          # the context is usable again
    assert c.finalize().n == 34
    c.close()


def test_fastq_like_input_is_rejected_emu(emu_factory):
    fastq_like_is_rejected(emu_factory)


@pytest.mark.gpu
def test_fasta_ingest_gpu(gpu_ctx_factory):
    run_cases(gpu_ctx_factory, 8)
    fastq_like_is_rejected(gpu_ctx_factory)


@pytest.mark.gpu
def test_file_ingest_gpu(gpu_ctx_factory, tmp_path):
    for blk in (4096, 1 << 20, 0):
        check_file_ingest(gpu_ctx_factory, str(tmp_path), blk)


@pytest.mark.gpu
def test_fasta_ingest_large_gpu(gpu_ctx_factory):
    """40 Mbase in 60-column lines, several 1 MiB+ device chunks, pinned and pageable source"""
    import bench
    seqs = bench.synth_seqs(5_000_000, 8, 77, (0, 0, 0, 0))
    raw = bytearray()
    for h, s in enumerate(seqs):
        raw += b">hap%d\n" % h
        raw += np.concatenate([s.reshape(-1, 50), np.full((s.size // 50, 1), 10, np.uint8)], axis=1).tobytes()
    raw = bytes(raw)
    ref = oracle_run([bytes(s) for s in seqs], w=10, p=100, U=8)
    for chunk in (0, 1 << 22):
        c = gpu_ctx_factory(w=10, p=100, u64=True, sai=True)
        if chunk:
            c.debug_set(fasta_chunk_bytes=chunk)
        c.reserve(len(raw)); c.feed_fasta(raw, final=True)
        sz = c.finalize(); c.parse_bwt(); b = c.bwt_build(sa=False, rssa=True)
        res = {"n": sz.n, "r": b.r}; res.update(c.bwt_get())
        assert compare(res, ref, 8, names=("bwt", "ssa", "esa")) == [] and res["r"] == ref["r"]
        # .bwt from its run-length form into a page-locked buffer: host threads write runs from the front, the copy engine moves rows
        # from the back (1 MiB blocks here: 40 blocks) -- and into a pageable one (threads only)
        for pinned in (True, False):
            hb = np.full(sz.n + 1, 0xEE, np.uint8)
            if pinned: assert c.L.pfp_host_register(hb.ctypes.data, hb.size) == 0
            c.debug_set(expand_dma=2)
            c.bwt_get_expanded(hb.ctypes.data, res["ssa"], threads=4)
            if pinned: c.L.pfp_host_unregister(hb.ctypes.data)
            assert np.array_equal(hb, res["bwt"]), pinned
        c.close()
