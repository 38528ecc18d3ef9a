#!/usr/bin/env python3
"""bench.py -- BWT+SA build throughput of the MI355X engine (metric of BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its N ranks itself, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one complete build over one synthetic FASTA-equivalent text that is already resident in HBM: parse (trigger
scan, phrase de-duplication, dictionary suffix sort, ranks) -> BWT of the parse -> BWT (+ SA or run samples) emission,
outputs left in HBM.  `value` = bases / step time (device-resident; the host -> host figure is the `end_to_end` block).

Default workload = the headline of SURVEY.md 8(d) / BASELINE.json's north star: S-32G, 1000 synthetic haplotypes x
32 Mbase = 32 Gbase, seed 1000, -w 10 -p 100 -r (BWT + run-length sampled SA; a full .sa of 32 G x 8 B does not fit one
GPU), pfbwt-f64 path (64-bit uint_t).  N > 1 is STRONG scaling of that fixed collection: rank r holds haplotypes
[r*1000/N, (r+1)*1000/N); every rank parses its shard, ONE RCCL all-gather moves the per-rank dictionaries and parses to
every rank, every rank merges them and sorts the merged dictionary and the parse, and emits its own slice of the BWT rows
and of the run samples, which stay distributed in HBM.
--workload S-chr22 (configs[1] of BASELINE.json: one chromosome, L = 50 818 468, two N-runs, -s, 32-bit mode; at N > 1
weak scaling, haplotype r of the chromosome per rank), S-50M and S-5M are the smaller single-GPU cases.

Rank 0 prints ONE JSON line: the contract fields, `roofline` (dominant kernel, HIP-event timed inside the timed region),
`end_to_end` (N = 1: text in pinned host memory -> .bwt/.sa|.ssa/.esa in pinned host memory, PCIe included),
`cpu_baseline` (oracle/pfbwt_oracle pinned to one core with taskset; S-32G: the first 10 haplotypes) and `parity`:
sha256 of the engine's files == the oracle's on that sample, the engine run in a CHILD process with
PFP_FORCE_WIDE_ROWS=1 and a small PFP_EMIT_CHUNK_ROWS so that the instantiations that are timed at 32 Gbase (64-bit row
counters, windowed emission) are the ones that are checked.  A parity mismatch makes the process exit non-zero.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))

WORKLOADS = {
    # name: (L, H, seed, (r0, l0, r1, l1), w, p, u64)
    "S-chr22": (50_818_468, 1, 22, (10_000_000, 10_000_000, 35_000_000, 1_000_000), 10, 100, False),
    "S-50M": (5_000_000, 10, 12345, (0, 0, 0, 0), 10, 100, False),
    "S-5M": (5_000_000, 1, 22, (1_000_000, 500_000, 3_000_000, 50_000), 10, 100, False),
    # small collections (measurements of the text de-duplication: 8.6 % / 3 % of their phrases are distinct, 0.5 % of S-32G's)
    "S-20x32M": (32_000_000, 20, 1000, (0, 0, 0, 0), 10, 100, True),
    "S-100x32M": (32_000_000, 100, 1000, (0, 0, 0, 0), 10, 100, True),
    # the north-star shape: 1000 haplotypes x 32 Mbase = 32 Gbase on ONE GPU, -r (BWT + run-length sampled SA, a full
    # .sa of 32 G x 8 B does not fit), pfbwt-f64 path.
    "S-32G": (32_000_000, 1000, 1000, (0, 0, 0, 0), 10, 100, True),
    # configs[3] size on one GPU: 1000 haplotypes of chr22 length, -r
    "S-50G": (50_818_468, 1000, 1000, (0, 0, 0, 0), 10, 100, True),
    # configs[2] stand-in: one GRCh38-sized sequence (3.1 Gbase, 35 Mbp of N in two runs), -s -r, pfbwt-f64 path
    "S-3G": (3_100_000_000, 1, 38, (500_000_000, 30_000_000, 2_000_000_000, 5_000_000), 10, 100, True),
}
OUTPUTS = {"S-32G": (False, True), "S-50G": (False, True), "S-3G": (True, True), "S-20x32M": (False, True), "S-100x32M": (False, True)}      # (-s, -r); default: -s only
CPU_SAMPLE_HAPLOTYPES = 10        # large collections: the CPU baseline / oracle parity run covers the first 10 haplotypes
CPU_SAMPLE_BASES = 100_000_000    # one huge sequence: its first 100 Mbase (~20 s of CPU work)


def outputs_of(workload):
    return OUTPUTS.get(workload, (True, False))


def out_names(want_sa, want_rssa):
    return ("bwt",) + (("sa",) if want_sa else ()) + (("ssa", "esa") if want_rssa else ())


def oracle_mode(want_sa, want_rssa):
    return ["-s"] * want_sa + ["-r"] * want_rssa
PARITY_ENV = {"PFP_TEST_HOOKS": "1", "PFP_FORCE_WIDE_ROWS": "1", "PFP_EMIT_CHUNK_ROWS": str(1 << 25)}     # S-32G parity child: k_*<u64, u64>, ~10 windows

KERNEL_SYMBOL = {"emit": "pfp::k_emit", "emit_large": "pfp::k_emit_groups_large<", "emit_big": "pfp::k_emit<", "fill": "pfp::k_fill<", "radix_scatter": "pfp::k_seg_scatter<unsigned long", "radix_hist": "pfp::k_seg_hist<unsigned long",
                 "class_sort": "pfp::k_round<", "emit_count": "pfp::k_emit_slots<", "samples": "pfp::k_sample_values<", "phrase_hash": "pfp::k_dedup_insert",
                 "trigger_scan": "pfp::k_trigger_scan", "ss_write_rank": "pfp::k_round_apply<"}


def synth_lib():
    lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
    lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
    return lib


def synth_seqs(L, H, seed, nruns, h0=0, out=None):
    """haplotypes [h0, h0 + H) of the synthetic collection as rows of one (H, L) uint8 array (`out`, e.g. pinned memory)"""
    lib = synth_lib()
    big = out if out is not None else np.empty((H, L), np.uint8)
    if H > 16:      # big collections: generated by a thread pool (the C generator releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            list(ex.map(lambda h: lib.pfp_synth_haplotype(seed, L, h0 + h, *nruns, big[h].ctypes.data_as(C.c_void_p)), range(H)))
    else:
        for h in range(H):
            lib.pfp_synth_haplotype(seed, L, h0 + h, *nruns, big[h].ctypes.data_as(C.c_void_p))
    return [big[h] for h in range(H)]


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def cpu_baseline(seqs, w, p, u64, mode=("-s",)):
    """Time oracle/pfbwt_oracle (CPU restatement, one thread, pinned to one core) on the input; returns (dict, digests)."""
    exe = os.path.join(ROOT, "oracle", "pfbwt_oracle")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="pfbwt_bench_")
    fa = os.path.join(tmp, "in.fa")
    with open(fa, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">hap%d\n" % i)
            for k in range(0, s.size, 60000):
                f.write(s[k:k + 60000].tobytes()); f.write(b"\n")
    n = sum(int(s.size) + w for s in seqs)
    mode = list(mode)
    cmd = [exe] + mode + ["--u64" if u64 else "--u32", "-w", str(w), "-p", str(p), "-o", os.path.join(tmp, "out"), fa]
    core = None
    try:        # the reference is single-threaded: one host core of the GPU box
        core = sorted(os.sched_getaffinity(0))[-1]
        if subprocess.run(["taskset", "-c", str(core), "true"], capture_output=True).returncode == 0:
            cmd = ["taskset", "-c", str(core)] + cmd
        else:
            core = None
    except (OSError, AttributeError):
        core = None
    t0 = time.time()
    pr = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    if pr.returncode != 0:
        raise RuntimeError("oracle failed: " + pr.stderr[-400:])
    stages = {}
    for line in pr.stderr.splitlines():
        if line.startswith("TASK\t"):
            _, name, sec = line.split("\t")
            stages[name] = float(sec.rstrip("s"))
    compute = sum(v for k, v in stages.items() if k != "reading input")
    # the pieces of the REFERENCE that build here from its own sources (oracle/Makefile ref; src/pfbwt-f.cpp itself needs sdsl-lite):
    # merge_pfp64 --parse-bwt = PfParser::add_fasta + finalize + bwt_of_parse (src/merge_pfp.cpp:115-170), gsacak = the dictionary
    # suffix sort + LCP of pfbwt.hpp:211 -- timed on the same sample, pinned to the same core, beside the port's stages
    ref_pieces = None
    mp = os.path.join(ROOT, "oracle", "_ref", "merge_pfp64"); gs = os.path.join(ROOT, "oracle", "_ref", "libgsacak64.so")
    if os.path.exists(mp) and os.path.exists(gs):
        try:
            pin = ["taskset", "-c", str(core)] if core is not None else []
            t0 = time.time()
            pr2 = subprocess.run(pin + [mp, "-w", str(w), "-p", str(p), "-s", "--parse-bwt", "-o", os.path.join(tmp, "ref"), fa], capture_output=True, text=True)
            t_parse = time.time() - t0
            if pr2.returncode != 0:
                raise RuntimeError(pr2.stderr[-300:])
            code = ("import ctypes,sys,time,numpy as np\n"
                    "L=ctypes.CDLL(sys.argv[1]); d=np.fromfile(sys.argv[2],np.uint8); n=d.size\n"
                    "SA=np.empty(n,np.uint64); LCP=np.empty(n,np.int64)\n"
                    "L.gsacak.argtypes=[ctypes.c_void_p]*4+[ctypes.c_uint64]; t=time.time()\n"
                    "L.gsacak(d.ctypes.data,SA.ctypes.data,LCP.ctypes.data,None,n); print(n, time.time()-t)\n")
            pr3 = subprocess.run(pin + [sys.executable, "-c", code, gs, os.path.join(tmp, "ref.dict")], capture_output=True, text=True)
            if pr3.returncode != 0:
                raise RuntimeError(pr3.stderr[-300:])
            dn, t_gsa = pr3.stdout.split()
            port_parse = sum(v for k, v in stages.items() if k.startswith(("parsing", "writing dict", "ranking")))
            ref_pieces = {"merge_pfp64 --parse-bwt (reference parse + parse-BWT, FASTA read and file writes included) s": round(t_parse, 2),
                          "port, same stages s": round(port_parse + stages.get("reading input", 0.0), 2),
                          "gsacak (reference dictionary gSA + LCP) s": round(float(t_gsa), 2), "dictionary bytes": int(dn),
                          "parse Mbases/s reference": round(n / t_parse / 1e6, 1), "gsacak Mchars/s": round(int(dn) / float(t_gsa) / 1e6, 2)}
        except Exception as e:
            ref_pieces = {"error": repr(e)}
    dig = {ext: sha_file(os.path.join(tmp, "out." + ext)) for ext in out_names("-s" in mode, "-r" in mode)}
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    os.rmdir(tmp)
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return ({"value": n / compute / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port", "reference_pieces_same_sample": ref_pieces,
             "sample": "n=%d bases through oracle/pfbwt_oracle %s, one thread%s; stage seconds %s; wall incl. FASTA read + file writes %.1f s; host has %d cores (%s)"
                       % (n, " ".join(mode), (" pinned with taskset -c %d" % core) if core is not None else "", json.dumps(stages), wall, os.cpu_count(), cpu_model)}, dig)


def parity_child(a):
    """child process of the parity leg: the engine on the CPU sample of the workload (its first `--parity-child` haplotypes, cut to
    `--parity-bases` bases each), under the environment the parent chose; prints the sha256 of its output files as one JSON line"""
    import pfbwt_hip
    L, H, seed, nruns, w, p, u64 = WORKLOADS[a.workload]
    want_sa, want_rssa = outputs_of(a.workload)
    sub = [t[:a.parity_bases] if a.parity_bases else t for t in synth_seqs(L, a.parity_child, seed, nruns)]
    c = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=0)
    for t in sub:
        c.feed(t, True)
    c.finalize(); c.parse_bwt(); c.bwt_build(sa=want_sa, rssa=want_rssa)
    o = c.bwt_get(); c.close()
    print(json.dumps({k: hashlib.sha256(o[k].tobytes()).hexdigest() for k in out_names(want_sa, want_rssa)}), flush=True)


def write_fasta_image(path, rows, line=60000):
    """the collection as a FASTA file (one record per haplotype, >hap<h>, 60 000-character lines), written by a
    thread pool with pwrite (the target is tmpfs: the image is memory resident, like page-cached input)"""
    from concurrent.futures import ThreadPoolExecutor
    L = int(rows[0].size)
    nfull, rem = divmod(L, line)
    body = nfull * (line + 1) + ((rem + 1) if rem else 0)
    heads = [b">hap%d\n" % h for h in range(len(rows))]
    offs = np.concatenate([[0], np.cumsum([len(hd) + body for hd in heads])]).astype(np.int64)
    fd = os.open(path, os.O_CREAT | os.O_TRUNC | os.O_WRONLY, 0o600)
    os.ftruncate(fd, int(offs[-1]))

    def one(h):
        buf = np.empty(len(heads[h]) + body, np.uint8)
        buf[:len(heads[h])] = np.frombuffer(heads[h], np.uint8)
        b = buf[len(heads[h]):]
        if nfull:
            v = b[:nfull * (line + 1)].reshape(nfull, line + 1)
            v[:, :line] = rows[h][:nfull * line].reshape(nfull, line); v[:, line] = 10
        if rem:
            b[nfull * (line + 1):-1] = rows[h][nfull * line:]; b[-1] = 10
        mv, o = memoryview(buf), int(offs[h])
        while len(mv):
            k = os.pwrite(fd, mv[:1 << 30], o); mv = mv[k:]; o += k
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(len(rows))))
    os.close(fd)
    return int(offs[-1])


def host_wordsum(buf, threads=16):
    """sum of the 64-bit words of a host buffer and of word * (index + 1), modulo 2^64 (= pfp_debug_wordsum on the device)"""
    from concurrent.futures import ThreadPoolExecutor
    nb = buf.size
    nw = nb // 8
    wv = buf[:nw * 8].view(np.uint64)
    step = 1 << 24

    def part(i0):
        x = wv[i0:i0 + step]
        idx = np.arange(i0 + 1, i0 + 1 + x.size, dtype=np.uint64)
        return int(x.sum(dtype=np.uint64)), int((x * idx).sum(dtype=np.uint64))
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(part, range(0, nw, step)))
    a = sum(p[0] for p in parts); b = sum(p[1] for p in parts)
    if nb % 8:
        v = int.from_bytes(buf[nw * 8:].tobytes(), "little"); a += v; b += v * (nw + 1)
    return [a & (2 ** 64 - 1), b & (2 ** 64 - 1)]


def e2e_child(a):
    """`--e2e-child IMAGE`: a FRESH process builds the index of the FASTA image -- through the engine's own file reader, the
    same call the command line makes (pfp_parse_feed_fasta_file) -- into host memory, twice: run 0 is the cold figure (clock
    starts before the library is loaded; the output buffers are faulted in and page-locked by helper threads while the file
    is read), run 1 the warm one (context, committed HBM and page-locked buffers kept).  Prints one JSON line."""
    t_start = time.perf_counter()
    import threading
    import pfbwt_hip
    L, H, seed, nruns, w, p, u64 = WORKLOADS[a.workload]
    U = 8 if u64 else 4
    want_sa, want_rssa = outputs_of(a.workload)
    lib = pfbwt_hip.load_library()
    fsize = os.path.getsize(a.e2e_child)
    outs = {}

    def prepare_outputs():      # n + 1 <= file size: buffers for the worst case, faulted in by 16 threads, then page-locked
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        bufs = {"bwt": np.empty(fsize + 64, np.uint8)}
        if want_sa:
            bufs["sa"] = np.empty((fsize + 64) * U, np.uint8)
        if want_rssa:      # r is not known in advance: page-locked room for n / 128 runs of a collection (4 * U bytes each; S-32G has n / 383), pageable arrays if there are more
            bufs["samples"] = np.empty(((fsize // 128 if H > 1 else fsize * 4 // 5) + 64) * 4 * U, np.uint8)      # (one non-repetitive sequence: r is of the order of n)
        step = 1 << 28
        with ThreadPoolExecutor(max_workers=6) as ex:      # few threads: the box gives a one-GPU job 16 cores, and the file reader's 8 threads must not be starved
            for b in bufs.values():
                list(ex.map(lambda i, b=b: b[i:i + step].fill(0), range(0, b.size, step)))
        t1 = time.perf_counter()
        for b in bufs.values():
            if lib.pfp_host_register(b.ctypes.data, b.size) != 0:
                raise RuntimeError("pfp_host_register failed")
        outs.update(bufs); outs["fault_ms"] = 1e3 * (t1 - t0); outs["register_ms"] = 1e3 * (time.perf_counter() - t1)

    th = threading.Thread(target=prepare_outputs); th.start()
    ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=0)
    runs = []
    rle = want_rssa and not want_sa      # -r: the .bwt crosses PCIe as one byte per run and is written out by host threads (pfp_bwt_get_expanded)
    for it in range(3 if rle else 2):
        expanded = rle and it < 2          # run 2 (warm): every row over PCIe while the emission runs (pfp_bwt_build_stream), for comparison
        t0 = t_start if it == 0 else time.perf_counter()
        t_in0 = time.perf_counter()
        info = ctx.feed_fasta_file(a.e2e_child)
        t1 = time.perf_counter()
        sz = ctx.finalize(); ctx.parse_bwt()
        t2 = time.perf_counter()
        th.join()
        t3 = time.perf_counter()
        if expanded:
            b = ctx.bwt_build(sa=False, rssa=True)
        else:
            b = ctx.bwt_build_stream(outs["bwt"].ctypes.data, outs["sa"].ctypes.data if want_sa else None, rssa=want_rssa)
        t4 = time.perf_counter()
        ssa = esa = None
        if want_rssa:
            if 4 * U * b.r <= outs["samples"].size:
                sv = outs["samples"].view(np.uint64 if u64 else np.uint32)
                ssa, esa = ctx.samples_get(out={"ssa": sv[:2 * b.r], "esa": sv[2 * b.r:4 * b.r]})
            else:
                ssa, esa = ctx.samples_get()
        t5 = time.perf_counter()
        if expanded:
            ctx.bwt_get_expanded(outs["bwt"].ctypes.data, ssa, threads=0)      # 0: one writer per CPU of this process
        t6 = time.perf_counter()
        n = sz.n
        if expanded and it == 1:      # checked apart from the streamed copy that overwrites the buffer in run 2
            exp_sum = host_wordsum(outs["bwt"][:n + 1])
        runs.append({"ms": 1e3 * (t6 - t0), "startup_ms": 1e3 * (t_in0 - t0), "ingest_ms": 1e3 * (t1 - t_in0), "reader_GBps": info.raw_bytes / (t1 - t_in0) / 1e9,
                     "reader_mode": info.mode, "reader_wait_ms": info.read_wait_ms, "parse_ms": 1e3 * (t2 - t1), "wait_for_output_buffers_ms": 1e3 * (t3 - t2),
                     ("emit_ms" if expanded else "emit_and_download_ms"): 1e3 * (t4 - t3), "samples_download_ms": 1e3 * (t5 - t4),
                     "bwt_from_runs_ms": 1e3 * (t6 - t5) if expanded else None, "bwt_path": ("one byte per run over PCIe + %d host threads writing runs from the front while the copy engine moves rows from the back (pfp_bwt_get_expanded)" % len(os.sched_getaffinity(0))) if expanded else "every row over PCIe, overlapped with the emission (pfp_bwt_build_stream)",
                     "value": n / (t6 - t0) / 1e9, "unit": "Gbases/s", "n": int(n), "r": int(b.r), "raw_bytes": int(info.raw_bytes), "records": int(info.records)})
    res = {"cold": runs[0], "warm": runs[1], "output_buffers": {"fault_ms": outs["fault_ms"], "register_ms": outs["register_ms"]}}
    if rle:
        res["warm_streamed"] = runs[2]; res["sum_bwt_from_runs"] = exp_sum
    # what reached host memory: word sums of .bwt (and .sa), compared by the parent with the device-resident outputs of the timed steps
    n = runs[1]["n"]
    res["sums"] = {"bwt": host_wordsum(outs["bwt"][:n + 1])}
    if want_rssa:
        res["sums"]["ssa"] = host_wordsum(ssa.view(np.uint8)); res["sums"]["esa"] = host_wordsum(esa.view(np.uint8))
    if want_sa:
        res["sums"]["sa"] = host_wordsum(outs["sa"][:(n + 1) * U])
    t6 = time.perf_counter()
    ctx.close()
    res["teardown_ms"] = 1e3 * (time.perf_counter() - t6)
    print(json.dumps(res), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` invoked plainly: start the N ranks as fresh child processes (this process has not
    touched the GPU) and exit with their status; rank 0's JSON line goes to the inherited stdout.  The ranks meet through a
    file store in a fresh temporary directory (no TCP port to race for with another bench on the same box)."""
    d = tempfile.mkdtemp(prefix="pfbwt_bench_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), PFP_BENCH_INIT_FILE=os.path.join(d, "store"), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    try:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    except OSError:
        pass
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="S-32G", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--parity-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--parity-bases", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--e2e-child", default="", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.parity_child:
        return parity_child(a)
    if a.e2e_child:
        return e2e_child(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(a.gpus)

    # stdout carries exactly ONE line, the JSON record: RCCL (version banner, NCCL WARN lines) and any other library that
    # writes to file descriptor 1 is sent to stderr from here on
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import pfbwt_hip
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    torch.cuda.set_device(lrank)
    dist = None
    forced = os.environ.get("PFP_BENCH_FORCE_SHARDED") == "1"     # rehearsal of the N > 1 code path on one GPU (world size 1 group)
    if world > 1 or forced:
        import torch.distributed as dist
        if forced and "RANK" not in os.environ:      # the rehearsal started plainly: a group of one rank
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", str(29000 + os.getpid() % 2000))})
        if os.environ.get("PFP_BENCH_INIT_FILE"):      # started by spawn_ranks
            dist.init_process_group("nccl", init_method="file://" + os.environ["PFP_BENCH_INIT_FILE"], rank=rank, world_size=world, device_id=torch.device("cuda", lrank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", lrank))
    pfbwt_hip.load_library()  # raises if the gfx950 library is absent

    L, H, seed, nruns, w, p, u64 = WORKLOADS[a.workload]
    U = 8 if u64 else 4
    want_sa, want_rssa = outputs_of(a.workload)
    big = n_is_big = H * (L + w) > (1 << 31)      # the CPU leg and the oracle parity run cover a sample of such workloads
    strong = a.workload in ("S-32G", "S-50G")      # the north-star collection is FIXED (1000 haplotypes); N GPUs share it
    if strong:
        if H % world:
            raise SystemExit("%s: %d haplotypes do not split evenly over %d ranks" % (a.workload, H, world))
        H //= world
    # rank r holds haplotypes [r*H, (r+1)*H) of the same synthetic collection (S-32G: H = 1000 / N, the total is fixed =
    # strong scaling; other workloads: fixed bases per GPU = weak scaling).  The host copy lives in page-locked memory.
    h_all = torch.empty((H, L), dtype=torch.uint8, pin_memory=True)
    seqs = synth_seqs(L, H, seed, nruns, h0=rank * H, out=h_all.numpy())
    d_all = h_all.to("cuda", non_blocking=False)
    n_local = H * (L + w)
    n = n_local * world
    # the workspace is sized and allocated ONCE, before the timed region (a cold hipMalloc of the slab takes seconds)
    ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=lrank)
    dev = torch.device("cuda", lrank)

    def feed_local(c):
        # the haplotype rows are resident in HBM (d_all) and stay there for the whole run: the engine reads them in place (the trigger
        # scan of finalize writes the text); a shard with left context is fed by copy (a view must be a parse's whole text)
        if world == 1 and not forced:
            c.feed_device_view(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))
        else:
            c.feed_device_batch(d_all.data_ptr(), d_all.shape[0], d_all.shape[1], d_all.stride(0))

    def step():
        if world == 1 and not forced:
            feed_local(ctx)
            ctx.finalize(); ctx.parse_bwt()
            return ctx.bwt_build(sa=want_sa, rssa=want_rssa)
        import pfbwt_dist
        return pfbwt_dist.sharded_build(ctx, feed_local, w, dev, sa=want_sa, rssa=want_rssa)[1]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup; the first warmup step is profiled kernel-by-kernel to find the dominant kernel
    dominant = None
    for i in range(max(a.warmup, 1)):
        if i == 0:
            ctx.profile_enable(True); ctx.profile_reset()
        b = step()
        if i == 0:
            rows = ctx.profile(); ctx.profile_enable(False)
            dominant = max((r for r in rows if r["kernel"] != "misc"), key=lambda r: r["ms"])["kernel"]
            warm_rows = rows
            if dist is not None:   # every rank times the kernel that dominates on rank 0
                obj = [dominant]; dist.broadcast_object_list(obj, src=0); dominant = obj[0]
    ctx.profile_select(dominant); ctx.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        b = step()
    sync()
    dt = time.perf_counter() - t0
    prof = [r for r in ctx.profile() if r["kernel"] == dominant]
    ctx.profile_enable(False)
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rt = torch.tensor([int(b.r)], dtype=torch.int64, device="cuda")   # runs that start in this rank's slice
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    dt = float(tt.item()); r_total = int(rt.item())
    rc = 0
    if rank == 0:
        ms_per_step = 1e3 * dt / a.steps
        value = n / (dt / a.steps) / 1e9
        pr = prof[0]
        ach = pr["bytes"] / pr["launches"] / (pr["ms"] / pr["launches"] * 1e-3) / 1e9
        total_ms = sum(r["ms"] for r in warm_rows)
        traffic = None
        try:   # HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/)
            per_wl = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % a.workload)      # per-workload passes (tools/gpu_final.sh); pmc_traffic_latest.json = the default workload's
            pt = json.load(open(per_wl if os.path.exists(per_wl) else os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            prefix = KERNEL_SYMBOL.get(dominant, "pfp::k_" + dominant)
            cand = [rec for kname, rec in pt.items() if prefix in kname and pt.get("_workload") == a.workload]
            if cand:       # FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE (tools/pmc_traffic.py keeps the raw sum too);
                # a kernel id can cover several instantiations (k_round<DICT, K>): launch-weighted mean, like `achieved`
                traffic = sum(r["launches"] * r["hbm_bytes_per_launch_corrected"] for r in cand) / sum(r["launches"] for r in cand)
        except (OSError, ValueError, KeyError):
            pass
        out_list = ", ".join("." + k for k in out_names(want_sa, want_rssa))
        res = {
            "metric": "Gbases/s end-to-end BWT+SA build; bit-exact .bwt/.sa vs reference", "value": value,
            "value_is": "device-resident build (text already in HBM, outputs left in HBM), as the bench contract prescribes; the FASTA-bytes -> host-memory figures of the metric's wording are the end_to_end_fasta block (cold / warm), host rows -> host is end_to_end", "unit": "Gbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8 text / u32 indices / u64 hash", "data": "synthetic",
            "config": {"workload": a.workload, "L": L, "H": H * world, "seed": seed, "n_runs": list(nruns), "w": w, "p": p, "flags": " ".join(oracle_mode(want_sa, want_rssa)), "uint_t": 64 if u64 else 32,
                       "n": n, "r": r_total, "input": "text resident in HBM, outputs (%s) left in HBM" % out_list,
                       "per_rank": ("%d haplotype(s) of the collection per rank; parse sharded, one RCCL all-gather of dictionaries + parses, "
                                    "merge + dictionary/parse suffix sorts on every rank, emission (and run samples) sliced over the ranks" % H) if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                         "launches_per_step": pr["launches"] / a.steps, "avg_launch_us": 1e3 * pr["ms"] / pr["launches"],
                         "alg_bytes_per_launch": pr["bytes"] / pr["launches"],
                         "share_of_kernel_time": max(r["ms"] for r in warm_rows if r["kernel"] != "misc") / total_ms,
                         "end_to_end_alg_GBps": (2 * n + (4 * U * r_total if want_rssa else 0) + (U * n if want_sa else 0)) / (dt / a.steps) / 1e9},
            "stage_ms": ctx.stage_ms(),
        }
        dev_sums = None
        if world == 1 and not forced and want_rssa:
            # full-size ordering proof on the outputs of the last timed step: all r - 1 pairs of adjacent rows (.esa[k], .ssa[k + 1])
            # compared suffix against suffix on the resident text (pfp_debug_check_sample_order)
            res["full_size_order_check"] = ctx.check_sample_order()
            if res["full_size_order_check"]["order_violations"] or res["full_size_order_check"]["rows_not_adjacent"]:
                rc = 3
        if world == 1 and not forced and not a.no_end_to_end:
            # word sums of the device-resident outputs of the last timed step: what the end-to-end legs deliver to host memory is
            # compared with them (include/pfbwt_hip_dev.h: pfp_debug_wordsum)
            import ctypes
            dp = dict(zip(("bwt", "sa", "ssa", "esa"), ctx.bwt_device_ptrs()))
            def dsum(ptr, nbytes):
                o = (ctypes.c_uint64 * 2)(); ctx._check(ctx.L.pfp_debug_wordsum(ctx.h, ctypes.c_void_p(ptr), int(nbytes), o)); return [int(o[0]), int(o[1])]
            dev_sums = {"bwt": dsum(dp["bwt"], n + 1)}
            if want_rssa:
                dev_sums["ssa"] = dsum(dp["ssa"], 2 * r_total * U); dev_sums["esa"] = dsum(dp["esa"], 2 * r_total * U)
            if want_sa:
                dev_sums["sa"] = dsum(dp["sa"], (n + 1) * U)
            # host -> host, rows: the text sits in pinned host memory as headerless rows, the outputs end in pinned host memory
            # page-locked output buffers of the exact size (torch's pinned allocator rounds to powers of two and keeps what it freed: on S-3G,
            # r = 0.74 n, that alone was 160 GB of host memory next to the child process of the FASTA leg)
            def pinned(nbytes):
                arr = np.empty(nbytes, np.uint8); arr.fill(0)
                if ctx.L.pfp_host_register(arr.ctypes.data, arr.size) != 0:
                    raise RuntimeError("pfp_host_register failed")
                return arr
            ob_raw = {"bwt": pinned(n + 1)}
            if want_rssa:
                ob_raw["ssa"] = pinned(2 * r_total * U); ob_raw["esa"] = pinned(2 * r_total * U)
            if want_sa:
                ob_raw["sa"] = pinned((n + 1) * U)
            ob = {k: (v if k == "bwt" else v.view(np.uint64 if u64 else np.uint32)) for k, v in ob_raw.items()}
            best = None
            for _ in range(2):
                torch.cuda.synchronize(); e0 = time.perf_counter()
                ctx.feed_host_batch(h_all.data_ptr(), H, L, h_all.stride(0))
                e1 = time.perf_counter()
                ctx.finalize(); ctx.parse_bwt()
                e2 = time.perf_counter()
                ctx.bwt_build_stream(ob["bwt"].ctypes.data, ob["sa"].ctypes.data if want_sa else None, rssa=want_rssa)
                if want_rssa:
                    ctx.samples_get(out=ob)
                e3 = time.perf_counter()
                cur = {"ms": 1e3 * (e3 - e0), "h2d_ms": 1e3 * (e1 - e0), "parse_ms": 1e3 * (e2 - e1), "emit_and_d2h_ms": 1e3 * (e3 - e2)}
                if best is None or cur["ms"] < best["ms"]:
                    best = cur
            in_b = n_local; out_b = (n + 1) + (4 * U * r_total if want_rssa else 0) + (U * (n + 1) if want_sa else 0)
            best.update({"value": n / (best["ms"] * 1e-3) / 1e9, "unit": "Gbases/s", "h2d_GBps": in_b / best["h2d_ms"] / 1e6, "d2h_GBps": out_b / best["emit_and_d2h_ms"] / 1e6,
                         "what": "headerless rows in page-locked host memory -> one DMA transfer per row -> parse -> emission with every finished window of rows "
                                 "on its way to page-locked host memory while the next one is emitted (pfp_bwt_build_stream) -> run samples",
                         "pcie_cap_Gbases_per_s": n / ((in_b + out_b) / 57e9) / 1e9,
                         "pcie_cap_note": "57 GB/s per direction measured on this box (profiles/r03a_alloc_bench.log; both directions at once: 25 GB/s each, so upload and download "
                                          "are kept apart); the outputs depend on the whole text, so upload + parse + download are serial"})
            best["outputs_match_device"] = {k: host_wordsum(ob_raw[k]) == v for k, v in dev_sums.items()}
            res["end_to_end"] = best
            for v in ob_raw.values():
                ctx.L.pfp_host_unregister(v.ctypes.data)
            del ob, ob_raw
        ctx.close(); del d_all
        torch.cuda.empty_cache()
        if dev_sums is not None:
            # FASTA bytes -> outputs in host memory, in a FRESH process (cold, then warm) through the engine's file reader
            try:
                img = None
                need = int(H * (L + L // 60000 + 16))
                for d in ("/dev/shm", tempfile.gettempdir()):
                    st = os.statvfs(d)
                    if st.f_bavail * st.f_frsize > need + (1 << 30):
                        img = os.path.join(d, "pfbwt_bench_%d.fa" % os.getpid()); break
                if img is None:
                    raise RuntimeError("no room for a %d-byte FASTA image" % need)
                t0 = time.perf_counter()
                fbytes = write_fasta_image(img, seqs)
                t_img = time.perf_counter() - t0
                if big:      # the CPU leg's sample outlives the page-locked rows
                    seqs = [np.array(s[:CPU_SAMPLE_BASES]) for s in seqs[:CPU_SAMPLE_HAPLOTYPES]]
                del h_all
                # this process has just released its HBM: the driver wipes freed VRAM in the background (~40 GB/s) and a large
                # allocation waits for a pending wipe (profiles/r03a_alloc_fresh.log) -- "cold" means a fresh process on an idle card
                time.sleep(8.0 if n > (1 << 30) else 2.0)      # (up to ~250 GB were released)
                t0 = time.perf_counter()
                pc = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", a.workload, "--e2e-child", img], capture_output=True, text=True)
                wall = time.perf_counter() - t0
                os.remove(img)
                if pc.returncode != 0:
                    raise RuntimeError("child failed: " + pc.stderr[-600:])
                ej = json.loads(pc.stdout.strip().splitlines()[-1])
                ej["match_device_outputs"] = {k: ej["sums"][k] == v for k, v in dev_sums.items()}
                if "sum_bwt_from_runs" in ej:
                    ej["match_device_outputs"]["bwt_from_runs"] = ej.pop("sum_bwt_from_runs") == dev_sums["bwt"]
                del ej["sums"]
                ej["child_process_wall_ms"] = 1e3 * wall
                ej["image"] = {"bytes": fbytes, "where": os.path.dirname(img), "written_in_s": t_img, "format": "one record per haplotype, >hap<h>, 60 000-character lines"}
                ej["what"] = ("FASTA image (memory-resident file) -> pfp_parse_feed_fasta_file (the call PfParser::add_fasta makes: parallel pread into page-locked blocks, "
                              "upload, header / newline stripping on the device) -> parse -> emission -> run samples + .bwt to page-locked host memory (-r: the .bwt as one byte per run, written out by 16 host "
                              "threads; warm_streamed: every row over PCIe while the emission runs).  cold: clock starts "
                              "before the library is loaded in a fresh process on an idle card; warm: second build in that process")
                res["end_to_end_fasta"] = ej
                if not all(ej["match_device_outputs"].values()) or not all(res["end_to_end"]["outputs_match_device"].values()):
                    rc = 3
            except Exception as e:      # the headline does not depend on this leg; its absence is visible in the record
                res["end_to_end_fasta"] = {"error": repr(e)}
        if not a.no_cpu_baseline and world == 1:
            # bounded CPU sample + parity: the oracle on the sample, the engine on the same sample in a child process
            nh = min(CPU_SAMPLE_HAPLOTYPES, H) if big else H
            nbases = min(CPU_SAMPLE_BASES, L) if big else L
            sub = [s[:nbases] for s in seqs[:nh]]
            env = dict(os.environ)
            if big:
                env.update(PARITY_ENV)
            pc = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", a.workload, "--parity-child", str(nh)] + (["--parity-bases", str(nbases)] if nbases < L else []),
                                env=env, capture_output=True, text=True)
            cb, dig = cpu_baseline(sub, w, p, u64, mode=oracle_mode(want_sa, want_rssa))
            res["cpu_baseline"] = cb
            try:
                got = json.loads(pc.stdout.strip().splitlines()[-1]) if pc.returncode == 0 else {}
            except (ValueError, IndexError):
                got = {}
            ok = bool(got) and all(got.get(k) == v for k, v in dig.items())
            what = "sha256(%s) of the engine == oracle/pfbwt_oracle (the repo's CPU restatement, pinned to the reference's goldens)" % ", ".join("." + k for k in dig)
            scope = ("on a sample: the first %d sequence(s), %d bases each (%d bases), engine run with %s so that the 64-bit-row / windowed kernels timed above are the ones "
                     "checked; the full-size outputs are covered by full_size_order_check (all adjacent run-boundary rows, suffix against suffix) and by the word sums of the "
                     "end-to-end legs" % (nh, nbases, nh * (nbases + w), " ".join("%s=%s" % kv for kv in PARITY_ENV.items()))) if big else "on the whole input"
            res["parity"] = ("bit-exact: " if ok else "MISMATCH: ") + what + " " + scope
            if not ok:
                res["parity_detail"] = {"engine": got, "oracle": dig, "child_rc": pc.returncode, "child_stderr": pc.stderr[-500:]}
                rc = 3
        json_out.write(json.dumps(res) + "\n"); json_out.flush()
    else:
        ctx.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
