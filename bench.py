#!/usr/bin/env python3
"""bench.py -- end-to-end BWT+SA build throughput of the MI355X engine (metric of BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one complete build over one synthetic FASTA-equivalent text that is already resident in
HBM: parse (trigger scan, phrase de-duplication, dictionary suffix sort, ranks) -> BWT of the parse ->
BWT + full SA emission, outputs left in HBM.  Workload at N=1: S-chr22 (SURVEY.md 8(d), configs[1] of
BASELINE.json): one synthetic chromosome, L = 50 818 468, seed 22, two N-runs (10 Mbp + 1 Mbp),
-w 10 -p 100 -s, 32-bit mode.  N>1 (weak scaling): rank r holds haplotype r of the same synthetic chromosome
(S-chr22 shape per GPU); every rank parses its shard, ONE RCCL all-gather moves the per-rank dictionaries and
parses to every rank, every rank merges them and sorts the merged dictionary and the parse (identical, redundant work),
and emits its own slice of the BWT/SA rows, which stay distributed in HBM; value = total bases / step time.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel, HIP-event timed
inside the timed region) and `cpu_baseline` (oracle/pfbwt_oracle, single thread, same input).
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
}


def synth_seqs(L, H, seed, nruns, h0=0):
    lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
    lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
    out = []
    for h in range(h0, h0 + H):
        a = np.empty(L, np.uint8)
        lib.pfp_synth_haplotype(seed, L, h, *nruns, a.ctypes.data_as(C.c_void_p))
        out.append(a)
    return out


def cpu_baseline(seqs, w, p, u64, want_digest):
    """Time oracle/pfbwt_oracle (CPU restatement, one thread) on the same input; returns (dict, digests)."""
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
    t0 = time.time()
    pr = subprocess.run([exe, "-s", "--u64" if u64 else "--u32", "-w", str(w), "-p", str(p), "-o", os.path.join(tmp, "out"), fa],
                        capture_output=True, text=True)
    wall = time.time() - t0
    if pr.returncode != 0:
        raise RuntimeError("oracle failed: " + pr.stderr[-400:])
    stages = {}
    for line in pr.stderr.splitlines():
        if line.startswith("TASK\t"):
            _, name, sec = line.split("\t")
            stages[name] = float(sec.rstrip("s"))
    compute = sum(v for k, v in stages.items() if k != "reading input")
    dig = {}
    if want_digest:
        for ext in ("bwt", "sa"):
            h = hashlib.sha256()
            with open(os.path.join(tmp, "out." + ext), "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            dig[ext] = h.hexdigest()
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
    return ({"value": n / compute / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
             "sample": "the full workload (n=%d) through oracle/pfbwt_oracle -s, one thread; stage seconds %s; wall incl. FASTA read + file writes %.1f s; host has %d cores (%s)"
                       % (n, json.dumps(stages), wall, os.cpu_count(), cpu_model)}, dig)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="S-chr22", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import pfbwt_hip
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    torch.cuda.set_device(lrank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", lrank))
    pfbwt_hip.load_library()  # raises if the gfx950 library is absent

    L, H, seed, nruns, w, p, u64 = WORKLOADS[a.workload]
    # rank r holds haplotypes [r*H, (r+1)*H) of the same synthetic collection (weak scaling: fixed bases per GPU)
    seqs = synth_seqs(L, H, seed, nruns, h0=rank * H)
    d_seqs = [torch.from_numpy(s).to("cuda") for s in seqs]
    n_local = sum(int(s.size) + w for s in seqs)
    n = n_local * world
    ctx = pfbwt_hip.PfpContext(w=w, p=p, u64=u64, sai=True, device=lrank)
    dev = torch.device("cuda", lrank)

    def feed_local(c):
        for t in d_seqs:
            c.feed_device(t.data_ptr(), t.numel(), True)

    def step():
        if world == 1:
            feed_local(ctx)
            ctx.finalize(); ctx.parse_bwt()
            return ctx.bwt_build(sa=True, rssa=False)
        import pfbwt_dist
        return pfbwt_dist.sharded_build(ctx, feed_local, w, dev, sa=True)[1]

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
            dominant = max(rows, key=lambda r: r["ms"])["kernel"]
            warm_rows = rows
            if dist is not None:   # every rank times the kernel that dominates on rank 0 (it runs the single-GPU stages)
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

    if rank == 0:
        out = ctx.bwt_get() if (not a.no_cpu_baseline and world == 1) else None
        ms_per_step = 1e3 * dt / a.steps
        value = n / (dt / a.steps) / 1e9
        pr = prof[0]
        ach = pr["bytes"] / pr["launches"] / (pr["ms"] / pr["launches"] * 1e-3) / 1e9
        total_ms = sum(r["ms"] for r in warm_rows)
        traffic = None
        try:   # HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/)
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            for kname, rec in pt.items():
                if dominant.replace("radix_scatter", "k_seg_scatter").replace("emit_count", "k_emit_slots") in kname or dominant in kname:
                    traffic = rec["hbm_bytes_per_launch_corrected"]
        except (OSError, ValueError, KeyError):
            pass
        res = {
            "metric": "Gbases/s end-to-end BWT+SA build; bit-exact .bwt/.sa vs reference", "value": value, "unit": "Gbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 text / u32 indices / u64 hash", "data": "synthetic",
            "config": {"workload": a.workload, "L": L, "H": H, "seed": seed, "n_runs": list(nruns), "w": w, "p": p, "flags": "-s", "uint_t": 64 if u64 else 32,
                       "n": n, "r": r_total, "input": "text resident in HBM, outputs (.bwt, .sa) left in HBM",
                       "per_rank": ("haplotype r of the collection per rank; parse sharded, one RCCL all-gather of dictionaries, "
                                    "merge + dictionary/parse suffix sorts on every rank, emission sliced over the ranks") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                         "launches_per_step": pr["launches"] / a.steps, "avg_launch_us": 1e3 * pr["ms"] / pr["launches"],
                         "alg_bytes_per_launch": pr["bytes"] / pr["launches"],
                         "share_of_kernel_time": max(r["ms"] for r in warm_rows) / total_ms,
                         "end_to_end_alg_GBps": (6 if not u64 else 10) * n / (dt / a.steps) / 1e9},
            "stage_ms": ctx.stage_ms(),
        }
        if not a.no_cpu_baseline and world == 1:
            cb, dig = cpu_baseline(seqs, w, p, u64, True)
            res["cpu_baseline"] = cb
            ok = (hashlib.sha256(out["bwt"].tobytes()).hexdigest() == dig["bwt"] and hashlib.sha256(out["sa"].tobytes()).hexdigest() == dig["sa"])
            res["parity"] = "bit-exact (.bwt, .sa sha256 == CPU oracle on the same input)" if ok else "MISMATCH vs CPU oracle"
        print(json.dumps(res), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
