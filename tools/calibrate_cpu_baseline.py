#!/usr/bin/env python3
"""CPU-baseline calibration (VERDICT r01 missing #7): the repo's CPU restatement (oracle/) timed beside
binaries built from the reference's OWN sources (oracle/_ref, `make -C oracle ref`) on the same input in the same
container, stage by stage where the reference stage is buildable here:
  * parse + parse-BWT : oracle/pfbwt_oracle --parse-only      vs  oracle/_ref/merge_pfp64 (src/merge_pfp.cpp + pfparser.hpp)
  * dictionary gSA+LCP: oracle orc_gsa_lcp (SA-IS + Kasai)     vs  oracle/_ref/libgsacak64.so gsacak() (gsa/gsacak.c)
  * emission          : not buildable from the reference here (include/pfbwt.hpp needs sdsl-lite); the survey-time figure of
                        the compiled reference is quoted next to the oracle's.
Writes profiles/<tag>_cpu_calibration.json.  Needs /root/reference only through the prebuilt oracle/_ref files."""
import ctypes as C, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import synth_seqs
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
L, H, seed, w, p = 5_000_000, 10, 12345, 10, 100           # S-50M
tmp = tempfile.mkdtemp(prefix="pfp_calib_")
fa = os.path.join(tmp, "s50m.fa")
seqs = synth_seqs(L, H, seed, (0, 0, 0, 0))
with open(fa, "wb") as f:
    for i, s in enumerate(seqs):
        f.write(b">hap%d\n" % i)
        for k in range(0, s.size, 60000):
            f.write(s[k:k + 60000].tobytes()); f.write(b"\n")
n = H * (L + w)


def timed(cmd):
    t0 = time.time(); pr = subprocess.run(cmd, capture_output=True, text=True); dt = time.time() - t0
    assert pr.returncode == 0, pr.stderr[-800:]
    return dt, pr.stderr


def best(cmd, reps=3):
    return min(timed(cmd) for _ in range(reps))


res = {"input": "S-50M: %d haplotypes x %d bases, seed %d, n = %d, -w %d -p %d, 64-bit" % (H, L, seed, n, w, p), "machine": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip(), "threads": 1}
orc = os.path.join(ROOT, "oracle", "pfbwt_oracle")
ref_merge = os.path.join(ROOT, "oracle", "_ref", "merge_pfp64")
# parse (+ parse-BWT): wall of the whole process, both read the same FASTA and write the same files
dt_o, err = best([orc, "--parse-only", "-s", "--u64", "-w", str(w), "-p", str(p), "-o", os.path.join(tmp, "o"), fa])
dt_r, _ = best([ref_merge, "-w", str(w), "-p", str(p), "-s", "--parse-bwt", "-o", os.path.join(tmp, "r"), fa])
same = all(open(os.path.join(tmp, "o." + e), "rb").read() == open(os.path.join(tmp, "r." + e), "rb").read() for e in ("dict", "occ", "parse", "bwlast", "ilist", "bwsai"))
res["parse_plus_parse_bwt"] = {"oracle_s": dt_o, "reference_merge_pfp64_s": dt_r, "oracle_over_reference": dt_o / dt_r, "files_identical": same}
# dictionary suffix sort + LCP
d = np.fromfile(os.path.join(tmp, "r.dict"), np.uint8); dn = d.size
dwords = int((d == 1).sum())
G = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libgsacak64.so"))
G.gsacak.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
O.orc_gsa_lcp.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
SA = np.zeros(dn, np.uint64); LCP = np.zeros(dn, np.int64); a = np.zeros(dn, np.uint64); b = np.zeros(dn, np.uint64)
tr = []; to = []
for _ in range(2):
    t0 = time.time(); G.gsacak(d.ctypes.data_as(C.c_void_p), SA.ctypes.data_as(C.c_void_p), LCP.ctypes.data_as(C.c_void_p), None, dn); tr.append(time.time() - t0)
    t0 = time.time(); assert O.orc_gsa_lcp(d.ctypes.data_as(C.c_void_p), dn, dwords, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0; to.append(time.time() - t0)
res["dict_gsa_lcp"] = {"dict_bytes": int(dn), "oracle_s": min(to), "reference_gsacak_s": min(tr), "oracle_over_reference": min(to) / min(tr),
                       "arrays_identical": bool(np.array_equal(SA, a) and np.array_equal(LCP.astype(np.uint64), b))}
# whole pipeline of the oracle, for the emission share
dt_all, err = best([orc, "-s", "-r", "--u64", "-w", str(w), "-p", str(p), "-o", os.path.join(tmp, "o"), fa], reps=1)
st = {}
for line in err.splitlines():
    if line.startswith("TASK\t"):
        _, nm, sec = line.split("\t"); st[nm] = float(sec.rstrip("s"))
res["oracle_full_pipeline"] = {"wall_s": dt_all, "stages_s": st, "Mbases_per_s": n / sum(v for k, v in st.items() if k != "reading input") / 1e6,
                               "reference_survey_note": "compiled reference, same shape of input, survey container (BASELINE.md section 2): 12.25 s wall, BWT stage 10.64 s incl. gsacak 2.29 s -- per-record fwrite in out_fn"}
for fn in os.listdir(tmp):
    os.remove(os.path.join(tmp, fn))
os.rmdir(tmp)
out = os.path.join(ROOT, "profiles", tag + "_cpu_calibration.json")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
