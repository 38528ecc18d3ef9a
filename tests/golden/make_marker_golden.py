#!/usr/bin/env python3
"""Fixtures of the marker-array post-pass, made in the build container from the reference's own test data:
  tests/data/<case>.sa       (text, the golden suffix array -- already kept as reference_golden.sa.u64.gz)
  tests/data/<case>.markers  (text form of the golden marker array, scripts/readable_markers.py) -> reference_golden.markers.gz
The goldens were produced with --ma_wsize 1 (tests/vcf_to_bwt_test.sh:29), so every marker list holds one marker and the
marker-positions stream can be recovered from (sa, markers): markers.mps.  expected.ma = oracle/marker_oracle.py on
(markers.mps, sa); its readable form must equal the reference's golden text (asserted here and in tests/test_markers.py).
usage: python3 tests/golden/make_marker_golden.py   (needs /root/reference)"""
import gzip, hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import marker_oracle as mo
REF = "/root/reference/tests/data"
for case in ("single_chrom", "mult_chroms"):
    sa = np.array([int(x) for x in open(os.path.join(REF, case + ".sa")).read().split()], np.uint64)
    kept = np.frombuffer(gzip.open(os.path.join(HERE, case, "reference_golden.sa.u64.gz")).read(), "<u8")
    assert np.array_equal(sa, kept)
    txt = open(os.path.join(REF, case + ".markers")).read()
    mps = mo.mps_from_golden(sa, txt)
    ma = mo.marker_array(mps, sa)
    assert mo.readable(ma) == txt, case
    with gzip.GzipFile(os.path.join(HERE, case, "reference_golden.markers.gz"), "wb", mtime=0) as f:
        f.write(txt.encode())
    mps.astype("<u8").tofile(os.path.join(HERE, case, "markers.mps"))
    json.dump({"mps_words": int(mps.size), "ma_words": int(ma.size), "ma_sha256": hashlib.sha256(ma.astype("<u8").tobytes()).hexdigest(),
               "rows_with_markers": len(txt.splitlines())}, open(os.path.join(HERE, case, "markers.json"), "w"), indent=1)
    print(case, "mps words", mps.size, "ma words", ma.size)
