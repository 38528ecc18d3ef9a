#!/usr/bin/env python3
"""One-off (round 4): parse of a synthetic panel of H sequences x L bases with PFP_VERBOSE=1 -- with profiles/r04fw_follow_path.patch applied it prints the hit rate
of the "follow the previous match" path of the text de-duplication (PFP_TEST_HOOKS=1 PFP_DEDUP_FOLLOW=<phrases of the first launch>; the path was measured slower and is not
in the tree); without the patch: sizes of the parse and, for collections, the order in which the de-duplication visits the text.
usage: python tools/follow_check.py H L [u64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import bench, pfbwt_hip
H, L = int(sys.argv[1]), int(sys.argv[2]); u64 = len(sys.argv) > 3
seqs = bench.synth_seqs(L, H, 1000, (0, 0, 0, 0))
kw = dict(lib=os.environ["PFBWT_LIB"]) if os.environ.get("PFBWT_LIB") else dict(device=0)
c = pfbwt_hip.PfpContext(w=10, p=100, u64=u64, sai=True, **kw)
for t in seqs: c.feed(t, True)
sz = c.finalize(); print("n=%d m=%d dwords=%d" % (sz.n, sz.m, sz.dwords), flush=True)
c.close()
