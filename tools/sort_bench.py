#!/usr/bin/env python3
"""Development aid: time the engine's pair sort alone (pfp_debug_sort, include/pfbwt_hip_dev.h).
usage: tools/sort_bench.py [pairs] [key bits]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pfbwt-f_amd", "python"))
import pfbwt_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 55_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = pfbwt_hip.PfpContext(lib=os.environ.get("PFP_LIB"))
L = ctx.L
L.pfp_debug_sort.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
ms, bad = C.c_double(), C.c_uint32()
rc = L.pfp_debug_sort(ctx.h, n, bits, 3, C.byref(ms), C.byref(bad))
passes = (bits + 7) // 8
print("rc %d  %.3f ms total, %.3f ms/pass, %.0f GB/s alg per pass, unsorted %d" % (rc, ms.value, ms.value / passes, n * 24 / (ms.value / passes) / 1e6, bad.value), flush=True)
