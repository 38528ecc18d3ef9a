"""ctypes binding of libpfbwt_hip.so (include/pfbwt_hip.h) -- the MI355X engine.

Thin by design: numpy arrays in/out, no algorithmic work here.  The product library is
`pfbwt-f_amd/lib/libpfbwt_hip.so` (hipcc, gfx950).  If it is missing or cannot be loaded this module
raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), "lib", "libpfbwt_hip.so")

PFP_OK = 0
FLAG_U64, FLAG_NON_ACGT_TO_A, FLAG_SAI = 1, 2, 4
E_ARG = -1
E_INVALID_CHAR, E_TOO_LARGE, E_NOMEM, E_HIP, E_ONE_WORD, E_STATE, E_CORRUPT = -2, -3, -4, -5, -6, -7, -8


class PfpError(RuntimeError):
    def __init__(self, status, msg, pos=None, ch=None):
        super().__init__("pfbwt_hip: %s (status %d)" % (msg, status))
        self.status, self.pos, self.ch = status, pos, ch


class ParseSizes(C.Structure):
    _fields_ = [("n", C.c_uint64), ("m", C.c_uint64), ("dwords", C.c_uint64), ("dsize", C.c_uint64)]


class BwtSizes(C.Structure):
    _fields_ = [("nout", C.c_uint64), ("r", C.c_uint64), ("easy_cases", C.c_uint64), ("hard_cases", C.c_uint64)]


class ShardView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("m", C.c_uint64), ("dwords", C.c_uint64), ("dsize", C.c_uint64),
                ("d_dict", C.c_void_p), ("d_ws", C.c_void_p), ("d_pid", C.c_void_p), ("d_ye", C.c_void_p), ("d_last", C.c_void_p),
                ("left_context", C.c_uint64)]

    def nbytes(self, compact=False):
        """byte sizes of the device arrays, in field order (compact: dictionary, word starts, phrase ids -- pfp_merge_shards derives
        the phrase ends and last bytes)"""
        return [self.dsize, 4 * (self.dwords + 1), 4 * self.m] + ([] if compact else [8 * self.m, self.m])


_libs = {}


class IngestInfo(C.Structure):
    _fields_ = [("raw_bytes", C.c_uint64), ("records", C.c_uint64), ("n", C.c_uint64), ("read_wait_ms", C.c_double), ("total_ms", C.c_double), ("mode", C.c_int)]


def load_library(path=None):
    path = os.path.abspath(path or os.environ.get("PFBWT_HIP_LIB") or DEFAULT_LIB)   # PFBWT_HIP_LIB: another build of the same library
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OSError("HIP library %s not built: run `make -C pfbwt-f_amd` (hipcc --offload-arch=gfx950)" % path)
    L = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.pfp_create.restype = vp
    L.pfp_create.argtypes = [i32, u64, C.c_uint, i32, u64, C.POINTER(i32)]
    L.pfp_destroy.argtypes = [vp]
    L.pfp_destroy.restype = None
    L.pfp_strerror.restype = C.c_char_p
    L.pfp_strerror.argtypes = [i32]
    L.pfp_backend.restype = C.c_char_p
    L.pfp_error_detail.argtypes = [vp, C.POINTER(u64), C.POINTER(i32)]
    L.pfp_workspace_needed.restype = u64
    L.pfp_workspace_needed.argtypes = [vp]
    L.pfp_reset.argtypes = [vp]
    L.pfp_parse_feed.argtypes = [vp, vp, u64, i32]
    L.pfp_parse_feed_batch.argtypes = [vp, vp, u64, u64, u64]
    L.pfp_parse_feed_device.argtypes = [vp, vp, u64, i32]
    L.pfp_parse_finalize.argtypes = [vp, C.POINTER(ParseSizes)]
    L.pfp_parse_finalize_shard.argtypes = [vp, C.POINTER(ParseSizes)]
    L.pfp_parse_get.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pfp_parse_bwt.argtypes = [vp]
    L.pfp_parse_bwt_get.argtypes = [vp, vp, vp, vp]
    L.pfp_bwt_load.argtypes = [vp, vp, u64, vp, u64, vp, vp, vp, u64, u64]
    L.pfp_bwt_build.argtypes = [vp, i32, i32, C.POINTER(BwtSizes)]
    L.pfp_parse_feed_device_batch.argtypes = [vp, vp, u64, u64, u64]
    L.pfp_bwt_build_slice.argtypes = [vp, i32, i32, i32, i32, C.POINTER(BwtSizes), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.pfp_bwt_get.argtypes = [vp, vp, vp, vp, vp]
    L.pfp_bwt_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pfp_shard_view_get.argtypes = [vp, C.POINTER(ShardView)]
    L.pfp_parse_feed_left_context.argtypes = [vp]
    L.pfp_shard_load.argtypes = [vp, vp, u64, vp, u64]
    L.pfp_device_copy.argtypes = [vp, vp, vp, u64]
    L.pfp_merge_shards.argtypes = [vp, i32, C.POINTER(ShardView), C.POINTER(ParseSizes)]
    L.pfp_sacak_int_u32.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.pfp_sacak_int_u64.argtypes = [vp, vp, u64, u64]
    L.pfp_marker_array.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.pfp_marker_array_get.argtypes = [vp, vp]
    L.pfp_gsacak_u32.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.pfp_gsacak_u64.argtypes = [vp, vp, vp, vp, u64]
    L.pfp_profile_enable.argtypes = [vp, i32]
    L.pfp_profile_reset.argtypes = [vp]
    L.pfp_profile_select.argtypes = [vp, C.c_char_p]
    L.pfp_profile_get.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.pfp_stage_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.pfp_debug_set.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.pfp_parse_feed_fasta.argtypes = [vp, vp, u64, C.c_uint, C.POINTER(u64)]
    L.pfp_parse_fasta_records.argtypes = [vp, vp, vp]
    L.pfp_parse_reserve.argtypes = [vp, u64]
    L.pfp_parse_feed_fasta_file.argtypes = [vp, C.c_char_p, C.c_uint, C.POINTER(IngestInfo)]
    L.pfp_bwt_build_stream.argtypes = [vp, i32, i32, vp, vp, C.POINTER(BwtSizes)]
    L.pfp_text_length.argtypes = [vp, C.POINTER(u64)]
    L.pfp_bwt_get_expanded.argtypes = [vp, vp, vp, i32]
    L.pfp_text_view.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.pfp_host_register.argtypes = [vp, u64]
    L.pfp_host_unregister.argtypes = [vp]
    L.pfp_debug_wordsum.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.pfp_debug_check_sample_order.argtypes = [vp, C.POINTER(u64)]
    L.pfp_debug_check_sa.argtypes = [vp, C.POINTER(u64)]
    L.pfp_parse_feed_device_view.argtypes = [vp, vp, u64, u64, u64]
    L.pfp_debug_check_samples.argtypes = [vp, C.POINTER(u64)]
    L.pfp_sharded_create.restype = vp
    L.pfp_sharded_create.argtypes = [i32, u64, C.c_uint, i32, C.POINTER(i32), u64, C.POINTER(i32)]
    L.pfp_sharded_destroy.argtypes = [vp]; L.pfp_sharded_destroy.restype = None
    L.pfp_sharded_ctx.restype = vp; L.pfp_sharded_ctx.argtypes = [vp, i32]
    L.pfp_sharded_ranks.argtypes = [vp]
    L.pfp_sharded_build.argtypes = [vp, i32, i32, vp, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.pfp_sharded_reset.argtypes = [vp]
    L.pfp_sharded_error.restype = C.c_char_p; L.pfp_sharded_error.argtypes = [vp]
    L.pfp_parse_docs.argtypes = [vp, C.POINTER(u64)]
    L.pfp_parse_doc_get.argtypes = [vp, u64, C.POINTER(C.c_char_p), C.POINTER(u64)]
    _libs[path] = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class PfpContext:
    """One engine context = one HIP stream + one device workspace (include/pfbwt_hip.h)."""

    def __init__(self, w=10, p=100, u64=True, non_acgt_to_a=False, sai=True, device=0, workspace_bytes=0, lib=None):
        self.L = load_library(lib)
        self.u64 = bool(u64)
        self.udt = np.uint64 if u64 else np.uint32
        flags = (FLAG_U64 if u64 else 0) | (FLAG_NON_ACGT_TO_A if non_acgt_to_a else 0) | (FLAG_SAI if sai else 0)
        self.sai = bool(sai)
        st = C.c_int(0)
        self.h = self.L.pfp_create(int(w), int(p), flags, int(device), int(workspace_bytes), C.byref(st))
        if not self.h:
            raise PfpError(st.value, self.L.pfp_strerror(st.value).decode())
        self.sizes = None
        self.bsizes = None

    def debug_set(self, **switches):
        """route / tuning switches of this context (include/pfbwt_hip_dev.h: pfp_debug_set), e.g. force_wide_rows=1"""
        for k, v in switches.items():
            self._check(self.L.pfp_debug_set(self.h, k.encode(), int(v)))

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):      # (the contexts of a ShardedBuild belong to its handle)
                self.L.pfp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != PFP_OK:
            pos, ch = C.c_uint64(0), C.c_int(0)
            self.L.pfp_error_detail(self.h, C.byref(pos), C.byref(ch))
            raise PfpError(st, self.L.pfp_strerror(st).decode(), pos.value, ch.value)

    def reset(self):
        self._check(self.L.pfp_reset(self.h))

    # ---- stage 1
    def feed(self, bases, end_of_seq=True):
        a = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray, memoryview)) else np.ascontiguousarray(bases, dtype=np.uint8)
        self._check(self.L.pfp_parse_feed(self.h, _ptr(a) if a.size else None, a.size, 1 if end_of_seq else 0))

    def reserve(self, text_bytes):
        self._check(self.L.pfp_parse_reserve(self.h, int(text_bytes)))

    def feed_fasta(self, raw, final=True, records=False, ptr=None, nbytes=None):
        """raw FASTA bytes (bytes / uint8 array, or ptr + nbytes of host memory); returns the (raw offset, text position) pairs of
        the records that start in this call when records=True"""
        if ptr is None:
            a = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray, memoryview)) else np.ascontiguousarray(raw, dtype=np.uint8)
            ptr, nbytes = a.ctypes.data, a.size
        nrec = C.c_uint64(0)
        self._check(self.L.pfp_parse_feed_fasta(self.h, C.c_void_p(ptr), int(nbytes), (1 if final else 0) | (2 if records else 0), C.byref(nrec)))
        if not records:
            return None
        ro = np.empty(nrec.value, np.uint64); tp = np.empty(nrec.value, np.uint64)
        self._check(self.L.pfp_parse_fasta_records(self.h, _ptr(ro), _ptr(tp)))
        return ro, tp

    def feed_fasta_file(self, path, records=False):
        info = IngestInfo()
        self._check(self.L.pfp_parse_feed_fasta_file(self.h, os.fsencode(path), 2 if records else 0, C.byref(info)))
        return info

    def text_length(self):
        n = C.c_uint64(0); self._check(self.L.pfp_text_length(self.h, C.byref(n))); return n.value

    def bwt_build_stream(self, host_bwt_ptr, host_sa_ptr=None, rssa=False):
        """emission with the rows streamed to host memory (n + 1 bytes at host_bwt_ptr, n + 1 U-wide values at host_sa_ptr)"""
        b = BwtSizes()
        self._check(self.L.pfp_bwt_build_stream(self.h, 1 if host_sa_ptr else 0, 1 if rssa else 0, C.c_void_p(host_bwt_ptr), C.c_void_p(host_sa_ptr) if host_sa_ptr else None, C.byref(b)))
        self.bsizes, self._want, self._rows, self.esa_pairs = b, (bool(host_sa_ptr), bool(rssa)), b.nout, b.r
        return b

    def check_sample_order(self):
        """adjacent-row order check of the run samples on the device (include/pfbwt_hip_dev.h)"""
        o = (C.c_uint64 * 5)()
        self._check(self.L.pfp_debug_check_sample_order(self.h, o))
        return {"pairs": int(o[0]), "order_violations": int(o[1]), "rows_not_adjacent": int(o[2]), "max_lcp": int(o[3]), "mean_lcp": (int(o[4]) / int(o[0])) if o[0] else 0.0}

    def check_sa(self):
        """permutation / T[SA-1] == BWT / one EOS byte, checked on the device against the resident text (include/pfbwt_hip_dev.h)"""
        o = (C.c_uint64 * 5)()
        self._check(self.L.pfp_debug_check_sa(self.h, o))
        return {"rows": int(o[0]), "out_of_range": int(o[1]), "duplicates": int(o[2]), "bwt_mismatches": int(o[3]), "eos_bytes": int(o[4])}

    def check_samples(self):
        """the run samples of a -s -r build against its own .bwt / .sa, on the device (include/pfbwt_hip_dev.h)"""
        o = (C.c_uint64 * 3)()
        self._check(self.L.pfp_debug_check_samples(self.h, o))
        return {"runs": int(o[0]), "row_errors": int(o[1]), "value_errors": int(o[2])}

    def bwt_get_expanded(self, host_bwt_ptr, ssa=None, threads=16):
        """.bwt into host memory from its run-length form (one byte per run over PCIe, host threads write the runs)"""
        self._check(self.L.pfp_bwt_get_expanded(self.h, C.c_void_p(host_bwt_ptr), _ptr(ssa), int(threads)))

    def samples_get(self, out=None):
        """the run samples of the last build: (ssa, esa) as 2*r U-wide arrays (into out["ssa"] / out["esa"] if given)"""
        b = self.bsizes
        ssa = out["ssa"] if out else np.empty(2 * b.r, self.udt); esa = out["esa"] if out else np.empty(2 * b.r, self.udt)
        self._check(self.L.pfp_bwt_get(self.h, None, None, _ptr(ssa), _ptr(esa)))
        return ssa, esa

    def feed_host_batch(self, host_ptr, count, length, stride):
        """`count` equal-length records in host memory (pinned: one strided DMA transfer; pageable: staging ring)"""
        self._check(self.L.pfp_parse_feed_batch(self.h, C.c_void_p(int(host_ptr)), int(count), int(length), int(stride)))

    def feed_device_view(self, dptr, count, length, stride):
        """`count` equal-length records in device memory become the text of this parse WITHOUT a copy (they must stay valid until
        finalize returns): pfp_parse_feed_device_view"""
        self._check(self.L.pfp_parse_feed_device_view(self.h, C.c_void_p(int(dptr)), int(count), int(length), int(stride)))

    def feed_device(self, dptr, nbytes, end_of_seq=True):
        self._check(self.L.pfp_parse_feed_device(self.h, C.c_void_p(int(dptr)), int(nbytes), 1 if end_of_seq else 0))

    def finalize(self, shard=False):
        """shard=True: pfp_parse_finalize_shard -- phrases, dictionary words and ids only (a shard that is going to be merged)"""
        s = ParseSizes()
        self._check((self.L.pfp_parse_finalize_shard if shard else self.L.pfp_parse_finalize)(self.h, C.byref(s)))
        self.sizes = s
        return s

    def parse_get(self):
        s = self.sizes
        out = {"dict": np.empty(s.dsize, np.uint8), "occ": np.empty(s.dwords, self.udt), "parse": np.empty(s.m, np.uint32),
               "last": np.empty(s.m, np.uint8), "sai": np.empty(s.m, self.udt) if self.sai else None}
        self._check(self.L.pfp_parse_get(self.h, _ptr(out["dict"]), _ptr(out["occ"]), _ptr(out["parse"]), _ptr(out["last"]), _ptr(out["sai"])))
        return out

    def parse_bwt(self):
        self._check(self.L.pfp_parse_bwt(self.h))

    def parse_bwt_get(self):
        nr = self.sizes.m + 1
        out = {"bwlast": np.empty(nr, np.uint8), "ilist": np.empty(nr, self.udt), "bwsai": np.empty(nr, self.udt) if self.sai else None}
        self._check(self.L.pfp_parse_bwt_get(self.h, _ptr(out["bwlast"]), _ptr(out["ilist"]), _ptr(out["bwsai"])))
        return out

    # ---- multi-GPU sharding
    def feed_device_batch(self, dev_ptr, count, length, stride):
        """`count` equal-length records that already sit in device memory, `stride` bytes apart"""
        self._check(self.L.pfp_parse_feed_device_batch(self.h, C.c_void_p(int(dev_ptr)), int(count), int(length), int(stride)))

    def feed_left_context(self, w=None):
        """shard r > 0: the w 'A's that end the previous shard (pfparser.hpp:335-337); recorded in the shard view"""
        self._check(self.L.pfp_parse_feed_left_context(self.h))

    def shard_load(self, dict_image, parse):
        """a parse saved as .dict / .parse becomes a (stand-alone) shard on the device"""
        d = np.ascontiguousarray(dict_image, np.uint8); p = np.ascontiguousarray(parse, np.uint32)
        self._check(self.L.pfp_shard_load(self.h, _ptr(d), d.size, _ptr(p), p.size))

    def shard_view(self):
        v = ShardView()
        self._check(self.L.pfp_shard_view_get(self.h, C.byref(v)))
        return v

    def device_copy(self, dst_ptr, src_ptr, nbytes):
        self._check(self.L.pfp_device_copy(self.h, C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), int(nbytes)))

    def merge_shards(self, views):
        arr = (ShardView * len(views))(*views)
        s = ParseSizes()
        self._check(self.L.pfp_merge_shards(self.h, len(views), arr, C.byref(s)))
        self.sizes = s
        return s

    # ---- stage 2
    def bwt_load(self, dict_, occ, bwlast, ilist, bwsai=None, n_hint=0):
        d = np.ascontiguousarray(dict_, np.uint8); o = np.ascontiguousarray(occ, self.udt)
        bl = np.ascontiguousarray(bwlast, np.uint8); il = np.ascontiguousarray(ilist, self.udt)
        bs = None if bwsai is None else np.ascontiguousarray(bwsai, self.udt)
        if il.size != bl.size or (bs is not None and bs.size != bl.size):      # the ABI takes one row count for the three arrays
            raise PfpError(E_CORRUPT, "bwlast / ilist / bwsai hold different numbers of rows")
        self._check(self.L.pfp_bwt_load(self.h, _ptr(d), d.size, _ptr(o), o.size, _ptr(bl), _ptr(il), _ptr(bs), bl.size, int(n_hint)))

    def bwt_build(self, sa=True, rssa=False):
        b = BwtSizes()
        self._check(self.L.pfp_bwt_build(self.h, 1 if sa else 0, 1 if rssa else 0, C.byref(b)))
        self.bsizes, self._want, self._rows, self.esa_pairs = b, (bool(sa), bool(rssa)), b.nout, b.r
        return b

    def bwt_build_slice(self, slice_index, nslices, sa=True, rssa=False):
        """multi-GPU emission: this context emits only its slice of the output rows; returns (sizes, begin, rows).
        rssa: also the slice's part of the run samples (b.r ssa pairs, self.esa_pairs esa pairs; see include/pfbwt_hip.h)"""
        b = BwtSizes(); beg, rows, ep = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.pfp_bwt_build_slice(self.h, 1 if sa else 0, 1 if rssa else 0, int(slice_index), int(nslices), C.byref(b), C.byref(beg), C.byref(rows), C.byref(ep)))
        self.bsizes, self._want, self._rows, self.esa_pairs = b, (bool(sa), bool(rssa)), rows.value, ep.value
        return b, beg.value, rows.value

    def bwt_get(self, out=None):
        """copies the results to host arrays; `out` may hold caller-owned (e.g. page-locked) arrays bwt / sa / ssa / esa that
        are at least as large as needed"""
        b = self.bsizes
        sa, rssa = self._want
        if out is not None:
            need = {"bwt": self._rows, "sa": self._rows if sa else None, "ssa": 2 * b.r if rssa else None, "esa": 2 * getattr(self, "esa_pairs", b.r) if rssa else None}
            for k, cnt in need.items():
                if cnt is not None and (out.get(k) is None or out[k].size < cnt or out[k].dtype != (np.uint8 if k == "bwt" else self.udt)):
                    raise ValueError("bwt_get: out[%r] missing, too small or of the wrong type" % k)
            self._check(self.L.pfp_bwt_get(self.h, _ptr(out["bwt"]), _ptr(out["sa"]) if sa else None, _ptr(out["ssa"]) if rssa else None, _ptr(out["esa"]) if rssa else None))
            return out
        out = {"bwt": np.empty(self._rows, np.uint8), "sa": np.empty(self._rows, self.udt) if sa else None,
               "ssa": np.empty(2 * b.r, self.udt) if rssa else None, "esa": np.empty(2 * getattr(self, "esa_pairs", b.r), self.udt) if rssa else None}
        self._check(self.L.pfp_bwt_get(self.h, _ptr(out["bwt"]), _ptr(out["sa"]), _ptr(out["ssa"]), _ptr(out["esa"])))
        return out

    def bwt_device_ptrs(self):
        p = [C.c_void_p(0) for _ in range(4)]
        self._check(self.L.pfp_bwt_device_ptrs(self.h, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def marker_array(self, mps, sa=None):
        """marker-array post-pass (include/marker_array.hpp:138-174): the .ma stream (uint64 words) for the .mps stream `mps`;
        sa=None: fused with the last bwt_build(sa=True) of this context, else the suffix array given (uint_t values, BWT order)"""
        mps = np.ascontiguousarray(mps, np.uint64)
        n = C.c_uint64(0)
        if sa is None:
            self._check(self.L.pfp_marker_array(self.h, _ptr(mps), mps.size, None, 0, C.byref(n)))
        else:
            sa = np.ascontiguousarray(sa, self.udt)
            self._check(self.L.pfp_marker_array(self.h, _ptr(mps), mps.size, _ptr(sa), sa.size, C.byref(n)))
        out = np.empty(n.value, np.uint64)
        self._check(self.L.pfp_marker_array_get(self.h, _ptr(out) if n.value else None))
        return out

    # ---- instrumentation
    def profile_enable(self, on=True):
        self._check(self.L.pfp_profile_enable(self.h, 1 if on else 0))

    def profile_select(self, kernel):
        self._check(self.L.pfp_profile_select(self.h, kernel.encode()))

    def profile_reset(self):
        self._check(self.L.pfp_profile_reset(self.h))

    def profile(self):
        rows, i = [], 0
        while True:
            name, n, ms, by = C.c_char_p(), C.c_uint64(), C.c_double(), C.c_double()
            if self.L.pfp_profile_get(self.h, i, C.byref(name), C.byref(n), C.byref(ms), C.byref(by)) != PFP_OK:
                break
            if n.value:
                rows.append({"kernel": name.value.decode(), "launches": n.value, "ms": ms.value, "bytes": by.value})
            i += 1
        return rows

    def stage_ms(self):
        a = (C.c_double * 3)()
        self._check(self.L.pfp_stage_ms(self.h, a))
        return {"parse_finalize": a[0], "parse_bwt": a[1], "bwt_build": a[2]}

    def backend(self):
        return self.L.pfp_backend().decode()


class ShardedBuild:
    """N devices driven from one process (include/pfbwt_hip.h: pfp_sharded_*): rank r's run of whole sequences is fed into
    self.rank(r) (a PfpContext view of the library-owned context), build() makes every rank parse, exchange (RCCL between distinct
    devices), merge, sort and emit its slice; self.rank(r).bwt_get() then returns slice r."""

    def __init__(self, ndev, devices=None, w=10, p=100, u64=True, non_acgt_to_a=False, sai=True, workspace_bytes=0, lib=None):
        self.L = load_library(lib)
        flags = (FLAG_U64 if u64 else 0) | (FLAG_NON_ACGT_TO_A if non_acgt_to_a else 0) | (FLAG_SAI if sai else 0)
        st = C.c_int(0)
        dv = (C.c_int * ndev)(*devices) if devices is not None else None
        self.h = self.L.pfp_sharded_create(int(w), int(p), flags, int(ndev), dv, int(workspace_bytes), C.byref(st))
        if not self.h:
            raise PfpError(st.value, self.L.pfp_strerror(st.value).decode())
        self.ndev = ndev
        self._ranks = []
        for r in range(ndev):      # views of the library-owned contexts (closing them is the sharded handle's business)
            c = PfpContext.__new__(PfpContext)
            c.L, c.u64, c.udt, c.sai, c.sizes, c.bsizes = self.L, bool(u64), (np.uint64 if u64 else np.uint32), bool(sai), None, None
            c.h = self.L.pfp_sharded_ctx(self.h, r); c._borrowed = True
            self._ranks.append(c)

    def rank(self, r):
        return self._ranks[r]

    def build(self, sa=True, rssa=False):
        """returns (parse sizes of the whole collection, [(BwtSizes, first row, rows) per rank])"""
        n = self.ndev
        ps = ParseSizes(); bs = (BwtSizes * n)(); beg = (C.c_uint64 * n)(); rows = (C.c_uint64 * n)(); ep = (C.c_uint64 * n)()
        st = self.L.pfp_sharded_build(self.h, 1 if sa else 0, 1 if rssa else 0, C.byref(ps), bs, beg, rows, ep)
        if st != PFP_OK:
            raise PfpError(st, self.L.pfp_strerror(st).decode() + " [" + self.L.pfp_sharded_error(self.h).decode() + "]")
        out = []
        for r, c in enumerate(self._ranks):
            b = BwtSizes(); C.memmove(C.byref(b), C.byref(bs[r]), C.sizeof(BwtSizes))
            c.bsizes, c._want, c._rows, c.esa_pairs = b, (bool(sa), bool(rssa)), int(rows[r]), int(ep[r])
            out.append((b, int(beg[r]), int(rows[r])))
        return ps, out

    def reset(self):
        st = self.L.pfp_sharded_reset(self.h)
        if st != PFP_OK:
            raise PfpError(st, self.L.pfp_strerror(st).decode())

    def close(self):
        if getattr(self, "h", None):
            for c in self._ranks:
                c.h = None
            self.L.pfp_sharded_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sacak_int(s, k, u64=False, lib=None):
    """Drop-in for sacak_int (gsa/gsacak.h:88): suffix array of an integer string ending in a unique 0."""
    L = load_library(lib)
    s = np.ascontiguousarray(s, np.uint32)
    SA = np.empty(s.size, np.uint64 if u64 else np.uint32)
    fn = L.pfp_sacak_int_u64 if u64 else L.pfp_sacak_int_u32
    r = fn(_ptr(s), _ptr(SA), s.size, int(k))
    if r < 0:
        raise PfpError(r, "sacak_int failed")
    return SA, r


def gsacak(s, lcp=False, da=False, u64=False, lib=None):
    """Drop-in for gsacak (gsa/gsacak.h:86-96) on a dictionary image: returns (SA, LCP or None, DA or None, rounds)."""
    L = load_library(lib)
    s = np.ascontiguousarray(s, np.uint8)
    ut, it = (np.uint64, np.int64) if u64 else (np.uint32, np.int32)
    SA = np.empty(s.size, ut); LCP = np.empty(s.size, it) if lcp else None; DA = np.empty(s.size, it) if da else None
    r = (L.pfp_gsacak_u64 if u64 else L.pfp_gsacak_u32)(_ptr(s), _ptr(SA), _ptr(LCP), _ptr(DA), s.size)
    if r < 0:
        raise PfpError(r, "gsacak failed")
    return SA, LCP, DA, r
