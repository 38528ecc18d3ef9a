"""CPU tests of the drop-in boundary: the product library loads and exports every symbol that
include/pfbwt_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess
import pytest
from pfp_testlib import ROOT

LIB = os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfbwt_hip.so")
HDR = os.path.join(ROOT, "include", "pfbwt_hip.h")
HDRS = [HDR, os.path.join(ROOT, "include", "pfbwt_hip_dev.h")]


def declared_symbols(hdrs=HDRS):
    out = set()
    for h in hdrs:
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(pfp_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_product_header_holds_no_development_hooks():
    syms = declared_symbols([HDR])
    assert not [s for s in syms if s.startswith("pfp_debug") or s.startswith("pfp_profile")], syms


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("pfp_create", "pfp_parse_feed", "pfp_parse_finalize", "pfp_parse_finalize_shard", "pfp_parse_bwt", "pfp_bwt_load", "pfp_bwt_build", "pfp_bwt_get", "pfp_sacak_int_u32"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "all"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.pfp_backend.restype = ctypes.c_char_p
    assert lib.pfp_backend() == b"hip-gfx950"
    lib.pfp_strerror.restype = ctypes.c_char_p
    assert b"invalid character" in lib.pfp_strerror(-2)


def test_library_contains_gfx950_code_object():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", LIB], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out
    blob = open(LIB, "rb").read()
    assert b"gfx950" in blob


def test_python_binding_fails_loudly_without_library(tmp_path):
    import pfbwt_hip
    with pytest.raises(OSError):
        pfbwt_hip.load_library(str(tmp_path / "nope.so"))
