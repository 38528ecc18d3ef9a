import os
import sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Factory of engine contexts bound to the PRODUCT library (hipcc/gfx950).  Fails loudly when it is missing."""
    import pfp_testlib  # noqa: F401  (sets sys.path for pfbwt_hip)
    import pfbwt_hip
    lib = pfbwt_hip.load_library()  # raises OSError if pfbwt-f_amd/lib/libpfbwt_hip.so is absent
    assert lib.pfp_backend().decode() == "hip-gfx950"
    return lambda **kw: pfbwt_hip.PfpContext(**kw)
