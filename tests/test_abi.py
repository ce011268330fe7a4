"""The C-ABI shared library loads without a GPU and exports every entry point that
include/nalu_gpu.h declares; the ctypes table of the Python host layer covers the same set.
No compute call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_entry_points():
    text = open(os.path.join(ROOT, "include", "nalu_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments name functions too
    return sorted(set(re.findall(r"\b(nalu_gpu_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import nalu_amd
    assert os.path.exists(nalu_amd.LIB_PATH), "libnalu_gpu.so is missing: run __graft_entry__.build()"
    return C.CDLL(nalu_amd.LIB_PATH)


def test_every_declared_symbol_is_exported(lib):
    names = declared_entry_points()
    assert len(names) >= 40 and "nalu_gpu_solve" in names and "nalu_gpu_assemble_continuity_edges" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/nalu_gpu.h but not exported: {missing}"


def test_python_binding_table_matches_the_header():
    from nalu_amd._lib import SYMBOLS
    assert sorted(SYMBOLS) == declared_entry_points()


def test_rccl_library_exports_its_header():
    """include/nalu_gpu_rccl.h <-> libnalu_gpu_rccl.so <-> nalu_amd.rccl.SYMBOLS (no call made)"""
    from nalu_amd import rccl
    text = open(os.path.join(ROOT, "include", "nalu_gpu_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(nalu_gpu_rccl_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(rccl.SYMBOLS) and len(names) == 7
    assert os.path.exists(rccl.RCCL_LIB_PATH), "libnalu_gpu_rccl.so is missing: run __graft_entry__.build()"
    import torch  # noqa: F401  (its bundled HIP / RCCL runtime is the one the library binds to)
    lib = C.CDLL(rccl.RCCL_LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_calls_that_need_no_device(lib):
    lib.nalu_gpu_version.restype = C.c_int
    lib.nalu_gpu_last_error.restype = C.c_char_p
    assert lib.nalu_gpu_version() >= 1
    # a null handle is an argument error, not a crash
    lib.nalu_gpu_zero_system.argtypes = [C.c_void_p]
    assert lib.nalu_gpu_zero_system(None) == 1  # NALU_GPU_EINVAL
    assert b"null" in lib.nalu_gpu_last_error().lower()
    lib.nalu_gpu_destroy.argtypes = [C.c_void_p]
    assert lib.nalu_gpu_destroy(None) in (0, 1)


def test_no_device_fails_loudly(lib):
    """the product has no CPU fallback: without a GPU, create() reports a HIP error"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    lib.nalu_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
    lib.nalu_gpu_last_error.restype = C.c_char_p
    rc = lib.nalu_gpu_create(C.byref(h), 0, 1, None)
    assert rc == 3 and not h.value  # NALU_GPU_EHIP
    assert lib.nalu_gpu_last_error()
    import nalu_amd
    with pytest.raises(nalu_amd.NaluGpuError):
        nalu_amd.GpuLinearSystem()
