"""The C-ABI library loads without a GPU and exports exactly what include/curv_hip.h declares."""
import os
import re

from curvature_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "curv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(curv_[a-z0-9_]+)\s*\(", text))


def test_library_builds_and_loads():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib().curv_version() == _lib.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    handle = _lib.lib()
    declared = declared_functions()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in curv_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_host_only_calls():
    """Entry points that do no device work behave without a GPU."""
    L = _lib.lib()
    assert L.curv_gemm_workspace_bytes(3) >= 3 * 64
    assert L.curv_kfac_accumulate(None, None, 0, None, 0) == 0          # empty batch is a no-op
    assert L.curv_chol_inv_lower(None, None, 0, None, None, 0) == 0
    assert L.curv_rsqrt_affine(None, None, 1.0, 0.0, None, 0) == 0
    bad = (_lib.curv_factor_desc * 1)()
    assert L.curv_kfac_workspace_bytes(bad, 1) == 0                      # invalid geometry -> 0, error text set
    assert b"factor 0" in L.curv_last_error()
