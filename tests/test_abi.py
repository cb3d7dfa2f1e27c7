"""The C-ABI library loads and exports every symbol include/gsdf.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "gsdf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gsdf_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported(pkg):
    L = pkg.binding.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libgsdf.so does not export " + n
    assert sorted(pkg.binding.ABI_SYMBOLS) == names


def test_library_has_no_hip_runtime_dependency_and_no_torch(pkg):
    out = os.popen("readelf -d %s" % pkg.binding.LIB_PATH).read()
    assert "libamdhip64" not in out and "torch" not in out and "oracle" not in out


def test_debug_switches_only_in_the_test_build(pkg):
    """The path-forcing / measurement hooks live in libgsdf_test.so; the production library has none."""
    L = pkg.binding.load()
    assert not hasattr(L, "gsdf_debug_flags") and b"experiments" not in L.gsdf_version()
    nm = os.popen("nm -D --defined-only %s" % pkg.binding.LIB_PATH).read()
    assert "gsdf_debug" not in nm
    T = pkg.binding.load_test_lib()
    assert b"experiments" in T.gsdf_version() and hasattr(T, "gsdf_debug_flags")
    for n in _declared():
        assert hasattr(T, n), "libgsdf_test.so does not export " + n


def test_version_and_error_strings(pkg):
    L = pkg.binding.load()
    assert b"gfx950" in L.gsdf_version()
    assert isinstance(L.gsdf_last_error(), bytes)


def test_no_cpu_fallback(pkg):
    """Without a GPU the product path must fail loudly (GSDF_ERR_NO_DEVICE), never compute on the CPU."""
    L = pkg.binding.load()
    h = ctypes.c_void_p()
    rc = L.gsdf_create(ctypes.byref(h), np.float32(0.01), np.float32(0.1), 16, 0)
    if rc == 0:
        L.gsdf_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == pkg.binding.ERR_NO_DEVICE
    assert b"no CPU fallback" in L.gsdf_last_error()
    with pytest.raises(pkg.GsdfError):
        pkg.GradSdf(0.01, 0.1, 64, 48, pkg.synth.intrinsics(64, 48))


def test_bad_arguments_are_rejected(pkg):
    L = pkg.binding.load()
    h = ctypes.c_void_p()
    assert L.gsdf_create(ctypes.byref(h), np.float32(-1), np.float32(0.1), 16, 0) == pkg.binding.ERR_INVALID
    assert L.gsdf_create(ctypes.byref(h), np.float32(0.01), np.float32(0.1), 5, 0) == pkg.binding.ERR_INVALID
    assert L.gsdf_create(ctypes.byref(h), np.float32(0.2), np.float32(2.0), 16, 0) == pkg.binding.ERR_INVALID   # T < 2 m (fixed-point range)
    assert L.gsdf_sync(None) == pkg.binding.ERR_INVALID


def test_product_sources_do_not_reference_the_oracle():
    pdir = os.path.join(ROOT, "gradient-sdf_amd")
    for d, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(d, f)).read()
                assert "gsdf_oracle" not in txt and "import oracle" not in txt, f
