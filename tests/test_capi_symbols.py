"""CPU: the C-ABI library loads and exports every symbol include/*.h declares
(no compute call is made without a GPU), and the Python binding refuses to run
without a device instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names += re.findall(r"\b(sc_[a-z_]+)\s*\(", txt)
    return sorted(set(names))


def test_library_exports_header_symbols():
    from rambl_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.EXPORTS) == names


def test_no_cpu_fallback():
    import torch
    from rambl_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.StrainCallError) as e:
        capi.Context(0, 1)
    assert e.value.code == -1


def test_product_never_touches_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rambl_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"oracle/|liboracle|straincall_oracle|import\s+oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert bad == []
