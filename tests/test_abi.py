"""The drop-in boundary: libnprealign.so loads, exports every symbol include/nprealign.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

from helpers import ROOT
from nanopore_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nprealign.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(npr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libnprealign.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared            # the binding covers the whole header
    assert L.npr_abi_version() == 1


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_lib.Params) == 64
    assert ctypes.sizeof(_lib.ReadResult) == 56 and _lib.RESULT_DTYPE.itemsize == 56
    assert ctypes.sizeof(_lib.BatchStats) == 64


def test_error_strings():
    for code in range(0, -10, -1):
        assert _lib.strerror(code) and _lib.strerror(code) != "unknown error"
    assert _lib.strerror(-99) == "unknown error"


def test_no_cpu_fallback_without_a_gpu():
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present: the failure path is exercised on the CPU box")
    from nanopore_amd import realign
    with pytest.raises(_lib.NprError) as e:
        realign.Context(0)
    assert e.value.code == _lib.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under nanopore_amd/ or scripts/ may reference it."""
    bad = []
    for base in ("nanopore_amd", "scripts"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".h", ".hip")):
                    src = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|realign_oracle\.h", src, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
