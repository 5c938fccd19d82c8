"""The C-ABI libraries load without a GPU and export every function the headers declare."""
import ctypes as C
import os
import re

import pytest

from pathtracer_amd import core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pth?_[a-z0-9_]+)\s*\(", text)))


def test_ptcore_exports_every_declared_symbol():
    names = _declared("ptcore.h")
    assert sorted(names) == sorted(core.CORE_SYMBOLS)
    lib = core.core_lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_pthost_exports_every_declared_symbol():
    names = _declared("pthost.h")
    assert sorted(names) == sorted(core.HOST_SYMBOLS)
    lib = core.host_lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_sizes_match_the_header():
    assert C.sizeof(core.pt_node) == 32
    assert C.sizeof(core.pt_buffer_desc) == 16 and C.sizeof(core.pt_primset_desc) == 64
    assert C.sizeof(core.pt_instance_desc) == 144 and C.sizeof(core.pt_material_desc) == 96
    assert C.sizeof(core.pt_hit) == 24 and core.HIT_DTYPE.itemsize == 24


def test_no_device_means_a_clean_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = core.core_lib().pt_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in core.core_lib().pt_last_error(None) or b"device" in core.core_lib().pt_last_error(None)


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pathtracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "oracle/" not in text, f
