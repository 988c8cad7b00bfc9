"""The C-ABI library loads (no GPU needed for dlopen) and exports every function
that include/scarlet_amd.h declares; the ctypes table covers the same set."""

import ctypes
import os

import pytest
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "scarlet_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_both_seams():
    names = declared_functions()
    for n in ("smi_prox_weighted_monotonic_f32", "smi_prox_weighted_monotonic_f64",
              "smi_apply_filter_f32", "smi_apply_filter_f64", "smi_batch_create",
              "smi_batch_step", "smi_batch_fit", "smi_batch_forward", "smi_batch_gradient"):
        assert n in names


def test_library_exports_every_declared_symbol():
    from scarlet_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == declared_functions()
    assert lib.smi_version is not None


def test_no_device_is_reported_not_hidden():
    """Without a GPU the compute entry points fail loudly (no CPU fallback)."""
    import numpy as np
    import pytest
    from scarlet_amd import _lib, operator

    lib = _lib.load()
    if lib.smi_device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.arange(25, dtype=np.float64).reshape(5, 5)
    prox = operator.prox_weighted_monotonic((5, 5), "angle", 0, (2, 2))
    with pytest.raises(_lib.ScarletAmdError):
        prox(x, 0)
    from scarlet_amd import BlendBatch, ComponentSpec

    with pytest.raises(_lib.ScarletAmdError):
        BlendBatch(np.zeros((1, 2, 8, 8), np.float32), np.ones((1, 2, 8, 8), np.float32),
                   [[ComponentSpec(np.ones(2), np.ones((3, 3)), (1, 1))]])


def test_product_does_not_import_the_oracle():
    """scarlet_amd/ must never reach into oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "scarlet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text or f.endswith(".md"), f


def test_header_is_plain_c_and_links(tmp_path):
    """include/scarlet_amd.h is a C header (extern "C" ABI, no C++ or torch types): a C99
    translation unit including it compiles with -pedantic and links against the library"""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text(
        '#include "scarlet_amd.h"\n'
        "int main(void) {\n"
        "    smi_batch_desc d; smi_components c; (void)d; (void)c;\n"
        "    return smi_version() == 0;\n"
        "}\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                           "-I", os.path.join(root, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])
