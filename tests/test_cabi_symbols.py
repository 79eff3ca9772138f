"""The product library: it loads, exports every symbol include/gsi_hip.h declares, and refuses to
run without a GPU (no CPU fallback).  No compute calls here -- CPU only."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "gsi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsi_[a-z_A-Z0-9]+)\s*\(", src)) - {"gsi_randn_fn"})


def test_header_and_binding_table_agree(gsi):
    assert header_symbols() == sorted(gsi._lib.SIGNATURES)


def test_product_library_exports_every_symbol(gsi):
    if not os.path.exists(gsi.LIB_PATH):
        pytest.skip("libgsi_hip.so not built (run __graft_entry__.build())")
    lib = gsi.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.gsi_version() == 100
    assert lib.gsi_backend_name() == b"hip-gfx950"


def test_no_cpu_fallback(gsi):
    """Without a visible gfx950 device a context cannot be created -- the package never degrades."""
    if not os.path.exists(gsi.LIB_PATH):
        pytest.skip("libgsi_hip.so not built")
    import shutil
    import subprocess
    has_gpu = False
    if shutil.which("rocminfo"):
        r = subprocess.run(["rocminfo"], capture_output=True, text=True)
        has_gpu = "gfx950" in r.stdout
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(gsi.GsiError) as ei:
        gsi.Context(0)
    assert ei.value.code == 4


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "geostatinversion.jl_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            txt = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+\S*oracle", txt, flags=re.M), f
            assert "cpuref" not in txt and "libgsi_oracle" not in txt, f
    for f in os.listdir(os.path.join(pkg, "csrc")):
        if os.path.isdir(os.path.join(pkg, "csrc", f)):
            continue
        txt = open(os.path.join(pkg, "csrc", f)).read()
        assert "gsi_oracle" not in txt and "gsio_" not in txt, f
