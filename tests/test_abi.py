"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/amdrec.h declares; the ctypes binding covers exactly that set; struct layouts agree.
No compute call is made (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "amdrec.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amdrec_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("amdrec_build", os.path.join(ROOT, "movie-recommender-demo_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.build()
    return ctypes.CDLL(path)


def test_library_exports_every_declared_symbol(lib):
    names = _declared()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/amdrec.h but not exported"


def test_binding_matches_header(lib):
    from amdrec import _lib
    assert sorted(_lib.exported_symbols()) == _declared()
    assert _lib.load().amdrec_abi_version() == _lib.ABI_VERSION
    src = open(HEADER).read()
    assert int(re.search(r"#define AMDREC_ABI_VERSION (\d+)", src).group(1)) == _lib.ABI_VERSION
    assert int(re.search(r"#define AMDREC_MAX_K (\d+)", src).group(1)) == _lib.MAX_K


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """sizeof/offsetof of the parameter structs as gcc sees include/amdrec.h == ctypes mirror."""
    from amdrec import weights
    c = tmp_path / "layout.c"
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "amdrec.h"\nint main(){'
                 'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(amdrec_tower_params), offsetof(amdrec_tower_params, tables),'
                 'offsetof(amdrec_tower_params, b), sizeof(amdrec_encoder_layer), sizeof(amdrec_ranker_params),'
                 'offsetof(amdrec_ranker_params, layers), offsetof(amdrec_ranker_params, head_b3));return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    T, E, R = weights.TowerParams, weights.EncoderLayer, weights.RankerParams
    assert got == [ctypes.sizeof(T), T.tables.offset, T.b.offset, ctypes.sizeof(E), ctypes.sizeof(R),
                   R.layers.offset, R.head_b3.offset]


def test_missing_library_fails_loudly(monkeypatch):
    from amdrec import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libamdrec.so")
    with pytest.raises(_lib.AmdrecError):
        _lib.load()


def test_cpu_tensors_are_refused_not_silently_computed():
    import torch
    from amdrec import _lib
    with pytest.raises(_lib.AmdrecError):
        _lib.require_gpu(torch.zeros(3), "x")
