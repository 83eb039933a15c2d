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


def test_host_only_entry_points_validate_arguments_without_a_gpu():
    """Workspace queries and argument validation return before any HIP call: exercised on CPU."""
    import ctypes as C
    from amdrec import _lib
    lib = _lib.load()
    n = C.c_size_t(0)
    # search workspace: grows with nq and k, covers the candidate lists
    assert lib.amdrec_flat_search_workspace(512, 1_000_000, 500, C.byref(n)) == 0
    big = n.value
    assert big >= 512 * 8192 * 8
    assert lib.amdrec_flat_search_workspace(1, 1_000_000, 500, C.byref(n)) == 0 and n.value < big
    assert lib.amdrec_flat_search_workspace(1, 100, 0, C.byref(n)) == -1          # k out of range
    assert b"k=0" in lib.amdrec_last_error()
    assert lib.amdrec_flat_search_workspace(1, 100, _lib.MAX_K + 1, C.byref(n)) == -1
    # search: bad arguments are refused before anything is launched
    assert lib.amdrec_flat_search(None, 10, 256, 250, None, 1, 256, 5, 0, None, None, None, 0, None, None) == -1
    assert b"multiple of 4" in lib.amdrec_last_error()
    assert lib.amdrec_flat_search(None, 10, 256, 256, None, 0, 256, 5, 0, None, None, None, 0, None, None) == 0   # nq = 0
    assert lib.amdrec_flat_search(None, 10, 256, 256, None, 1, 256, 5, 0, None, None, None, 0, None, None) == -1  # nulls
    assert lib.amdrec_l2_normalize(None, 256, None, 256, 0, 256, None) == 0        # rows = 0: nothing to do
    assert lib.amdrec_l2_normalize(None, 256, None, 256, 4, 255, None) == -1
    assert lib.amdrec_select_topk(None, 0, 3, 5, None, 1, 500, 10, None, None, None, None) == -1   # bad task index
    assert lib.amdrec_topk_merge(None, None, 40, 96, 0, 1, 500, None, None, None) == -1            # 40*500 > 16384
    assert lib.amdrec_ivf_select(None, 0, None, 0, 10, None, None, None) == 0                      # nq = 0
    assert lib.amdrec_topk_merge_partial(None, None, 8, 128, 96, 0, 1, 500, None, None, None, None) == -1   # no counter
    assert b"n_inexact" in lib.amdrec_last_error()


def test_model_workspace_queries_follow_the_architecture():
    import ctypes as C
    from amdrec import _lib, synth, weights
    lib = _lib.load()
    user, ad, nnum = synth.demo_dims()
    n = C.c_size_t(0)
    p, _ = weights.pack_tower(synth.two_tower_state(user, ad, nnum, seed=1), "ad_tower", list(ad), 0, "cpu")
    assert lib.amdrec_tower_workspace(C.byref(p), 1000, C.byref(n)) == 0
    assert n.value >= 2 * 1000 * 512 * 4                   # two ping-pong buffers of the widest hidden layer
    p.n_layers = 99
    assert lib.amdrec_tower_workspace(C.byref(p), 1000, C.byref(n)) == -1 and b"n_layers" in lib.amdrec_last_error()
    rp, _, _ = weights.pack_ranker(synth.ranker_state(user, ad, nnum, seed=2), list(user), list(ad), nnum, "cpu")
    assert lib.amdrec_ranker_workspace(C.byref(rp), 500, C.byref(n)) == 0
    assert n.value >= 500 * (3 * 256 + 1024) * 4
    rp.d_model = 512
    assert lib.amdrec_ranker_workspace(C.byref(rp), 500, C.byref(n)) == -1 and b"d_model" in lib.amdrec_last_error()
