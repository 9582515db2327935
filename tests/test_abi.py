"""The C-ABI library loads and exports every symbol include/rt_mi355x.h declares (no GPU needed)."""

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "rt_mi355x.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    so = ROOT / "raytracer_amd" / "librt_mi355x.so"
    if not so.exists():
        pytest.fail("raytracer_amd/librt_mi355x.so is not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(str(so))
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"


def test_binding_lists_the_same_symbols():
    from raytracer_amd import _lib

    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_renderer_argument_checks_do_not_need_a_gpu():
    from raytracer_amd.renderer import is_positive_int

    assert is_positive_int(3) == 3
    assert is_positive_int(2.9) == 2
    with pytest.raises(TypeError):
        is_positive_int(-1)
    with pytest.raises(TypeError):
        is_positive_int("4")


def test_rt_create_failure_reports_and_leaves_nothing_behind():
    """A failing rt_create returns non-zero, describes the failure and hands back no context (it
    frees whatever it had created); rt_destroy(NULL) is a no-op.  Works with or without a GPU: without
    one the device count query fails, with one device 9999 does not exist."""
    from raytracer_amd import _lib

    L = _lib.load()
    for _ in range(3):
        ctx = ctypes.c_void_p(0xDEAD)
        assert L.rt_create(9999, ctypes.byref(ctx)) != 0
        assert ctx.value is None
        assert len(L.rt_last_error()) > 0
    assert L.rt_create(0, None) != 0
    assert b"out is null" in L.rt_last_error()
    assert L.rt_destroy(None) == 0
    assert L.rt_resize(None, 64, 64) != 0
    assert b"null context" in L.rt_last_error()
