"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares (not gpu)."""

import os
import re

from nerf_sampling_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nerf_sampling_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ns_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.ns_version() >= 1


def test_argument_errors_without_a_gpu():
    """Validation happens before any HIP call, so bad arguments are reported even on a CPU-only box."""
    lib = _lib.load()
    assert lib.ns_raw2outputs(None, None, None, None, 4, 0, 1, None, None, None, None, None, None, None) == -1
    assert b"ns_raw2outputs" in lib.ns_last_error()
    assert lib.ns_sort_rows(None, 1, 4096, None, None) == -1
    assert lib.ns_render_workspace_bytes(1000, 64) > 1000 * 64 * 16
    assert lib.ns_get_rays(0, 0, 1.0, 1.0, 0.0, 0.0, None, 0, 0, 0.0, 1.0, None, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest

    with pytest.raises(_lib.NativeLibraryError):
        _lib.load()
