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


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """The three argument structs are mirrored by hand in _lib.py: a C program that includes the public header prints sizeof and
    the offset of every field (gcc; the header is plain C), and the ctypes classes must agree field by field -- a field added to
    the header and not to the mirror (or the other way round) fails here, on a CPU-only box, instead of corrupting a launch."""
    import ctypes
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    pairs = {"ns_render_args": _lib.RenderArgs, "ns_hier_args": _lib.HierArgs, "ns_gemm_problem": _lib.GemmProblem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nerf_sampling_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    # every field the header declares is mirrored: the struct bodies name as many members as the ctypes classes
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nerf_sampling_hip.h")).read(), flags=re.S)
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", text, flags=re.S).group(1)
        n_members = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert n_members == len(cls._fields_), (cname, n_members, len(cls._fields_))
