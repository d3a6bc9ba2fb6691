"""The C-ABI library loads without a GPU and exports every symbol include/m3hip.h declares."""
import ctypes as C
import os
import re

import pytest

from mach3_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "m3hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(m3hip_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.M3HIP_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(capi.lib_path("libm3hip.so"))
    for name in declared_symbols():
        assert hasattr(lib, name), f"libm3hip.so does not export {name}"
    capi.load_m3hip()  # argtypes for every symbol resolve
    assert b"gfx950" in capi.load_m3hip().m3hip_version()


def test_create_fails_loudly_without_device():
    """No CPU fallback: on a box without a HIP device m3hip_create must fail, not degrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load_m3hip()
    h = C.c_void_p()
    rc = lib.m3hip_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no CPU fallback" in lib.m3hip_last_error(None)
    from mach3_amd.engine import Engine
    with pytest.raises(capi.M3HipError):
        Engine(0)


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: the header compiles as C99 (plain pointers and sizes, no C++ or torch types)"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "m3hip.h"\nint main(void) { return M3HIP_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
