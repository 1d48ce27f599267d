"""CPU-side checks of the C-ABI library: it loads, and exports every symbol that
include/obtg.h declares.  No compute call is made (there is no GPU here)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "obtg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(obtg_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built_lib():
    from optimalbeziertrajectorygeneration_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    import ctypes
    from optimalbeziertrajectorygeneration_amd import _capi
    lib = ctypes.CDLL(built_lib)
    declared = _header_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "libobtg_hip.so does not export " + name
    # the Python binding table and the header agree
    assert sorted(_capi.abi_symbol_names()) == declared
    # and so does the library's own list
    lib.obtg_abi_symbols.restype = ctypes.c_void_p
    p = lib.obtg_abi_symbols()
    names, cur = [], b""
    while True:
        ch = ctypes.string_at(p, 1)
        p += 1
        if ch == b"\0":
            if not cur:
                break
            names.append(cur.decode())
            cur = b""
        else:
            cur += ch
    assert sorted(names) == declared


def test_no_gpu_means_loud_failure(built_lib):
    from optimalbeziertrajectorygeneration_amd import _capi
    lib = _capi.load()
    assert lib.obtg_strerror(-3).decode().startswith("no usable")
    if _capi.device_count() == 0:
        with pytest.raises(RuntimeError):
            _capi.Context(2, 2, 5)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import or load it."""
    pkg = os.path.join(REPO, "optimalbeziertrajectorygeneration_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "libobtg_oracle" not in src and "obtg_oracle_" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
