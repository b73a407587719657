"""The C-ABI library loads (no GPU needed) and exports every symbol include/psg_hip.h declares;
the ctypes signature table matches the header's parameter counts."""
import os
import re

from openpsg_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "psg_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int|const char\*)\s+(psg_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_header_symbols_are_exported():
    lib = _lib.load()
    decl = _declared()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in psg_hip.h but not exported by libpsg_hip.so"


def test_ctypes_table_matches_header():
    decl = _declared()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in decl, f"{name} bound in _lib.py but not declared in psg_hip.h"
        assert len(argtypes) == decl[name], f"{name}: {len(argtypes)} ctypes args vs {decl[name]} in the header"
    assert set(decl) - set(_lib.SIGNATURES) == {"psg_last_error"}


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        return
    try:
        _lib.ctx(0)
    except _lib.PsgHipError as e:
        assert "no HIP device" in str(e)
    else:
        raise AssertionError("psg_create succeeded without a GPU")


def test_abi_version_of_header_binding_and_library_agree():
    """psg_version() of the built library == PSG_ABI_VERSION of include/psg_hip.h == the binding's constant
    (an integrator built against another header gets an error at load time, not shifted arguments)."""
    m = re.search(r"#define\s+PSG_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert m, "PSG_ABI_VERSION missing from psg_hip.h"
    assert int(m.group(1)) == _lib.PSG_ABI_VERSION == _lib.load().psg_version()
