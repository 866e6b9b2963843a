"""CPU checks of the C-ABI boundary: the library loads, exports exactly what include/dinox.h declares,
and the ctypes signature table matches the header prototypes.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dinox.h")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(dinox_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        protos[m.group(1)] = n
    return protos


def test_header_declares_entry_points():
    protos = header_prototypes()
    assert len(protos) >= 25
    for must in ("dinox_gemm", "dinox_layernorm_fwd", "dinox_attention_bwd", "dinox_dino_ce", "dinox_adamw_ema"):
        assert must in protos


def test_library_exports_every_header_symbol():
    from dinox import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_prototypes():
        assert hasattr(lib, name), f"{name} declared in dinox.h but not exported"


def test_ctypes_table_matches_header():
    from dinox import _lib
    protos = header_prototypes()
    assert set(_lib.SIGNATURES) == set(protos)
    for name, (_, args) in _lib.SIGNATURES.items():
        assert len(args) == protos[name], f"{name}: ctypes has {len(args)} args, header has {protos[name]}"


def test_version_and_error_plumbing():
    from dinox import _lib
    assert _lib.lib.dinox_version() == 1
    # argument validation happens on the host before any launch: safe without a GPU
    rc = _lib.lib.dinox_layernorm_fwd(None, None, None, None, None, None, 4, 8, 1e-5, 0, None)
    assert rc == -1 and "null pointer" in _lib.last_error()
    rc = _lib.lib.dinox_gemm(None, None)
    assert rc == -1
    with pytest.raises(RuntimeError, match="null"):
        _lib.check(rc, "dinox_gemm")


def test_gemm_args_struct_layout():
    """sizeof/offsets of the ctypes mirror of dinox_gemm_args follow the C struct (natural alignment)."""
    from dinox._lib import GemmArgs
    assert ctypes.sizeof(GemmArgs) == 3 * 8 + 3 * 8 + 3 * 8 + 4 * 8 + 5 * 4 + 4 + 2 * 8 + 8 + 8 + 8 + 8
    assert GemmArgs.alpha.offset == 13 * 8 + 5 * 4
    assert GemmArgs.bias.offset == 13 * 8 + 24
