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
    assert _lib.lib.dinox_version() == 3
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
    assert ctypes.sizeof(GemmArgs) == 3 * 8 + 3 * 8 + 3 * 8 + 4 * 8 + 5 * 4 + 4 + 2 * 8 + 8 + 8 + 8 + 8 + 8        # (+ ws, ABI version 2)
    assert GemmArgs.ws.offset == ctypes.sizeof(GemmArgs) - 8
    assert GemmArgs.alpha.offset == 13 * 8 + 5 * 4
    assert GemmArgs.bias.offset == 13 * 8 + 24


def test_block_args_struct_layout(tmp_path):
    """The ctypes mirrors of dinox_block_fwd_args / dinox_block_bwd_args against the C structs themselves: a small C program that
    includes include/dinox.h prints sizeof / offsetof, compiled with the host compiler (a mismatch would hand the kernels garbage
    pointers)."""
    import shutil
    import subprocess
    from dinox._lib import BlockBwdArgs as B, BlockFwdArgs as F
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no host C compiler")
    src = tmp_path / "sz.c"
    src.write_text("#include <stdio.h>\n#include <stddef.h>\n#include \"dinox.h\"\nint main(void){\n"
                   "printf(\"%zu %zu %zu %zu %zu %zu\\n\", sizeof(dinox_block_fwd_args), offsetof(dinox_block_fwd_args, x0), offsetof(dinox_block_fwd_args, eps),"
                   " offsetof(dinox_block_fwd_args, next_eps), offsetof(dinox_block_fwd_args, yn), offsetof(dinox_block_fwd_args, b2));\n"
                   "printf(\"%zu %zu %zu %zu %zu\\n\", sizeof(dinox_block_bwd_args), offsetof(dinox_block_bwd_args, g), offsetof(dinox_block_bwd_args, dwqkv),"
                   " offsetof(dinox_block_bwd_args, g0_lowp), offsetof(dinox_block_bwd_args, tn_ws_bytes));\nreturn 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run([cc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE).stdout.decode().split()
    want = [ctypes.sizeof(F), F.x0.offset, F.eps.offset, F.next_eps.offset, F.yn.offset, F.b2.offset,
            ctypes.sizeof(B), B.g.offset, B.dwqkv.offset, B.g0_lowp.offset, B.tn_ws_bytes.offset]
    assert [int(v) for v in got] == want


def test_gemm_dispatch_by_shape(monkeypatch):
    """dinox_gemm_kernel_name is host logic (shapes, strides, alignment of the pointer VALUES; nothing is dereferenced): the
    hot-path shapes must reach the kernels DESIGN.md section 4 prices them on -- K = 384 NT products the register-prefetch kernel
    with every epilogue it took over, longer reductions the LDS-DMA ring, dW products the split-K TN kernel."""
    from dinox import _lib
    from dinox.ops import BF16, F32, EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_RESIDUAL

    def name(M, N, K, tA=0, tB=0, epi=0, out=BF16, aux=0, res=0):
        g = _lib.GemmArgs(A=0x10000, B=0x20000, C=0x30000, M=M, N=N, K=K, lda=(M if tA else K), ldb=(N if tB else K), ldc=N, batch=1,
                          strideA=0, strideB=0, strideC=M * N, transA=tA, transB=tB, in_dtype=BF16, out_dtype=out, epilogue=epi,
                          alpha=1.0, bias=0x40000 if epi & EPI_BIAS else None, residual=res or None, ldr=N, aux=aux or None, ldaux=N,
                          colsum=None)
        return _lib.lib.dinox_gemm_kernel_name(ctypes.byref(g)).decode()

    monkeypatch.delenv("DINOX_NT_AREG_MAXK", raising=False)
    monkeypatch.setenv("DINOX_NT_PP", "0")                  # the 128 x 128 kernels first (round 3's ping-pong kernels: below)
    T = 512 * 201
    assert name(T, 1152, 384, epi=EPI_BIAS) == "gemm_bf16_nt_areg"                                  # qkv
    assert name(T, 384, 384, epi=EPI_BIAS | EPI_RESIDUAL, out=F32, res=0x50000) == "gemm_bf16_nt_areg"   # proj
    assert name(T, 1536, 384, epi=EPI_BIAS | EPI_GELU, aux=0x60000) == "gemm_bf16_nt_areg"          # fc1
    assert name(T, 1536, 384, epi=EPI_DGELU, aux=0x60000) == "gemm_bf16_nt_areg"                    # GELU' product
    assert name(T, 384, 1536, epi=EPI_BIAS | EPI_RESIDUAL, out=F32, res=0x50000) == "gemm_bf16_nt_glds"  # fc2
    assert name(T, 384, 1152) == "gemm_bf16_nt_glds"                                                # dX of qkv
    assert name(T // 2, 1024, 1024) == "gemm_bf16_nt_glds"                                          # ViT-L width
    assert name(1536, 384, T, tA=1, tB=1, out=F32) == "gemm_bf16_tn_dma"                            # dW1
    monkeypatch.setenv("DINOX_NT_AREG_MAXK", str(1 << 30))
    assert name(T, 384, 1536) == "gemm_bf16_nt_areg" and name(T // 2, 1024, 1024) == "gemm_bf16_nt_glds"   # K % 192 rules
    monkeypatch.setenv("DINOX_NT_AREG_MAXK", "0")
    assert name(T, 1152, 384, epi=EPI_BIAS) == "gemm_bf16_nt_glds"
    # round 3, default policy (csrc/gemm_bf16.hip nt_pp_choice): which hot-path products go to the persistent ping-pong kernels
    from dinox.ops import EPI_AUXGRAD
    monkeypatch.delenv("DINOX_NT_PP", raising=False)
    monkeypatch.delenv("DINOX_NT_AREG_MAXK", raising=False)
    assert name(T, 1152, 384, epi=EPI_BIAS) == "gemm_bf16_nt_pp"                                                          # qkv: 256 x 256 tiles
    assert name(T, 1536, 384, epi=EPI_BIAS | EPI_GELU | EPI_AUXGRAD, aux=0x60000) == "gemm_bf16_nt_pp"                    # fc1: 256 x 256 tiles on a full chip ...
    assert name(T // 4, 1536, 384, epi=EPI_BIAS | EPI_GELU | EPI_AUXGRAD, aux=0x60000) == "gemm_bf16_nt_areg"             # ... the 128 x 128 kernel at bs 64
    assert name(T, 1536, 384, epi=EPI_DGELU | EPI_AUXGRAD, aux=0x60000) == "gemm_bf16_nt_pp"                              # GELU' product
    assert name(T, 384, 1536, epi=EPI_BIAS | EPI_RESIDUAL, out=F32, res=0x50000) == "gemm_bf16_nt_pp128"                  # fc2: 256 x 128 tiles
    assert name(T, 384, 1152) == "gemm_bf16_nt_pp384" and name(T, 384, 1536) == "gemm_bf16_nt_pp384"                      # dX of qkv / fc1: full-row 208 x 384 tiles
    assert name(T, 384, 1536, out=F32) == "gemm_bf16_nt_pp128" and name(4096, 384, 1536) == "gemm_bf16_nt_glds"           # (bf16 out, a chip's worth of rows)
    assert name(T, 384, 384) == "gemm_bf16_nt_pp128" and name(T // 4, 384, 384) == "gemm_bf16_nt_areg"                    # dX of proj: a full chip only
    assert name(T, 384, 384, epi=EPI_BIAS | EPI_RESIDUAL, out=F32, res=0x50000) == "gemm_bf16_nt_areg"                    # proj without the fused LayerNorm
    assert name(T // 2, 4096, 1024, epi=EPI_GELU | EPI_AUXGRAD, aux=0x60000) == "gemm_bf16_nt_pp"                         # ViT-L fc1
    assert name(T // 2, 1024, 4096, epi=EPI_RESIDUAL, out=F32, res=0x50000) == "gemm_bf16_nt_pp"                          # ViT-L fc2
    assert name(804, 1152, 384, epi=EPI_BIAS) == "gemm_bf16_nt_areg"                                                      # less than one round of tiles
    assert name(T, 1152, 384, epi=EPI_BIAS | EPI_GELU, aux=0x60000) == "gemm_bf16_nt_areg"                                # side tensor = pre-activation: not taken
    monkeypatch.setenv("DINOX_NT_PP", "2")
    assert name(804, 1152, 384, epi=EPI_BIAS) == "gemm_bf16_nt_pp128"                                                     # forced (tests)
