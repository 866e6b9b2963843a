"""Host-side wrappers of the C ABI: tensor -> pointer plumbing and autograd glue.

Every function here enqueues hand-written HIP kernels from libdinox_hip.so on torch's current
stream.  PyTorch is used for device memory (its caching allocator owns every buffer), streams and
the autograd graph only.  Tensors must live on a CUDA(HIP) device: CPU tensors raise -- the product
has no CPU fallback (the CPU restatement lives in oracle/ and is test infrastructure).

Numeric modes (see include/dinox.h): fp32 "parity" mode and bf16 "throughput" mode.  The mode is
taken from an explicit override (``compute_dtype(...)`` context manager) or, like the reference's
``--amp`` path (scripts/phase5_big_run.py:1716-1717), from the ambient ``torch.autocast`` state.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import BF16, EPI_ACCUM, EPI_AUXGRAD, EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_RESIDUAL, F32, BlockBwdArgs, BlockFwdArgs, GemmArgs, check, lib

Tensor = torch.Tensor
_OVERRIDE: list = []


# ------------------------------------------------------------------------------------------
# mode / plumbing helpers
# ------------------------------------------------------------------------------------------
@contextlib.contextmanager
def compute_dtype(dtype: torch.dtype):
    """Force the numeric mode (torch.float32 or torch.bfloat16) regardless of autocast state."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _OVERRIDE.append(dtype)
    try:
        yield
    finally:
        _OVERRIDE.pop()


def current_dtype() -> torch.dtype:
    if _OVERRIDE:
        return _OVERRIDE[-1]
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt != torch.bfloat16:
            raise RuntimeError(f"the HIP path supports bfloat16 autocast only (got {dt}); use --amp-dtype bfloat16")
        return torch.bfloat16
    return torch.float32


def _code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "dinox: tensors must be on a CUDA/HIP device -- the MI355X kernel library is the only compute "
                "path (no CPU fallback). Move the model and inputs to 'cuda'.")


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


class GemmTimer:
    """Timing of dinox_gemm launches with HIP events recorded on the launch stream, INSIDE the library (dinox_gemm_timer_start / _stop:
    the launches that dinox_block_forward / _backward issue from C are seen too; bench.py runs one over its timed region to price the
    dominant kernel against its roofline).  Keyed by the kernel the dispatcher picks.  Every launch is COUNTED (launches, algorithmic
    flops and bytes); one launch in `every` is TIMED -- an event pair per launch costs the timed region 2.3 % at bs 256 (40.94 vs 40.00 ms:
    two marker packets around each of ~300 launches per step keep the next kernel from starting under the previous one's tail), which a
    1-in-16 sample brings to nothing measurable (40.02 ms).  The pick is a fixed multiplicative hash of the launch counter, so every
    position of the step is reached over a few steps; a family's time is the sum over its shapes of (mean timed duration of the shape)
    x (its launches).  One timer per process; not under graph capture."""

    def __init__(self, every: int = 16) -> None:
        self.every = max(1, int(every))
        self.text = ""

    def start(self) -> None:
        check(lib.dinox_gemm_timer_start(self.every), "dinox_gemm_timer_start")

    def stop(self) -> None:
        """Synchronises the recorded events and collects the per-shape records."""
        buf = C.create_string_buffer(1 << 20)
        n = lib.dinox_gemm_timer_stop(buf, len(buf))
        if n < 0:
            raise RuntimeError("dinox_gemm_timer_stop: record buffer too small")
        self.text += buf.value.decode()

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()
        return False

    @staticmethod
    def algorithmic_bytes(M, N, K, batch, epilogue, in_dtype, out_dtype, has_aux, shared_b) -> float:
        """HBM bytes one launch has to move if every operand and result crosses the memory interface exactly once:
        A (M x K) and B (N x K, once if shared by the batch) in the input dtype, C (M x N) in the output dtype, plus the
        epilogue's extra tensors -- the fp32 residual read, the GELU' side tensor written (GELU) or read (DGELU), C read back
        under ACCUM."""
        isz = 2 if in_dtype == BF16 else 4
        osz = 2 if out_dtype == BF16 else 4
        mn = float(M) * N * batch
        b = float(M) * K * batch * isz + float(N) * K * isz * (1 if shared_b else batch) + mn * osz
        if epilogue & EPI_RESIDUAL:
            b += mn * 4
        if has_aux and (epilogue & (EPI_GELU | EPI_DGELU)):
            b += mn * osz
        if epilogue & EPI_ACCUM:
            b += mn * 4
        return b

    def summary(self) -> dict:
        """{kernel: {"launches", "timed", "flops", "bytes", "ms"}} -- after stop().  "ms" is the family's time over ALL its launches,
        estimated from the timed ones (see the class docstring)."""
        out: dict = {}
        untimed = []
        for line in self.text.splitlines():
            f = line.split()
            if len(f) != 13:
                continue
            name = f[0]
            M, N, K, batch, epi, idt, odt, aux, shb, launches, timed = (int(v) for v in f[1:12])
            ms = float(f[12])
            fl = 2.0 * M * N * K * batch * launches
            by = self.algorithmic_bytes(M, N, K, batch, epi, idt, odt, aux, shb) * launches
            d = out.setdefault(name, {"launches": 0, "timed": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "_tb": 0.0, "_tms": 0.0})
            d["launches"] += launches
            d["flops"] += fl
            d["bytes"] += by
            if timed:
                d["timed"] += timed
                d["ms"] += ms / timed * launches
                d["_tb"] += by / launches * timed
                d["_tms"] += ms
            else:
                untimed.append((name, by))
        for name, by in untimed:          # a shape the sample never reached: the family's measured time per algorithmic byte
            d = out[name]
            if d["_tb"] > 0:
                d["ms"] += by * d["_tms"] / d["_tb"]
        for name in [k for k, d in out.items() if d["timed"] == 0]:
            del out[name]
        for d in out.values():
            del d["_tb"], d["_tms"]
        self.text = ""
        return out


TRACE_KERNELS: Optional[list] = None      # tests set this to a list to learn which GEMM kernel each call used


# ------------------------------------------------------------------------------------------
# raw ops (no autograd)
# ------------------------------------------------------------------------------------------
def gemm(A: Tensor, B: Tensor, *, transA=False, transB=False, out: Optional[Tensor] = None,
         out_dtype: Optional[torch.dtype] = None, bias: Optional[Tensor] = None, gelu=False, aux: Optional[Tensor] = None,
         dgelu=False, residual: Optional[Tensor] = None, accumulate=False, alpha: float = 1.0,
         colsum_out: Optional[Tensor] = None, auxgrad=False) -> Tensor:
    """C = epilogue(alpha * op(A) op(B)^T); A: [M,K] (or [K,M] if transA), B: [N,K] (or [K,N] if transB).
    3-D operands are batched over dim 0 (B may be 2-D = shared)."""
    _need_cuda(A, B)
    assert A.dtype == B.dtype, (A.dtype, B.dtype)
    A, B = _c(A), _c(B)
    batched = A.dim() == 3
    if batched:
        batch = A.shape[0]
        a2, b2 = A.shape[1:], (B.shape[1:] if B.dim() == 3 else B.shape)
    else:
        batch, a2, b2 = 1, A.shape, B.shape
    M, K = (a2[1], a2[0]) if transA else (a2[0], a2[1])
    N, Kb = (b2[1], b2[0]) if transB else (b2[0], b2[1])
    assert K == Kb, f"inner dims differ: {tuple(A.shape)} vs {tuple(B.shape)} (transA={transA}, transB={transB})"
    odt = out_dtype or (out.dtype if out is not None else A.dtype)
    if out is None:
        out = torch.empty((batch, M, N) if batched else (M, N), dtype=odt, device=A.device)
    else:
        assert out.is_contiguous() and out.dtype == odt and out.numel() == batch * M * N
    epi = 0
    if bias is not None:
        epi |= EPI_BIAS
        assert bias.dtype == torch.float32 and bias.numel() == N
    if gelu:
        epi |= EPI_GELU
    if dgelu:
        epi |= EPI_DGELU
        assert aux is not None
    if aux is not None:
        assert aux.dtype == odt and aux.is_contiguous() and aux.numel() == batch * M * N
    if residual is not None:
        epi |= EPI_RESIDUAL
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.numel() == batch * M * N
    if accumulate:
        epi |= EPI_ACCUM
    if auxgrad:
        epi |= EPI_AUXGRAD
    if colsum_out is not None:
        assert transA and not batched and colsum_out.dtype == torch.float32 and colsum_out.numel() == M and colsum_out.is_contiguous()
    g = GemmArgs(
        A=_p(A), B=_p(B), C=_p(out), M=M, N=N, K=K, lda=a2[1], ldb=b2[1], ldc=N, batch=batch,
        strideA=a2[0] * a2[1] if batched else 0, strideB=(b2[0] * b2[1] if (batched and B.dim() == 3) else 0),
        strideC=M * N, transA=int(transA), transB=int(transB), in_dtype=_code(A.dtype), out_dtype=_code(odt),
        epilogue=epi, alpha=alpha, bias=_p(bias), residual=_p(residual), ldr=N, aux=_p(aux), ldaux=N, colsum=_p(colsum_out), ws=None)
    if (transA and transB and not batched and odt == torch.float32 and K >= 4096 and not (epi & ~EPI_ACCUM)
            and ((M + 127) // 128) * ((N + 127) // 128) < 256 and lib.dinox_gemm_kernel_name(C.byref(g)).decode().startswith("gemm_f32")):
        # dW products that land on the general exact-fp32 kernel -- the whole fp32 mode, and bf16 operands whose shape the MFMA bf16 kernels
        # do not take (patch 14: 3 x 14 x 14 = 588 columns) -- are a few dozen tiles with the whole token axis as reduction: 36-60 workgroups
        # on 256 CUs, 3-12 ms per launch.  Cut the reduction into chunks that run as ONE batched launch and meet in the fixed-order column sum.
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        splits = max(1, min(K // 1024, 1024 // tiles))
        while splits > 1 and K % splits:
            splits -= 1
        if splits > 1:
            kc = K // splits
            part = torch.empty((splits, M, N), dtype=torch.float32, device=A.device)
            g2 = GemmArgs(A=_p(A), B=_p(B), C=_p(part), M=M, N=N, K=kc, lda=a2[1], ldb=b2[1], ldc=N, batch=splits, strideA=kc * a2[1],
                          strideB=kc * b2[1], strideC=M * N, transA=1, transB=1, in_dtype=_code(A.dtype), out_dtype=F32, epilogue=0, alpha=alpha,
                          bias=None, residual=None, ldr=N, aux=None, ldaux=N, colsum=None, ws=None)
            if TRACE_KERNELS is not None:
                TRACE_KERNELS.append(lib.dinox_gemm_kernel_name(C.byref(g2)).decode())
            check(lib.dinox_gemm(C.byref(g2), _stream()), "dinox_gemm")
            colsum(part.view(splits, M * N), out=out.view(-1), accumulate=accumulate)
            if colsum_out is not None:                       # A is stored [K][M]: its column sums are the bias gradient
                check(lib.dinox_colsum(_p(A), _p(colsum_out), K, M, a2[1], _code(A.dtype), int(accumulate), _stream()), "dinox_colsum")
            return out
    if transA and transB and not batched and A.dtype == torch.bfloat16 and not _TN_ATOMICS:
        need = lib.dinox_gemm_ws_bytes(C.byref(g))       # split-K dW product: deterministic two-stage reduction through a workspace
        if need:
            g.ws = _p(_tn_workspace(need, A.device))
    if TRACE_KERNELS is not None:
        TRACE_KERNELS.append(lib.dinox_gemm_kernel_name(C.byref(g)).decode())
    check(lib.dinox_gemm(C.byref(g), _stream()), "dinox_gemm")
    return out


_TN_ATOMICS = bool(os.environ.get("DINOX_TN_ATOMICS"))      # A/B: split-K dW products through fp32 atomics (round-1 behaviour, not reproducible)
_TN_WS: dict = {}


def _tn_workspace(nbytes: int, device) -> Tensor:
    """One growable scratch buffer per (device, stream): consecutive split-K products of a stream reuse it in stream order."""
    key = (device, _stream())
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _TN_WS[key] = torch.empty(max(nbytes, 40 << 20), dtype=torch.uint8, device=device)
    return ws


def gemm_nt_f32_splitk(A: Tensor, B: Tensor, splits: int = 0) -> Tensor:
    """A [M,K] . B [N,K]^T in exact fp32 with the reduction cut into K-chunks that run as ONE batched launch (operand strides: chunk c
    starts K/splits elements further along every row) and meet in a fixed-order column sum.  For few-tile, long-K products: the KoLeo
    inner products are 512 x 512 x 8192 = 16 tiles, i.e. 16 workgroups on 256 CUs and 784 us as a plain product."""
    _need_cuda(A, B)
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.is_contiguous() and B.is_contiguous() and A.dim() == 2 and B.dim() == 2
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K
    if splits <= 0:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        splits = max(1, min(K // 256, 256 // max(tiles, 1)))
    while splits > 1 and K % splits:
        splits -= 1
    if splits == 1:
        return gemm(A, B, out_dtype=torch.float32)
    kc = K // splits
    part = torch.empty((splits, M, N), dtype=torch.float32, device=A.device)
    g = GemmArgs(A=_p(A), B=_p(B), C=_p(part), M=M, N=N, K=kc, lda=K, ldb=K, ldc=N, batch=splits, strideA=kc, strideB=kc, strideC=M * N,
                 transA=0, transB=0, in_dtype=F32, out_dtype=F32, epilogue=0, alpha=1.0, bias=None, residual=None, ldr=N, aux=None, ldaux=N,
                 colsum=None, ws=None)
    if TRACE_KERNELS is not None:
        TRACE_KERNELS.append(lib.dinox_gemm_kernel_name(C.byref(g)).decode())
    check(lib.dinox_gemm(C.byref(g), _stream()), "dinox_gemm")
    return colsum(part.view(splits, M * N)).view(M, N)


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate=False) -> Tensor:
    x2 = _c(x).reshape(-1, x.shape[-1])
    if out is None:
        out = torch.empty(x2.shape[1], dtype=torch.float32, device=x.device)
    check(lib.dinox_colsum(_p(x2), _p(out), x2.shape[0], x2.shape[1], x2.shape[1], _code(x.dtype), int(accumulate), _stream()), "dinox_colsum")
    return out


def cast_bf16(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    x = _c(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    else:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == x.numel()
    check(lib.dinox_cast_bf16(_p(x), _p(out), x.numel(), _stream()), "dinox_cast_bf16")
    return out


def cast_transpose_bf16(w: Tensor) -> Tensor:
    w = _c(w)
    R, Cc = w.shape
    out = torch.empty((Cc, R), dtype=torch.bfloat16, device=w.device)
    check(lib.dinox_cast_transpose_bf16(_p(w), _p(out), R, Cc, _stream()), "dinox_cast_transpose_bf16")
    return out


def to_mode(x: Tensor, dt: torch.dtype) -> Tensor:
    """Bring an activation to the GEMM operand dtype of the current mode."""
    if x.dtype == dt:
        return _c(x)
    if dt == torch.bfloat16 and x.dtype == torch.float32:
        return cast_bf16(x)
    if dt == torch.float32 and x.dtype == torch.bfloat16:
        return x.float()
    raise TypeError(f"cannot bring {x.dtype} to {dt}")


class ArenaShadow:
    """bf16 image of a whole flat fp32 parameter arena, refreshed by ONE cast launch after an optimiser step, plus transposed
    images (one launch for all of them) of the matrices a backward pass has asked for.  Replaces ~150 per-weight cast launches
    per ViT-S step.  A parameter whose version moved since the refresh (load_state_dict, a manual copy_) is not served."""

    def __init__(self, flat: Tensor, params, offsets) -> None:
        self.flat, self.params, self.offsets = flat, list(params), list(offsets)
        self.index = {p.data_ptr(): i for i, p in enumerate(self.params) if p.numel()}
        self.plain: Optional[Tensor] = None
        self.trans: Optional[Tensor] = None
        self.versions: list = []
        self.wanted: dict = {}          # parameter index -> (R, C) of matrices needed transposed
        self.table: Optional[Tensor] = None
        self.table_for: tuple = ()
        self.tiles = 0
        self.served_t: set = set()

    def refresh(self) -> None:
        # (ONE persistent image, rewritten in place: the operand views handed out earlier stay valid, and a step captured into a
        # hipGraph reads at replay n + 1 what replay n wrote)
        self.plain = cast_bf16(self.flat, out=self.plain)
        self.versions = [p._version for p in self.params]
        self.served_t = set()
        if self.wanted:
            key = tuple(sorted(self.wanted))
            if key != self.table_for:
                rows, tiles = [], 0
                for i in key:
                    R, Cc = self.wanted[i]
                    rows.append([self.offsets[i], R, Cc, tiles])
                    tiles += ((R + 31) // 32) * ((Cc + 31) // 32)
                self.table = torch.tensor(rows, dtype=torch.int64, device=self.flat.device)
                self.table_for, self.tiles = key, tiles
            if self.trans is None:
                self.trans = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            check(lib.dinox_cast_transpose_bf16_multi(_p(self.flat), _p(self.trans), _p(self.table), len(self.table_for), self.tiles,
                                                      _stream()), "dinox_cast_transpose_bf16_multi")
            self.served_t = set(self.table_for)

    def get(self, w: Tensor, transposed: bool) -> Optional[Tensor]:
        i = self.index.get(w.data_ptr())
        if i is None or self.params[i].shape != w.shape:
            return None
        R = w.shape[0]
        Cc = w.numel() // R
        if transposed:
            self.wanted.setdefault(i, (R, Cc))
        if self.plain is None or self.versions[i] != w._version:
            return None
        off = self.offsets[i]
        if not transposed:
            return self.plain[off:off + R * Cc].view(R, Cc)
        if i not in self.served_t:
            return None
        return self.trans[off:off + R * Cc].view(Cc, R)


class _WeightCache:
    """bf16 (and transposed bf16) copies of fp32 master weights.  Arena shadows registered by the training engine serve whole
    models from one cast launch; anything else is cast per weight and cached so the student forward, its backward and repeated
    teacher forwards of one step share one cast.

    An entry is valid only for the very tensor OBJECT it was made from (held by weak reference and compared with ``is``) at
    the version it had then: a freed model whose successor lands on the same addresses at version 0 (two ``load_model`` calls
    in a row) can therefore never be served the old model's weights.  Writers that go around torch's version counter (the
    ``dinox_adamw_ema`` kernel, any C-ABI caller writing through raw pointers) must call ``invalidate_weights()``."""

    def __init__(self) -> None:
        self.d: dict = {}
        self.extra: dict = {}          # other per-step images of a weight (the zero-padded patch-embedding operand), same lifetime
        self.shadows: list = []

    def clear(self) -> None:
        self.d.clear()
        self.extra.clear()

    def get(self, w: Tensor, transposed: bool) -> Tensor:
        for sh in self.shadows:
            hit = sh.get(w, transposed)
            if hit is not None:
                return hit
        key = (id(w), transposed)
        hit = self.d.get(key)
        if hit is not None and hit[0]() is w and hit[1] == (w.data_ptr(), w._version, tuple(w.shape)):
            return hit[2]
        if len(self.d) > 4096:
            self.d.clear()
        w2 = w.detach().reshape(w.shape[0], -1)
        val = cast_transpose_bf16(w2) if transposed else cast_bf16(w2)
        d = self.d
        self.d[key] = (weakref.ref(w, lambda _r, key=key, d=d: d.pop(key, None)), (w.data_ptr(), w._version, tuple(w.shape)), val)
        return val


weight_cache = _WeightCache()


def invalidate_weights() -> None:
    """Drop every cached bf16 weight image.  For callers that modify master weights through raw device pointers (the C ABI's
    ``dinox_adamw_ema``, a custom optimiser): torch's version counter does not see such writes."""
    weight_cache.clear()


def weight_operand(w: Tensor, dt: torch.dtype, transposed=False) -> Tensor:
    """W as [N,K] (or W^T as [K,N] when transposed) in the GEMM operand dtype."""
    w2 = w.detach().reshape(w.shape[0], -1)
    if dt == torch.float32:
        return w2 if not transposed else w2  # fp32 kernels take either layout through the trans flags
    return weight_cache.get(w, transposed)


def layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, out_dtype: torch.dtype, eps: float = 1e-5):
    _need_cuda(x, w, b)
    assert x.dtype == torch.float32
    x = _c(x)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.dinox_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, D, eps, _code(out_dtype), _stream()), "dinox_layernorm_fwd")
    return y, mean, rstd


_ROWLN_PP = os.environ.get("DINOX_ROWLN_PP")    # "0": never the full-row kernel's LayerNorm epilogue (the library reads it per call too)
_ROWLN = os.environ.get("DINOX_ROWLN")          # "1": every width-384 product, "0": none, unset: the short reductions (proj) only


def rowln_ok(M: int, N: int, K: int, dt: torch.dtype, y_dtype: Optional[torch.dtype] = None) -> bool:
    """Should this product + the LayerNorm behind it run as ONE launch (bf16 mode, N = 384)?  Measured on MI355X at the hot-path shape
    (M = 102 912):
      * csrc/gemm_bf16_pp384.hip's LayerNorm epilogue (bf16 y, M >= 40000 = about a round of its 208 x 384 tiles; DINOX_ROWLN_PP=0 disables):
        proj + LN 103 us against 85 + 44 us for the two launches; fc2 + LN 206 us against 177 + 44 -> both fused;
      * csrc/gemm_bf16_rowln.hip (128 x 384 tiles: fp32 y, small M): proj + LN 130 us against 85 + 44 -> fused for short reductions
        (K <= 576); fc2 + LN 256 us against 177 + 44 -> not fused unless DINOX_ROWLN=1."""
    if dt != torch.bfloat16 or _ROWLN == "0" or not lib.dinox_linear_residual_ln_ok(M, N, K):
        return False
    full_row = _ROWLN_PP != "0" and M >= 40000 and (y_dtype or dt) == torch.bfloat16 and K >= 128     # (the library's rule)
    return _ROWLN == "1" or K <= 576 or (full_row and os.environ.get("DINOX_ROWLN_FC2") != "0")


def linear_residual_ln(a: Tensor, w: Tensor, bias: Optional[Tensor], residual: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float,
                       y_dtype: torch.dtype):
    """x = residual + a W^T + bias (fp32) and y = LayerNorm(x) (y_dtype) with its row statistics, one launch.
    a [M,K] bf16, w [N,K] bf16.  Returns (x, y, mean, rstd)."""
    _need_cuda(a, w)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.is_contiguous() and w.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and (residual is None or (residual.dtype == torch.float32 and residual.is_contiguous() and residual.numel() == M * N))
    dev = a.device
    x = torch.empty((M, N), dtype=torch.float32, device=dev)
    y = torch.empty((M, N), dtype=y_dtype, device=dev)
    mean = torch.empty(M, dtype=torch.float32, device=dev)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    if TRACE_KERNELS is not None:
        TRACE_KERNELS.append("gemm_bf16_rowln")
    check(lib.dinox_linear_residual_ln(_p(a), _p(w), _p(bias), _p(residual), _p(x), _p(_c(gamma)), _p(_c(beta)), eps, _p(y), _code(y_dtype),
                                       _p(mean), _p(rstd), M, N, K, _stream()), "dinox_linear_residual_ln")
    return x, y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dx: Optional[Tensor] = None,
                  dx_add: Optional[Tensor] = None, want_lowp=False, b: Optional[Tensor] = None):
    """dx = (dx_add or 0) + LN'(dy); returns (dx, dw, db, bf16 copy of dx or None).  dx may alias dx_add.
    With the bias parameter ``b`` given and both affine parameters registered in the gradient sink, dw/db are added
    straight into the gradient arena and returned as None."""
    dy, x = _c(dy), _c(x)
    D = x.shape[-1]
    rows = x.numel() // D
    if dx is None:
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if dx_add is not None:
        assert dx_add.is_contiguous() and dx_add.dtype == torch.float32 and dx_add.numel() == x.numel()
    sw, sb = (grad_sink.lookup(w), grad_sink.lookup(b)) if b is not None else (None, None)
    sunk = sw is not None and sb is not None
    dw = sw[1].grad if sunk else torch.empty(D, dtype=torch.float32, device=x.device)
    db = sb[1].grad if sunk else torch.empty(D, dtype=torch.float32, device=x.device)
    lowp = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_lowp else None
    ws = torch.empty(lib.dinox_layernorm_bwd_ws_bytes(rows, D), dtype=torch.uint8, device=x.device)
    check(lib.dinox_layernorm_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dx_add), _p(lowp), _p(dw), _p(db), _p(ws),
                                  rows, D, _code(dy.dtype), int(sunk), _stream()), "dinox_layernorm_bwd")
    if sunk:
        grad_sink.ready(sw)
        grad_sink.ready(sb)
        return dx, None, None, lowp
    return dx, dw, db, lowp


_LNBWD_PP = os.environ.get("DINOX_LNBWD_PP")    # "0": never fuse the dX product with the LayerNorm backward behind it; "1": wherever the kernel applies


def linear_ln_bwd_ok(M: int, D: int, K: int, dt: torch.dtype) -> bool:
    """Should the input-gradient product into a LayerNorm and that LayerNorm's backward run as ONE launch (csrc/gemm_bf16_pp384.hip's
    LayerNorm-backward epilogue; bf16 mode, width 384)?  Unset: on a chip's worth of rows (M >= 8192); the results equal the two launches' to the last bit."""
    if dt != torch.bfloat16 or _LNBWD_PP == "0" or not lib.dinox_linear_ln_bwd_ok(M, D, K):
        return False
    return _LNBWD_PP == "1" or M >= 8192


def linear_ln_bwd(a: Tensor, w_t: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dx: Optional[Tensor] = None,
                  dx_add: Optional[Tensor] = None, want_lowp=False, b: Optional[Tensor] = None):
    """dy = a w_t^T (bf16), then layernorm_bwd(dy, x, w, mean, rstd, ...) in the same launch: returns (dx, dw, db, bf16 copy of dx or None)
    exactly as layernorm_bwd does.  a [M,K] bf16, w_t [D,K] bf16 (the W^T operand of the Linear in front of the LayerNorm)."""
    a, x = _c(a), _c(x)
    M, K = a.shape
    D = x.shape[-1]
    assert w_t.shape == (D, K) and w_t.is_contiguous() and x.numel() == M * D
    if dx is None:
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if dx_add is not None:
        assert dx_add.is_contiguous() and dx_add.dtype == torch.float32 and dx_add.numel() == x.numel()
    sw, sb = (grad_sink.lookup(w), grad_sink.lookup(b)) if b is not None else (None, None)
    sunk = sw is not None and sb is not None
    dw = sw[1].grad if sunk else torch.empty(D, dtype=torch.float32, device=x.device)
    db = sb[1].grad if sunk else torch.empty(D, dtype=torch.float32, device=x.device)
    lowp = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_lowp else None
    ws = torch.empty(lib.dinox_layernorm_bwd_ws_bytes(M, D), dtype=torch.uint8, device=x.device)
    if TRACE_KERNELS is not None:
        TRACE_KERNELS.append("gemm_bf16_nt_pp384(ln_bwd)")
    check(lib.dinox_linear_ln_bwd(_p(a), _p(w_t), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dx_add), _p(lowp), _p(dw), _p(db), _p(ws),
                                  M, D, K, int(sunk), _stream()), "dinox_linear_ln_bwd")
    if sunk:
        grad_sink.ready(sw)
        grad_sink.ready(sb)
        return dx, None, None, lowp
    return dx, dw, db, lowp


_ATTN_F32_REF = bool(os.environ.get("DINOX_ATTN_F32_REF"))      # fp32 attention by the per-lane reference kernels whatever the size (A/B, tests)


def _gemm_f32_raw(a: int, b: int, c: int, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, batch: int, sa: int, sb: int, sc: int,
                  ta: bool, tb: bool, alpha: float = 1.0) -> None:
    """One batched exact-fp32 product on raw addresses and element strides (operands are slices of packed tensors: no copies)."""
    g = GemmArgs(A=a, B=b, C=c, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, batch=batch, strideA=sa, strideB=sb, strideC=sc,
                 transA=int(ta), transB=int(tb), in_dtype=F32, out_dtype=F32, epilogue=0, alpha=alpha, bias=None, residual=None, ldr=N,
                 aux=None, ldaux=N, colsum=None, ws=None)
    check(lib.dinox_gemm(C.byref(g), _stream()), "dinox_gemm")


def _use_f32_products(qkv: Tensor, N: int, d: int) -> bool:
    # full-size fp32 attention: batched exact-fp32 products + softmax rows (the per-lane reference kernels take 4.3 / 32 ms per layer
    # forward / backward at bs 64; this form 0.7 / 1.7 ms).  Small problems keep the reference kernels (fewer launches).
    return qkv.dtype == torch.float32 and not _ATTN_F32_REF and N >= 32 and qkv.shape[0] * N >= 4096


def _bf16_attention_needs_products(qkv: Tensor, N: int, d: int, fwd: bool) -> bool:
    """bf16 mode, a head size the MFMA attention kernels do not take (not a multiple of 8, or above 128), big enough to matter.  Since round 3
    every other shape runs on the device kernels: head size 64 up to 288 / 544 tokens on the whole-strip kernels (csrc/attention_bf16.hip),
    longer sequences and head sizes up to 128 -- ViT-g: 88 -- on the tiled ones (csrc/attention_flash.hip)."""
    if qkv.dtype != torch.bfloat16 or _ATTN_F32_REF or N < 32 or qkv.shape[0] * N < 4096:
        return False
    return d % 8 != 0 or d > 128


_PRODUCTS_BUDGET = 256 << 20      # bytes of fp32 scores per buffer of the product form (S, and dP in the backward)


def _product_chunks(B: int, N: int):
    """Batch ranges of the fp32 product form such that a [b, N, N] fp32 score buffer stays within _PRODUCTS_BUDGET (at bs 256
    and 785 tokens the un-chunked buffers were 1.26 GB each)."""
    per = max(1, _PRODUCTS_BUDGET // (4 * N * N))
    return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]


def _attention_fwd_f32_products(qkv: Tensor, heads: int, o: Tensor, lse: Tensor) -> None:
    """softmax(Q K^T / sqrt(d)) V per head as S = scale Q K^T (NT), softmax rows, O = P V -- operands addressed inside the packed qkv
    [B, N, 3, heads, d] and o [B, N, heads, d] by leading dimensions and batch strides; the batch is walked in chunks that keep the
    score buffer within _PRODUCTS_BUDGET."""
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    d = Cc // heads
    scale = 1.0 / math.sqrt(d)
    chunks = _product_chunks(B, N)
    S = torch.empty((chunks[0][1] - chunks[0][0], N, N), dtype=torch.float32, device=qkv.device)
    E = 4
    for b0, b1 in chunks:
        nb = b1 - b0
        q0, o0 = qkv.data_ptr() + E * b0 * N * C3, o.data_ptr() + E * b0 * N * Cc
        for h in range(heads):
            qh, kh, vh = q0 + E * h * d, q0 + E * (Cc + h * d), q0 + E * (2 * Cc + h * d)
            _gemm_f32_raw(qh, kh, _p(S), N, N, d, C3, C3, N, nb, N * C3, N * C3, N * N, False, False, scale)          # S[b] = scale Q K^T
            check(lib.dinox_softmax_rows(_p(S), lse.data_ptr() + E * (b0 * heads + h) * N, nb * N, N, N, heads * N, _stream()), "dinox_softmax_rows")
            _gemm_f32_raw(_p(S), vh, o0 + E * h * d, N, d, N, N, C3, Cc, nb, N * N, N * C3, N * Cc, False, True)      # O[b] = P V


def _attention_bwd_f32_products(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, heads: int, dqkv: Tensor) -> None:
    """Backward of the same: P = exp(S - lse) recomputed per head, dP = dO V^T, dS = P o (dP - rowsum(P o dP)) * scale, then
    dV = P^T dO, dQ = dS K, dK = dS^T Q written straight into the packed dqkv (batch chunks as in the forward)."""
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    d = Cc // heads
    scale = 1.0 / math.sqrt(d)
    dev = qkv.device
    chunks = _product_chunks(B, N)
    nb0 = chunks[0][1] - chunks[0][0]
    S = torch.empty((nb0, N, N), dtype=torch.float32, device=dev)
    dP = torch.empty((nb0, N, N), dtype=torch.float32, device=dev)
    E = 4
    for b0, b1 in chunks:
        nb = b1 - b0
        q0, g0, do0 = qkv.data_ptr() + E * b0 * N * C3, dqkv.data_ptr() + E * b0 * N * C3, do.data_ptr() + E * b0 * N * Cc
        for h in range(heads):
            qh, kh, vh = q0 + E * h * d, q0 + E * (Cc + h * d), q0 + E * (2 * Cc + h * d)
            dqh, dkh, dvh = g0 + E * h * d, g0 + E * (Cc + h * d), g0 + E * (2 * Cc + h * d)
            doh = do0 + E * h * d
            _gemm_f32_raw(qh, kh, _p(S), N, N, d, C3, C3, N, nb, N * C3, N * C3, N * N, False, False, scale)          # S = scale Q K^T
            _gemm_f32_raw(doh, vh, _p(dP), N, N, d, Cc, C3, N, nb, N * Cc, N * C3, N * N, False, False)               # dP = dO V^T
            check(lib.dinox_softmax_bwd_rows(_p(S), _p(dP), lse.data_ptr() + E * (b0 * heads + h) * N, scale, nb * N, N, N, heads * N, _stream()),
                  "dinox_softmax_bwd_rows")                                                                              # S <- P, dP <- dS
            _gemm_f32_raw(_p(S), doh, dvh, N, d, N, N, Cc, C3, nb, N * N, N * Cc, N * C3, True, True)                  # dV = P^T dO
            _gemm_f32_raw(_p(dP), kh, dqh, N, d, N, N, C3, C3, nb, N * N, N * C3, N * C3, False, True)                 # dQ = dS K
            _gemm_f32_raw(_p(dP), qh, dkh, N, d, N, N, C3, C3, nb, N * N, N * C3, N * C3, True, True)                  # dK = dS^T Q


def attention_fwd(qkv: Tensor, heads: int):
    _need_cuda(qkv)
    qkv = _c(qkv)
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    d = Cc // heads
    o = torch.empty((B, N, Cc), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
    if _use_f32_products(qkv, N, d):
        _attention_fwd_f32_products(qkv, heads, o, lse)
        return o, lse
    if _bf16_attention_needs_products(qkv, N, d, fwd=True):
        # head sizes / sequence lengths the MFMA attention kernels do not take (ViT-g: 88 per head; > 288 tokens): the exact-fp32 product
        # form on a float copy instead of the per-lane reference kernels (which are two orders of magnitude slower at these sizes)
        q32 = qkv.float()
        o32 = torch.empty((B, N, Cc), dtype=torch.float32, device=qkv.device)
        _attention_fwd_f32_products(q32, heads, o32, lse)
        return o32.to(qkv.dtype), lse
    check(lib.dinox_attention_fwd(_p(qkv), _p(o), _p(lse), B, N, heads, d, _code(qkv.dtype), _stream()), "dinox_attention_fwd")
    return o, lse


# "1": no-grad blocks (teacher, encode()) run qkv projection + attention as one launch.  Off by default: measured no faster than the two
# launches at ViT-S (219-227 us against 208-222) and slower at ViT-L (500 against 408 us); csrc/attention_bf16.hip, DESIGN.md section 4
_QKV_FUSED = {"1": True, "0": False}.get(os.environ.get("DINOX_QKV_FUSED", ""))      # None: by width (below)


def _qkv_fused(D: int) -> bool:
    """No-grad blocks (the teacher of a step, encode()) run qkv projection + attention as ONE launch?  In isolation a tie at ViT-S and slower
    at ViT-L; in the step, where the teacher's qkv tensor is 237 MB written and read back between two kernels that
    each start cold, 0.2 ms per step faster at width 384 (35.01 -> 34.80 ms, interleaved on one box).  Unset: widths up to 512."""
    return _QKV_FUSED if _QKV_FUSED is not None else D <= 512


def qkv_attention_ok(B: int, N: int, heads: int, D: int, C: int) -> bool:
    """Is (B images of N tokens, width D in, heads x 64 out) inside the fused qkv-projection + attention kernel?"""
    return C % heads == 0 and bool(lib.dinox_qkv_attention_ok(B, N, heads, C // heads, D))


def qkv_attention(x: Tensor, w: Tensor, bias: Optional[Tensor], heads: int, want_qkv: bool = False, want_lse: bool = False):
    """o = attention(x w^T + bias) in ONE launch (dinox_qkv_attention_fwd): x [B,N,D] bf16, w [3C,D] bf16, bias [3C] fp32 or None
    -> o [B,N,C] bf16; with want_qkv / want_lse also the packed qkv rows [B,N,3C] / the log-sum-exp [B,heads,N] (else None).
    Replaces Attention.qkv + the attention core (zoo/arch.py:46-52) for passes that keep nothing for a backward."""
    _need_cuda(x, w)
    x, w = _c(x), _c(w)
    B, N, D = x.shape
    C = w.shape[0] // 3
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.shape == (3 * C, D) and C % heads == 0
    o = torch.empty((B, N, C), dtype=torch.bfloat16, device=x.device)
    qkv = torch.empty((B, N, 3 * C), dtype=torch.bfloat16, device=x.device) if want_qkv else None
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=x.device) if want_lse else None
    if TRACE_KERNELS is not None:
        TRACE_KERNELS.append("attn_qkv_fused_fwd")
    check(lib.dinox_qkv_attention_fwd(_p(x), _p(w), _p(None if bias is None else _c(bias)), _p(o), _p(qkv), _p(lse), B, N, heads, C // heads, D,
                                      _stream()), "dinox_qkv_attention_fwd")
    return o, qkv, lse


def attention_bwd(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, heads: int) -> Tensor:
    do = to_mode(do, qkv.dtype)
    B, N, C3 = qkv.shape
    d = C3 // 3 // heads
    dqkv = torch.empty_like(qkv)
    if _use_f32_products(qkv, N, d):
        _attention_bwd_f32_products(_c(do), _c(qkv), o, lse, heads, dqkv)
        return dqkv
    if _bf16_attention_needs_products(qkv, N, d, fwd=False):
        d32 = torch.empty(qkv.shape, dtype=torch.float32, device=qkv.device)
        _attention_bwd_f32_products(_c(do).float(), _c(qkv).float(), o, lse, heads, d32)
        return d32.to(qkv.dtype)
    ws = torch.empty(lib.dinox_attention_bwd_ws_bytes(B, N, heads), dtype=torch.uint8, device=qkv.device)
    check(lib.dinox_attention_bwd(_p(do), _p(qkv), _p(o), _p(lse), _p(dqkv), _p(ws), B, N, heads, d, _code(qkv.dtype), _stream()),
          "dinox_attention_bwd")
    return dqkv


# ------------------------------------------------------------------------------------------
# autograd functions
# ------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the fp32 residual stream (reference zoo/arch.py:89,91,187)."""

    @staticmethod
    def forward(ctx, x, w, b, out_dtype, eps):
        y, mean, rstd = layernorm_fwd(x, w, b, out_dtype, eps)
        ctx.save_for_backward(x, w, mean, rstd, b)
        ctx.mode_bf16 = current_dtype() == torch.bfloat16
        if ctx.needs_input_grad[1]:
            grad_sink.use(w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd, b = ctx.saved_tensors
        dx, dw, db, lowp = layernorm_bwd(dy, x, w, mean, rstd, want_lowp=ctx.mode_bf16, b=b if ctx.needs_input_grad[1] else None)
        if lowp is not None:
            lowp_cache.put(dx, lowp)
        return dx, dw, db, None, None


class LayerNormPrecomputedFn(torch.autograd.Function):
    """The model's final LayerNorm when its forward was already produced by the last block's fused fc2 epilogue
    (linear_residual_ln): hands the precomputed output to autograd; backward is the ordinary LayerNorm backward."""

    @staticmethod
    def forward(ctx, x, w, b, y, mean, rstd):
        ctx.save_for_backward(x, w, mean, rstd, b)
        ctx.mode_bf16 = current_dtype() == torch.bfloat16
        if ctx.needs_input_grad[1]:
            grad_sink.use(w, b)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd, b = ctx.saved_tensors
        dx, dw, db, lowp = layernorm_bwd(dy, x.reshape(-1, x.shape[-1]), w, mean, rstd, want_lowp=ctx.mode_bf16,
                                         b=b if ctx.needs_input_grad[1] else None)
        dx = dx.view(x.shape)
        if lowp is not None:
            lowp_cache.put(dx, lowp.view(x.shape))
        return dx, dw, db, None, None, None


class _LowpCache:
    """bf16 copies of residual-stream gradients, handed from the kernel that produced the fp32 gradient
    (LayerNorm backward writes both) to the next backward node, which needs the bf16 form as a GEMM operand."""

    def __init__(self) -> None:
        self.key = None
        self.val = None

    def put(self, t: Tensor, lowp: Tensor) -> None:
        self.key, self.val = (t.data_ptr(), t._version, tuple(t.shape)), lowp

    def take(self, t: Tensor) -> Optional[Tensor]:
        if self.key == (t.data_ptr(), t._version, tuple(t.shape)):
            v, self.key, self.val = self.val, None, None
            return v
        return None


lowp_cache = _LowpCache()


def grad_operand(g: Tensor, dt: torch.dtype) -> Tensor:
    """A residual-stream gradient as a GEMM operand of the current mode (cached bf16 copy if one was emitted)."""
    if dt == torch.float32:
        return _c(g) if g.dtype == torch.float32 else g.float()
    if g.dtype == torch.bfloat16:
        return _c(g)
    hit = lowp_cache.take(g)
    return hit if hit is not None else cast_bf16(g)


class _GradSink:
    """Gradient arena registered by the training engine.  A weight gradient whose parameter is registered is accumulated by the
    dW product itself straight into the parameter's slice of the flat arena (the split-K atomics / ACCUM epilogue land there, the
    bias gradient rides along) and autograd gets ``None``: no zero-fill of a temporary, no temporary, no ``grad += dW`` pass per
    parameter (rocprof: ~150 fills and ~150 adds per ViT-S step).  ``on_ready(i)`` replaces the post-accumulate-grad hook that
    autograd no longer fires for such a parameter (dp.GradBucketer counts gradients with it); it fires when as many products
    have landed as forward passes used the parameter (``use``), so a module applied twice is still exchanged once, complete."""

    def __init__(self) -> None:
        self.slots: dict = {}
        self.uses: dict = {}
        self.on_ready = None
        self.owner = None
        self.recomputing = 0        # > 0 while torch.utils.checkpoint re-runs a forward to rebuild saved tensors

    def register(self, owner, params, on_ready=None) -> None:
        self.slots = {p.data_ptr(): (i, p) for i, p in enumerate(params) if p.requires_grad and p.numel()}
        self.uses, self.on_ready, self.owner = {}, on_ready, owner

    def clear(self) -> None:
        self.slots, self.uses, self.on_ready, self.owner = {}, {}, None, None

    def lookup(self, w: Optional[Tensor]):
        if w is None or not self.slots:
            return None
        slot = self.slots.get(w.data_ptr())
        if slot is None:
            return None
        g = slot[1].grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != w.shape:
            return None
        return slot

    def use(self, *ws) -> None:
        """Forward passes that will back-propagate into these parameters announce themselves (DP only)."""
        if self.on_ready is None or self.recomputing:      # a recomputed forward adds no backward product of its own
            return
        for w in ws:
            slot = self.lookup(w)
            if slot is not None:
                self.uses[slot[0]] = self.uses.get(slot[0], 0) + 1

    def ready(self, slot) -> None:
        if self.on_ready is None:
            return
        left = self.uses.get(slot[0], 1) - 1
        self.uses[slot[0]] = max(left, 0)
        if left <= 0:
            self.on_ready(slot[0])


grad_sink = _GradSink()


@contextlib.contextmanager
def _recompute_scope():
    grad_sink.recomputing += 1
    try:
        yield
    finally:
        grad_sink.recomputing -= 1


def checkpoint_contexts():
    """``context_fn`` for ``torch.utils.checkpoint.checkpoint(use_reentrant=False)``: (original forward, recomputation).
    The recomputation runs every forward of the block a second time only to rebuild its saved tensors; marking it keeps the
    gradient sink's use counts equal to the number of backward products, so data-parallel buckets still fire DURING backward."""
    return contextlib.nullcontext(), _recompute_scope()


def small_grad(param: Optional[Tensor], g: Optional[Tensor]) -> Optional[Tensor]:
    """A small parameter gradient (tokens, position embedding, scale-embedding MLP) produced by a kernel as a tensor of its own:
    with the parameter registered in the gradient sink it is added into the arena by dinox_axpy and autograd gets None (instead of
    one framework add kernel per parameter: nine per ViT-S step)."""
    if g is None or param is None:
        return g
    slot = grad_sink.lookup(param)
    if slot is None or g.dtype != torch.float32 or not g.is_contiguous():
        return g
    axpy_(slot[1].grad.view(-1), g.view(-1), 1.0)
    grad_sink.ready(slot)
    return None


class _DwStream:
    """The weight-gradient products have no consumer inside backward (their results are read by the optimiser, or by a gradient
    bucket's all-reduce): with DINOX_DW_STREAM=1 they are enqueued on a second HIP stream, so that a dW product (matrix pipe) can
    share the chip with the memory-bound kernels of the dX chain (LayerNorm backward, attention backward) instead of standing in
    line between them.  `join()` makes the current stream wait for everything enqueued there (engine: before the optimiser;
    dp.GradBucketer: before a bucket is exchanged)."""

    def __init__(self) -> None:
        self.enabled = bool(os.environ.get("DINOX_DW_STREAM"))
        self.streams: dict = {}
        self.dirty = False

    def side(self, device):
        st = self.streams.get(device)
        if st is None:
            st = self.streams[device] = torch.cuda.Stream(device=device)
        return st

    def run(self, fn, *tensors) -> None:
        main = torch.cuda.current_stream()
        side = self.side(tensors[0].device)
        side.wait_stream(main)                     # operands (and the zeroed arena) are products of the main stream
        with torch.cuda.stream(side):
            fn()
        for t in tensors:                          # their memory must outlive the side stream's use of it
            t.record_stream(side)
        self.dirty = True

    def join(self) -> None:
        if self.dirty:
            main = torch.cuda.current_stream()
            for st in self.streams.values():
                main.wait_stream(st)
            self.dirty = False


dw_stream = _DwStream()


def weight_grad(dy: Tensor, x: Tensor, w: Tensor, bias: Optional[Tensor], want_db: bool):
    """dW = dy^T x  ([N,M].[M,K]) and, if wanted, db = column sums of dy, from one product.
    Returns (dw, db) for autograd -- or (None, None) after accumulating both into the engine's gradient arena."""
    sw = grad_sink.lookup(w)
    sb = grad_sink.lookup(bias) if want_db else None
    if sw is not None and (not want_db or sb is not None):
        def product():
            gemm(dy, x, transA=True, transB=True, out=sw[1].grad.view(w.shape[0], -1), accumulate=True,
                 colsum_out=sb[1].grad if want_db else None)
        if dw_stream.enabled and dy.is_cuda:
            dw_stream.run(product, dy, x)
        else:
            product()
        grad_sink.ready(sw)
        if want_db:
            grad_sink.ready(sb)
        return None, None
    db = torch.empty(w.shape[0], dtype=torch.float32, device=dy.device) if want_db else None
    dw = gemm(dy, x, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
    return dw.reshape(w.shape), db


_BLOCK_NATIVE = os.environ.get("DINOX_BLOCK_NATIVE", "1") != "0"      # "0": the block node issues its launches one by one from Python (A/B, tests)


def _block_native_ok(dt: torch.dtype, x0: Tensor, wqkv: Tensor) -> bool:
    """May this block run as ONE call into the library (dinox_block_forward / _backward, csrc/block.hip)?  bf16 throughput mode only (the
    fp32 parity mode composes full-size attention from several launches on the host), nobody listening to individual launches, and the
    weight-gradient side stream off (its products are enqueued from Python)."""
    return _BLOCK_NATIVE and dt == torch.bfloat16 and x0.is_cuda and TRACE_KERNELS is None and not dw_stream.enabled


def _block_forward_native(ctx, x0, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, pre_ln, next_ln, train):
    """BlockFn.forward through dinox_block_forward: the same kernels in the same order, one foreign call."""
    dt = torch.bfloat16
    V, N, D = x0.shape
    M, H = V * N, w1.shape[0]
    dev = x0.device
    bf = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    if pre_ln is None:
        xn1, mean1, rstd1 = bf(V, N, D), f32(M), f32(M)
    else:
        xn1, mean1, rstd1 = pre_ln
    if not train and _qkv_fused(D) and qkv_attention_ok(V, N, heads, D, D):      # no backward: projection + attention in one launch, no qkv tensor
        qkv, o, lse = None, bf(V, N, D), None
    else:
        qkv, o, lse = bf(M, 3 * D), bf(V, N, D), f32(V, heads, N)
    x1, xn2, mean2, rstd2 = f32(M, D), bf(M, D), f32(M), f32(M)
    act = bf(M, H)
    pre = bf(M, H) if train else None
    x2 = f32(V, N, D)
    nxt = None
    wops = [weight_operand(w_, dt) for w_ in (wqkv, wproj, w1, w2)]          # (held until the call is enqueued: a per-call cast would otherwise be freed)
    a = BlockFwdArgs(V=V, N=N, D=D, H=H, heads=heads, train=int(train), fuse_proj_ln=int(rowln_ok(M, D, D, dt)), fuse_fc2_ln=0, eps=eps,
                     x0=_p(x0), qkv=_p(qkv), o=_p(o), lse=_p(lse), x1=_p(x1), xn2=_p(xn2), mean2=_p(mean2), rstd2=_p(rstd2), act=_p(act), pre=_p(pre),
                     x2=_p(x2), n1w=_p(n1w), n1b=_p(n1b), n2w=_p(n2w), n2b=_p(n2b), wqkv=_p(wops[0]), wproj=_p(wops[1]), w1=_p(wops[2]), w2=_p(wops[3]),
                     bqkv=_p(bqkv), bproj=_p(bproj), b1=_p(b1), b2=_p(b2))
    if pre_ln is None:
        a.xn1, a.mean1, a.rstd1 = _p(xn1), _p(mean1), _p(rstd1)
    else:
        a.xn1_in, a.mean1_in, a.rstd1_in = _p(xn1), _p(mean1), _p(rstd1)
    if next_ln is not None:
        ydt = next_ln[3] or dt
        yn, mn, rn = torch.empty((V, N, D), dtype=ydt, device=dev), f32(M), f32(M)
        a.next_g, a.next_b, a.next_eps, a.next_dtype = _p(next_ln[0]), _p(next_ln[1]), next_ln[2], _code(ydt)
        a.yn, a.meann, a.rstdn = _p(yn), _p(mn), _p(rn)
        a.fuse_fc2_ln = int(rowln_ok(M, D, H, dt, ydt))
        nxt = (yn, mn, rn)
    check(lib.dinox_block_forward(C.byref(a), _stream()), "dinox_block_forward")
    if train:
        ctx.save_for_backward(x0, x1, xn1, xn2, qkv, o, lse, pre, act, mean1, rstd1, mean2, rstd2, n1w, n2w, wqkv, wproj, w1, w2,
                              bqkv, bproj, b1, b2, n1b, n2b)
        ctx.dt, ctx.heads, ctx.shape, ctx.native = dt, heads, (V, N, D), True
        grad_sink.use(wqkv, bqkv, wproj, bproj, w1, b1, w2, b2, n1w, n1b, n2w, n2b)
    if next_ln is None:
        return x2
    ctx.mark_non_differentiable(*nxt)
    return (x2,) + nxt


_BLOCK_TN_WS: dict = {}


def _block_tn_ws_bytes(M: int, D: int, H: int) -> int:
    """Workspace the four dW products of a block need for their deterministic split-K reduction (the largest of them), asked of the
    library once per shape (dinox_gemm_ws_bytes looks at shapes and pointer alignment only)."""
    key = (M, D, H)
    hit = _BLOCK_TN_WS.get(key)
    if hit is None:
        hit = 0
        for n_out, n_in in ((D, H), (H, D), (D, D), (3 * D, D)):
            g = GemmArgs(A=0x1000, B=0x1000, C=0x1000, M=n_out, N=n_in, K=M, lda=n_out, ldb=n_in, ldc=n_in, batch=1, strideA=0, strideB=0, strideC=n_out * n_in,
                         transA=1, transB=1, in_dtype=BF16, out_dtype=F32, epilogue=EPI_ACCUM, alpha=1.0, bias=None, residual=None, ldr=n_in, aux=None,
                         ldaux=n_in, colsum=0x1000, ws=0x1000)
            hit = max(hit, int(lib.dinox_gemm_ws_bytes(C.byref(g))))
        _BLOCK_TN_WS[key] = hit
    return hit


def _block_backward_native(ctx, g: Tensor, saved):
    """BlockFn.backward through dinox_block_backward -- when every parameter gradient of the block goes straight into the gradient arena
    (the training engine's registration); otherwise None and the composed path takes over.  `saved` = ctx.saved_tensors, unpacked by the
    caller (under torch.utils.checkpoint they may be unpacked only once)."""
    (x0, x1, xn1, xn2, qkv, o, lse, pre, act, mean1, rstd1, mean2, rstd2, n1w, n2w, wqkv, wproj, w1, w2, bqkv, bproj, b1, b2,
     n1b, n2b) = saved
    slots = [grad_sink.lookup(t) for t in (wqkv, bqkv, wproj, bproj, w1, b1, w2, b2, n1w, n1b, n2w, n2b)]
    have = [t is not None for t in (wqkv, bqkv, wproj, bproj, w1, b1, w2, b2, n1w, n1b, n2w, n2b)]
    if any(h and s is None for h, s in zip(have, slots)) or not all(have[i] for i in (0, 2, 4, 6, 8, 9, 10, 11)):
        return None
    dt, heads = torch.bfloat16, ctx.heads
    V, N, D = ctx.shape
    M, H = V * N, w1.shape[0]
    dev = g.device
    bf = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)
    g_lp = lowp_cache.take(g)
    g1 = torch.empty((V, N, D), dtype=torch.float32, device=dev)
    g0_lp = bf(V, N, D)
    gp = lambda sl: None if sl is None else sl[1].grad.data_ptr()
    tn_ws = _tn_workspace(max(_block_tn_ws_bytes(M, D, H), 1), dev)
    # scratch of the call: every tensor is HELD until the call has been enqueued (a temporary freed inside the argument list would hand
    # its memory to the next allocation of the same list: dpre, dxn2, ... would alias)
    scratch = dict(dpre=bf(M, H), dxn2=bf(M, D), d_o=bf(M, D), dqkv=bf(M, 3 * D), dxn1=bf(M, D), g1_lowp=bf(M, D),
                   g_lowp_buf=None if g_lp is not None else bf(M, D),
                   attn_ws=torch.empty(lib.dinox_attention_bwd_ws_bytes(V, N, heads), dtype=torch.uint8, device=dev),
                   ln_ws=torch.empty(lib.dinox_layernorm_bwd_ws_bytes(M, D), dtype=torch.uint8, device=dev))
    wts = [weight_operand(w_, dt, transposed=True) for w_ in (wqkv, wproj, w1, w2)]
    fuse_ln = linear_ln_bwd_ok(M, D, H, dt) and linear_ln_bwd_ok(M, D, 3 * D, dt)      # (the same decision as the Python-sequenced backward)
    a = BlockBwdArgs(V=V, N=N, D=D, H=H, heads=heads, reserved=int(fuse_ln), g=_p(g), g_lowp=_p(g_lp), g_lowp_buf=_p(scratch["g_lowp_buf"]),
                     x0=_p(x0), x1=_p(x1), xn1=_p(xn1), xn2=_p(xn2), qkv=_p(qkv), o=_p(o), lse=_p(lse), pre=_p(pre), act=_p(act), mean1=_p(mean1),
                     rstd1=_p(rstd1), mean2=_p(mean2), rstd2=_p(rstd2), n1w=_p(n1w), n2w=_p(n2w),
                     wqkv_t=_p(wts[0]), wproj_t=_p(wts[1]), w1_t=_p(wts[2]), w2_t=_p(wts[3]),
                     dwqkv=gp(slots[0]), dbqkv=gp(slots[1]), dwproj=gp(slots[2]), dbproj=gp(slots[3]), dw1=gp(slots[4]), db1=gp(slots[5]),
                     dw2=gp(slots[6]), db2=gp(slots[7]), dn1w=gp(slots[8]), dn1b=gp(slots[9]), dn2w=gp(slots[10]), dn2b=gp(slots[11]),
                     dpre=_p(scratch["dpre"]), dxn2=_p(scratch["dxn2"]), d_o=_p(scratch["d_o"]), dqkv=_p(scratch["dqkv"]), dxn1=_p(scratch["dxn1"]),
                     g1=_p(g1), g1_lowp=_p(scratch["g1_lowp"]), g0_lowp=_p(g0_lp), attn_ws=_p(scratch["attn_ws"]), ln_ws=_p(scratch["ln_ws"]),
                     tn_ws=_p(tn_ws), tn_ws_bytes=tn_ws.numel())
    check(lib.dinox_block_backward(C.byref(a), _stream()), "dinox_block_backward")
    for sl in slots:
        if sl is not None:
            grad_sink.ready(sl)
    lowp_cache.put(g1, g0_lp)
    return (g1,) + (None,) * 16


_outer_grad_mode = [True]


def block_fn(*args):
    """BlockFn.apply that also tells the node whether the caller runs with gradients enabled (a no-grad pass saves nothing: no GELU' side
    tensor, and with DINOX_QKV_FUSED=1 no packed qkv tensor either)."""
    prev = _outer_grad_mode[0]
    _outer_grad_mode[0] = torch.is_grad_enabled()
    try:
        return BlockFn.apply(*args)
    finally:
        _outer_grad_mode[0] = prev


class BlockFn(torch.autograd.Function):
    """One pre-norm transformer block as a single autograd node (reference zoo/arch.py:94-97 with
    Attention :43-54 and Mlp :75-76 inlined):  x1 = x0 + proj(attn(norm1(x0)));  x2 = x1 + fc2(gelu(fc1(norm2(x1)))).

    13 kernel launches forward, 15 backward, no torch elementwise kernels: residual adds live in GEMM
    epilogues, the skip-connection gradient add and the bf16 cast of the residual gradient live in the
    LayerNorm backward kernel, bias gradients ride along the dW products."""

    @staticmethod
    def forward(ctx, x0, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, pre_ln=None, next_ln=None):
        """``pre_ln`` = (norm1(x0), mean, rstd) when the producer of x0 already normalised it (the previous block's fc2 epilogue);
        ``next_ln`` = (gamma, beta, eps, out_dtype) of the LayerNorm that FOLLOWS this block (the next block's norm1, or the model's
        final norm): then the block also returns (LN(x2), mean, rstd) -- in bf16 mode at width 384 straight from the fc2 product's
        epilogue (linear_residual_ln), as the proj product's epilogue produces norm2(x1); otherwise from LayerNorm launches.
        Returns x2, or (x2, y_next, mean_next, rstd_next) with ``next_ln``."""
        _need_cuda(x0, wqkv)
        ctx.set_materialize_grads(False)      # the non-differentiable outputs (LN of the next block) would otherwise arrive as zero-FILLED tensors
        dt = current_dtype()
        x0 = _c(x0 if x0.dtype == torch.float32 else x0.float())
        V, N, D = x0.shape
        M = V * N
        # needs_input_grad says which inputs COULD take a gradient; under torch.no_grad() (encode(), an evaluation loop over a model whose
        # parameters still require grad) nothing will ever ask for one: block_fn() notes the caller's grad mode (inside forward it is always off)
        train = any(ctx.needs_input_grad) and _outer_grad_mode[0]
        if _block_native_ok(dt, x0, wqkv):
            return _block_forward_native(ctx, x0, n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, w1, b1, w2, b2, heads, eps, pre_ln, next_ln, train)
        ctx.native = False
        if pre_ln is None:
            xn1, mean1, rstd1 = layernorm_fwd(x0, n1w, n1b, dt, eps)
        else:
            xn1, mean1, rstd1 = pre_ln
        if not train and dt == torch.bfloat16 and _qkv_fused(D) and qkv_attention_ok(V, N, heads, D, D):
            o, qkv, lse = qkv_attention(xn1.view(V, N, D), weight_operand(wqkv, dt), bqkv, heads)      # one launch, no qkv tensor
        else:
            qkv = gemm(xn1.view(M, D), weight_operand(wqkv, dt), bias=bqkv, out_dtype=dt)
            o, lse = attention_fwd(qkv.view(V, N, 3 * D), heads)
        if rowln_ok(M, D, D, dt):
            x1, xn2, mean2, rstd2 = linear_residual_ln(o.view(M, D), weight_operand(wproj, dt), bproj, x0.view(M, D), n2w, n2b, eps, dt)
        else:
            x1 = gemm(o.view(M, D), weight_operand(wproj, dt), bias=bproj, residual=x0.view(M, D), out_dtype=torch.float32)
            xn2, mean2, rstd2 = layernorm_fwd(x1, n2w, n2b, dt, eps)
        nxt = None
        # (A fused fc1 -> GELU -> fc2 kernel for passes that save nothing -- hidden activation on chip, -632 MB per block -- was built in
        #  round 1 and removed in round 2: 509 us against 197 + 197 us for the two products at the hot-path shape)
        if True:
            H = w1.shape[0]
            pre = torch.empty((M, H), dtype=dt, device=x0.device) if train else None
            act = gemm(xn2, weight_operand(w1, dt), bias=b1, gelu=True, aux=pre, auxgrad=True, out_dtype=dt)   # pre := gelu'(fc1 out)
            if next_ln is not None and rowln_ok(M, D, H, dt, next_ln[3] or dt):
                x2, yn, mn, rn = linear_residual_ln(act, weight_operand(w2, dt), b2, x1, next_ln[0], next_ln[1], next_ln[2], next_ln[3] or dt)
                nxt = (yn.view(V, N, D), mn, rn)
            else:
                x2 = gemm(act, weight_operand(w2, dt), bias=b2, residual=x1, out_dtype=torch.float32)
            if train:
                ctx.save_for_backward(x0, x1, xn1, xn2, qkv, o, lse, pre, act, mean1, rstd1, mean2, rstd2, n1w, n2w, wqkv, wproj, w1, w2,
                                      bqkv, bproj, b1, b2, n1b, n2b)
                ctx.dt, ctx.heads, ctx.shape = dt, heads, (V, N, D)
                grad_sink.use(wqkv, bqkv, wproj, bproj, w1, b1, w2, b2, n1w, n1b, n2w, n2b)
        x2 = x2.view(V, N, D)
        if next_ln is None:
            return x2
        if nxt is None:
            yn, mn, rn = layernorm_fwd(x2, next_ln[0], next_ln[1], next_ln[3] or dt, next_ln[2])
            nxt = (yn, mn, rn)
        ctx.mark_non_differentiable(*nxt)
        return (x2,) + nxt

    @staticmethod
    def backward(ctx, g, *_unused):
        (x0, x1, xn1, xn2, qkv, o, lse, pre, act, mean1, rstd1, mean2, rstd2, n1w, n2w, wqkv, wproj, w1, w2, bqkv, bproj, b1, b2,
         n1b, n2b) = ctx.saved_tensors
        dt, heads = ctx.dt, ctx.heads
        V, N, D = ctx.shape
        M = V * N
        dev = g.device
        g = _c(g if g.dtype == torch.float32 else g.float())
        bf = dt == torch.bfloat16
        if ctx.native:
            native = _block_backward_native(ctx, g, (x0, x1, xn1, xn2, qkv, o, lse, pre, act, mean1, rstd1, mean2, rstd2, n1w, n2w, wqkv, wproj, w1, w2,
                                                     bqkv, bproj, b1, b2, n1b, n2b))
            if native is not None:
                return native

        def wt(w):      # W^T operand for dX = dY . W
            return (w.detach(), dict(transB=True)) if not bf else (weight_operand(w, dt, transposed=True), {})

        g_op = grad_operand(g, dt).view(M, D)
        # ---- MLP: x2 = x1 + fc2(gelu(fc1(xn2)))
        b, kw = wt(w2)
        dpre = gemm(g_op, b, dgelu=True, aux=pre, auxgrad=True, out_dtype=dt, **kw)
        dw2, db2 = weight_grad(g_op, act, w2, b2, b2 is not None)
        b, kw = wt(w1)
        fuse_ln = bf and linear_ln_bwd_ok(M, D, dpre.shape[1], dt) and linear_ln_bwd_ok(M, D, 3 * D, dt)
        if not fuse_ln:
            dxn2 = gemm(dpre, b, out_dtype=dt, **kw)
        dw1, db1 = weight_grad(dpre, xn2.view(M, D), w1, b1, b1 is not None)
        if fuse_ln:     # g1 = g + LN2'(dpre W1): product and LayerNorm backward in one launch
            g1, dn2w, dn2b, g1_lp = linear_ln_bwd(dpre, b, x1, n2w, mean2, rstd2, dx_add=g.view(M, D), want_lowp=True, b=n2b)
        else:
            g1, dn2w, dn2b, g1_lp = layernorm_bwd(dxn2, x1, n2w, mean2, rstd2, dx_add=g.view(M, D), want_lowp=bf, b=n2b)   # g1 = g + LN2'(.)
        del dpre
        g1_op = g1_lp if bf else g1
        # ---- attention: x1 = x0 + proj(attn(qkv(xn1)))
        b, kw = wt(wproj)
        do = gemm(g1_op, b, out_dtype=dt, **kw)
        dwp, dbp = weight_grad(g1_op, o.view(M, D), wproj, bproj, bproj is not None)
        dqkv = attention_bwd(do.view(V, N, D), qkv.view(V, N, 3 * D), o, lse, heads).view(M, 3 * D)
        b, kw = wt(wqkv)
        if not fuse_ln:
            dxn1 = gemm(dqkv, b, out_dtype=dt, **kw)
        dwq, dbq = weight_grad(dqkv, xn1.view(M, D), wqkv, bqkv, bqkv is not None)
        if not bf:
            dw_stream.join()        # fp32 mode: the proj dW product reads g1 itself (bf16 mode: its own low-precision copy)
        if fuse_ln:
            g0, dn1w, dn1b, g0_lp = linear_ln_bwd(dqkv, b, x0, n1w, mean1, rstd1, dx=g1, dx_add=g1, want_lowp=True, b=n1b)   # in place on our own g1
        else:
            g0, dn1w, dn1b, g0_lp = layernorm_bwd(dxn1, x0, n1w, mean1, rstd1, dx=g1, dx_add=g1, want_lowp=bf, b=n1b)      # in place on our own g1
        g0 = g0.view(V, N, D)
        if g0_lp is not None:
            lowp_cache.put(g0, g0_lp.view(V, N, D))
        return g0, dn1w, dn1b, dwq, dbq, dwp, dbp, dn2w, dn2b, dw1, db1, dw2, db2, None, None, None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b (+ residual), nn.Linear semantics (reference zoo/arch.py:40-41,46,53)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, out_dtype):
        _need_cuda(x, w)
        dt = current_dtype()
        xm = to_mode(x, dt)
        K = xm.shape[-1]
        x2 = xm.reshape(-1, K)
        odt = torch.float32 if residual is not None else (out_dtype or dt)
        y = gemm(x2, weight_operand(w, dt), bias=b, residual=None if residual is None else _c(residual).reshape(-1, w.shape[0]),
                 out_dtype=odt)
        ctx.save_for_backward(x2, w, b)
        if ctx.needs_input_grad[1]:
            grad_sink.use(w, b if ctx.needs_input_grad[2] else None)
        ctx.dt, ctx.has_bias, ctx.has_res, ctx.xshape, ctx.xdtype = dt, b is not None, residual is not None, x.shape, x.dtype
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, b = ctx.saved_tensors
        dt = ctx.dt
        dy2 = to_mode(dy.reshape(-1, w.shape[0]), dt)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if dt == torch.float32:
                dx = gemm(dy2, w.detach(), transB=True, out_dtype=ctx.xdtype)            # dy [M,N] . W [N,K]
            else:
                dx = gemm(dy2, weight_operand(w, dt, transposed=True), out_dtype=ctx.xdtype if ctx.xdtype == dt else dt)
                if dx.dtype != ctx.xdtype:
                    dx = dx.to(ctx.xdtype)
            dx = dx.reshape(ctx.xshape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = weight_grad(dy2, x2, w, b, want_db)                                            # dy^T [N,M] . x [M,K] (+ db)
        elif want_db:
            db = colsum(dy2)
        dres = dy if ctx.has_res else None
        return dx, dw, db, dres, None


class MlpFn(torch.autograd.Function):
    """fc2(GELU_erf(fc1(x))) (+ residual) with GELU / GELU' fused into the GEMM epilogues
    (reference zoo/arch.py:71-76; also the DINO head, zoo/arch.py:252-256)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, out_dtype):
        _need_cuda(x, w1, w2)
        dt = current_dtype()
        xm = to_mode(x, dt)
        x2 = xm.reshape(-1, xm.shape[-1])
        M, H = x2.shape[0], w1.shape[0]
        pre = torch.empty((M, H), dtype=dt, device=x.device) if any(ctx.needs_input_grad[:5]) else None
        act = gemm(x2, weight_operand(w1, dt), bias=b1, gelu=True, aux=pre, auxgrad=True, out_dtype=dt)
        odt = torch.float32 if residual is not None else (out_dtype or dt)
        y = gemm(act, weight_operand(w2, dt), bias=b2, residual=None if residual is None else _c(residual).reshape(M, w2.shape[0]),
                 out_dtype=odt)
        ctx.save_for_backward(x2, w1, w2, pre, act, b1, b2)
        if pre is not None:
            grad_sink.use(w1, b1, w2, b2)
        ctx.dt, ctx.has_res, ctx.xshape, ctx.xdtype = dt, residual is not None, x.shape, x.dtype
        return y.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, pre, act, b1, b2 = ctx.saved_tensors
        dt = ctx.dt
        dy2 = to_mode(dy.reshape(-1, w2.shape[0]), dt)
        if dt == torch.float32:
            dpre = gemm(dy2, w2.detach(), transB=True, dgelu=True, aux=pre, auxgrad=True, out_dtype=dt)
        else:
            dpre = gemm(dy2, weight_operand(w2, dt, transposed=True), dgelu=True, aux=pre, auxgrad=True, out_dtype=dt)
        dw2, db2 = weight_grad(dy2, act, w2, b2, b2 is not None)
        dx = None
        if ctx.needs_input_grad[0]:
            if dt == torch.float32:
                dx = gemm(dpre, w1.detach(), transB=True, out_dtype=dt)
            else:
                dx = gemm(dpre, weight_operand(w1, dt, transposed=True), out_dtype=dt)
            if dx.dtype != ctx.xdtype:
                dx = dx.to(ctx.xdtype)
            dx = dx.reshape(ctx.xshape)
        dw1, db1 = weight_grad(dpre, x2, w1, b1, b1 is not None)
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None), None


class AttentionCoreFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(d)) V on the packed qkv tensor (reference zoo/arch.py:45-52)."""

    @staticmethod
    def forward(ctx, qkv, heads):
        o, lse = attention_fwd(qkv, heads)
        ctx.save_for_backward(qkv, o, lse)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        return attention_bwd(do, qkv, o, lse, ctx.heads), None


class _UnfoldShare:
    """The student and the teacher of one training step see the same batch (scripts/phase5_big_run.py:1741-1743) and can share
    its unfolded form.  Sharing is OPT-IN and scoped: only inside ``with unfold_share():`` (TrainEngine.step) is an unfolded
    batch kept, the entry holds the input tensor itself (so its memory cannot be recycled under the entry) and is matched by
    object identity + version; leaving the scope drops it.  A plain ``PatchViT.forward`` (encode(), eval loops) never looks here."""

    def __init__(self) -> None:
        self.depth = 0
        self.entries: list = []

    def find(self, x: Tensor, patch: int, dt: torch.dtype) -> Optional[Tensor]:
        for (t, ver, p, d, u) in self.entries:
            if t is x and ver == x._version and p == patch and d == dt:
                return u
        return None

    def put(self, x: Tensor, patch: int, dt: torch.dtype, u: Tensor) -> None:
        self.entries = [e for e in self.entries if e[0] is not x][-3:] + [(x, x._version, patch, dt, u)]


_unfold_share = _UnfoldShare()


@contextlib.contextmanager
def unfold_share():
    _unfold_share.depth += 1
    try:
        yield
    finally:
        _unfold_share.depth -= 1
        if _unfold_share.depth == 0:
            _unfold_share.entries = []


def patch_cols(patch: int, dt: torch.dtype) -> int:
    """Columns of the unfolded patch matrix: 3 p^2, rounded up to a multiple of 64 in the bf16 mode when 3 p^2 is no multiple of 8 (patch
    14: 588 -> 640) -- the MFMA bf16 products take no other shape; the tail columns are zeros on both operands."""
    k = 3 * patch * patch
    return k if (dt != torch.bfloat16 or k % 8 == 0) else (k + 63) // 64 * 64


def padded_patch_weight(pw: Tensor, dt: torch.dtype, cols: int) -> Tensor:
    """The patch-embedding weight [D, 3 p^2] as a [D, cols] operand with zero tail columns (built once per optimiser step)."""
    key = (id(pw), "padK", cols)
    hit = weight_cache.extra.get(key)
    if hit is not None and hit[0]() is pw and hit[2] == (pw.data_ptr(), pw._version):
        return hit[1]
    w2 = pw.detach().reshape(pw.shape[0], -1)
    out = torch.zeros((w2.shape[0], cols), dtype=dt, device=pw.device)
    out[:, :w2.shape[1]].copy_(w2)
    weight_cache.extra[key] = (weakref.ref(pw), out, (pw.data_ptr(), pw._version))
    return out


class PatchOperand:
    """A batch of views that is ALREADY the patch-embed operand: ``u`` = [V*P, patch_cols(patch, dtype)], made by
    ``views.make_views(..., patch=p, operand_dtype=dt)`` (dinox_slice_views_patches writes it straight from the u16 stacks).  Stands in
    for the (V,3,S,S) fp32 image batch wherever the training path takes one -- PatchViT.forward, TrainEngine.step -- and answers
    the few questions they ask of it (shape, device); ``patch_unfold`` hands out ``u`` itself."""

    def __init__(self, u: Tensor, V: int, size: int, patch: int) -> None:
        g = size // patch
        assert size % patch == 0 and u.dim() == 2 and u.shape[0] == V * g * g and u.shape[1] >= 3 * patch * patch, (tuple(u.shape), V, size, patch)
        self.u, self.patch, self.size = u, patch, size
        self.shape = torch.Size((V, 3, size, size))

    device = property(lambda self: self.u.device)
    dtype = property(lambda self: self.u.dtype)
    is_cuda = property(lambda self: self.u.is_cuda)

    def dim(self) -> int:
        return 4

    def clone(self) -> "PatchOperand":
        return PatchOperand(self.u.clone(), self.shape[0], self.size, self.patch)

    def copy_(self, other: "PatchOperand", non_blocking: bool = False) -> "PatchOperand":
        if not isinstance(other, PatchOperand) or other.patch != self.patch:
            raise TypeError("a patch-operand batch can only be overwritten by another one of the same patch size")
        self.u.copy_(other.u, non_blocking=non_blocking)
        return self

    def record_stream(self, stream) -> None:
        self.u.record_stream(stream)


def patch_unfold(x, patch: int, dt: torch.dtype) -> Tensor:
    """[V,3,H,W] fp32 -> [V*P, patch_cols(patch, dt)] in dt (shared between student and teacher inside ``unfold_share()`` only).
    A ``PatchOperand`` is that matrix already."""
    if isinstance(x, PatchOperand):
        if x.patch != patch or x.u.dtype != dt or x.u.shape[1] != patch_cols(patch, dt):
            raise ValueError(f"the batch was unfolded for patch {x.patch} in {x.u.dtype} ({x.u.shape[1]} columns); this model takes "
                             f"patch {patch} in {dt} ({patch_cols(patch, dt)} columns)")
        _need_cuda(x.u)
        return x.u
    _need_cuda(x)
    sharing = _unfold_share.depth > 0
    if sharing:
        hit = _unfold_share.find(x, patch, dt)
        if hit is not None:
            return hit
    x0 = x
    x = _c(x)
    if x.dtype != torch.float32:
        x = x.float()
    V, Cn, H, W = x.shape
    assert Cn == 3, "2.5D slice stacks have 3 channels"
    cols = patch_cols(patch, dt)
    u = torch.empty((V * (H // patch) * (W // patch), cols), dtype=dt, device=x.device)
    if cols == 3 * patch * patch:
        check(lib.dinox_patch_unfold(_p(x), _p(u), V, H, W, patch, _code(dt), _stream()), "dinox_patch_unfold")
    else:
        check(lib.dinox_patch_unfold_ld(_p(x), _p(u), V, H, W, patch, cols, _code(dt), _stream()), "dinox_patch_unfold_ld")
    if sharing:
        _unfold_share.put(x0, patch, dt, u)
    return u


class TokensFn(torch.autograd.Function):
    """patch-embed conv (as GEMM) + [CLS|patches] + pos (+scale) + registers (reference zoo/arch.py:216-229)."""

    @staticmethod
    def forward(ctx, x, pw, pb, cls, pos, regs, scale, patch):
        dt = current_dtype()
        u = patch_unfold(x, patch, dt)
        V = x.shape[0]
        D = pw.shape[0]
        P = u.shape[0] // V
        R = 0 if regs is None else regs.shape[1]
        K0 = 3 * patch * patch
        wop = weight_operand(pw, dt) if u.shape[1] == K0 else padded_patch_weight(pw, dt, u.shape[1])
        patches = gemm(u, wop, bias=pb, out_dtype=dt)
        tokens = torch.empty((V, 1 + P + R, D), dtype=torch.float32, device=x.device)
        sc = None if scale is None else _c(scale).reshape(V, D)
        check(lib.dinox_tokens_fwd(_p(patches), _p(_c(cls)), _p(_c(pos)), _p(None if regs is None else _c(regs)), _p(sc), _p(tokens),
                                   V, P, R, D, _code(dt), _stream()), "dinox_tokens_fwd")
        ctx.save_for_backward(u, pw, pb)
        ctx.small = (cls, pos if isinstance(pos, torch.nn.Parameter) else None, regs)       # (an interpolated pos is not a leaf)
        grad_sink.use(pw, pb, cls, ctx.small[1], regs)
        ctx.dims, ctx.dt, ctx.has_scale = (V, P, R, D), dt, scale is not None
        return tokens

    @staticmethod
    def backward(ctx, dtok):
        u, pw, pb = ctx.saved_tensors
        V, P, R, D = ctx.dims
        dt = ctx.dt
        dtok = _c(dtok)
        dev = dtok.device
        dpatches = torch.empty((V * P, D), dtype=dt, device=dev)
        dcls = torch.empty((1, 1, D), dtype=torch.float32, device=dev)
        dpos = torch.empty((1, 1 + P, D), dtype=torch.float32, device=dev)
        dregs = torch.empty((1, R, D), dtype=torch.float32, device=dev) if R else None
        dscale = torch.empty((V, 1, D), dtype=torch.float32, device=dev) if ctx.has_scale else None
        check(lib.dinox_tokens_bwd(_p(dtok), _p(dpatches), _p(dcls), _p(dpos), _p(dregs), _p(dscale), V, P, R, D, _code(dt), _stream()),
              "dinox_tokens_bwd")
        K0 = pw[0].numel()
        if u.shape[1] == K0:
            dw, db = weight_grad(dpatches, u, pw, pb, pb is not None)
        else:
            # padded operand (patch 14): the product gives [D, cols]; its first 3 p^2 columns are the gradient.  One strided add per step
            # into the arena (the only place of a step where a framework kernel does arithmetic, and only at such patch sizes).
            dbt = torch.empty(D, dtype=torch.float32, device=dev) if pb is not None else None
            dwp = gemm(dpatches, u, transA=True, transB=True, out_dtype=torch.float32, colsum_out=dbt)
            sw, sb = grad_sink.lookup(pw), (grad_sink.lookup(pb) if pb is not None else None)
            if sw is not None and (pb is None or sb is not None):
                sw[1].grad.view(D, K0).add_(dwp[:, :K0])
                grad_sink.ready(sw)
                if pb is not None:
                    axpy_(sb[1].grad.view(-1), dbt, 1.0)
                    grad_sink.ready(sb)
                dw = db = None
            else:
                dw, db = dwp[:, :K0].reshape(pw.shape).contiguous(), dbt
        cls, pos, regs = ctx.small
        return None, dw, db, small_grad(cls, dcls), small_grad(pos, dpos), small_grad(regs, dregs), dscale, None


class ScaleEmbedFn(torch.autograd.Function):
    """ScaleEmbedding MLP (reference zoo/arch.py:119-140), fp32, fused per row."""

    @staticmethod
    def forward(ctx, spacing, w0, b0, w2, b2, lnw, lnb, eps):
        _need_cuda(spacing, w0)
        sp = _c(spacing.float())
        V, h, D = sp.shape[0], w0.shape[0], w2.shape[0]
        dev = sp.device
        out = torch.empty((V, 1, D), dtype=torch.float32, device=dev)
        hpre = torch.empty((V, h), dtype=torch.float32, device=dev)
        e = torch.empty((V, D), dtype=torch.float32, device=dev)
        mean = torch.empty(V, dtype=torch.float32, device=dev)
        rstd = torch.empty(V, dtype=torch.float32, device=dev)
        check(lib.dinox_scale_embed_fwd(_p(sp), _p(_c(w0)), _p(b0), _p(_c(w2)), _p(b2), _p(lnw), _p(lnb), _p(out), _p(hpre), _p(e),
                                        _p(mean), _p(rstd), V, h, D, eps, _stream()), "dinox_scale_embed_fwd")
        ctx.save_for_backward(sp, w0, w2, lnw, hpre, e, mean, rstd)
        ctx.small = (w0, b0, w2, b2, lnw, lnb)
        grad_sink.use(*ctx.small)
        return out

    @staticmethod
    def backward(ctx, dout):
        sp, w0, w2, lnw, hpre, e, mean, rstd = ctx.saved_tensors
        V, h, D = sp.shape[0], w0.shape[0], w2.shape[0]
        dev = sp.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        dw0, db0, dw2, db2, dlnw, dlnb = f(h, 3), f(h), f(D, h), f(D), f(D), f(D)
        dsp = f(V, 3) if ctx.needs_input_grad[0] else None
        ws = torch.empty(lib.dinox_scale_embed_bwd_ws_bytes(V, h, D), dtype=torch.uint8, device=dev)
        check(lib.dinox_scale_embed_bwd(_p(_c(dout)), _p(sp), _p(_c(w0)), _p(_c(w2)), _p(lnw), _p(hpre), _p(e), _p(mean), _p(rstd),
                                        _p(dw0), _p(db0), _p(dw2), _p(db2), _p(dlnw), _p(dlnb), _p(dsp), _p(ws), V, h, D, _stream()),
              "dinox_scale_embed_bwd")
        pw0, pb0, pw2, pb2, plnw, plnb = ctx.small
        return (dsp, small_grad(pw0, dw0), small_grad(pb0, db0), small_grad(pw2, dw2), small_grad(pb2, db2), small_grad(plnw, dlnw),
                small_grad(plnb, dlnb), None)


def dino_ce(s: Tensor, t: Tensor, center: Tensor, student_temp: float, teacher_temp: float, want_grad: bool, grad_scale: float = 1.0):
    """Returns (loss[1], ds or None).  s, t: [2B, K]."""
    _need_cuda(s, t, center)
    s, t = _c(s.float()), _c(t.float())
    rows, K = s.shape
    loss = torch.empty(1, dtype=torch.float32, device=s.device)
    row_loss = torch.empty(rows, dtype=torch.float32, device=s.device)
    ds = torch.empty_like(s) if want_grad else None
    check(lib.dinox_dino_ce(_p(s), _p(t), _p(_c(center).reshape(-1)), student_temp, teacher_temp, grad_scale, _p(loss), _p(ds), _p(row_loss),
                            rows, K, _stream()), "dinox_dino_ce")
    return loss, ds


def dino_ce_multi(s: Tensor, t: Tensor, center: Tensor, student_temp: float, teacher_temp: float, n_global: int, grad_scale: float = 1.0):
    """Multi-crop DINO CE (see DinoCEMultiFn): returns (loss[1], ds) with ds already multiplied by grad_scale."""
    _need_cuda(s, t, center)
    sf, tf = _c(s.float()), _c(t.float())
    K = sf.shape[1]
    B = tf.shape[0] // n_global
    n_views = sf.shape[0] // B
    assert tf.shape[0] == n_global * B and sf.shape[0] == n_views * B and tf.shape[1] == K, (tuple(s.shape), tuple(t.shape), n_global)
    loss = torch.empty(1, dtype=torch.float32, device=sf.device)
    ds = torch.empty_like(sf)
    ws = torch.empty((n_views + 2 * n_global) * B, dtype=torch.float32, device=sf.device)
    check(lib.dinox_dino_ce_multi(_p(sf), _p(tf), _p(_c(center).reshape(-1)), student_temp, teacher_temp, grad_scale, _p(loss), _p(ds), _p(ws),
                                  B, n_global, n_views, K, _stream()), "dinox_dino_ce_multi")
    return loss, ds


class DinoCEFn(torch.autograd.Function):
    """DINO centring/sharpening cross-entropy (reference scripts/phase5_big_run.py:692-717)."""

    @staticmethod
    def forward(ctx, s, t, center, student_temp, teacher_temp):
        loss, ds = dino_ce(s, t, center, student_temp, teacher_temp, ctx.needs_input_grad[0])
        ctx.save_for_backward(ds)
        ctx.sdtype = s.dtype
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return (ds * g).to(ctx.sdtype), None, None, None, None


class DinoCEMultiFn(torch.autograd.Function):
    """Multi-crop DINO cross-entropy (extension; the reference has 2 global views only): student rows view-major
    [(G+L) views][B], globals first; teacher rows [G][B]; mean over every (teacher view, other student view) pair.
    G = 2, L = 0 equals DinoCEFn."""

    @staticmethod
    def forward(ctx, s, t, center, student_temp, teacher_temp, n_global):
        _need_cuda(s, t, center)
        sf, tf = _c(s.float()), _c(t.float())
        K = sf.shape[1]
        B = tf.shape[0] // n_global
        n_views = sf.shape[0] // B
        assert tf.shape[0] == n_global * B and sf.shape[0] == n_views * B and tf.shape[1] == K, (tuple(s.shape), tuple(t.shape), n_global)
        loss = torch.empty(1, dtype=torch.float32, device=sf.device)
        ds = torch.empty_like(sf) if ctx.needs_input_grad[0] else None
        ws = torch.empty((n_views + 2 * n_global) * B, dtype=torch.float32, device=sf.device)
        check(lib.dinox_dino_ce_multi(_p(sf), _p(tf), _p(_c(center).reshape(-1)), student_temp, teacher_temp, 1.0, _p(loss), _p(ds), _p(ws),
                                      B, n_global, n_views, K, _stream()), "dinox_dino_ce_multi")
        ctx.save_for_backward(ds)
        ctx.sdtype = s.dtype
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return (ds * g).to(ctx.sdtype), None, None, None, None, None


_POS_W: dict = {}


def _bicubic_matrix_1d(n_in: int, n_out: int):
    """[n_out][n_in] weights of torch's non-antialiased bicubic resize (align_corners=False, Keys cubic A = -0.75, border
    replication): what F.interpolate(mode="bicubic") applies along one axis."""
    A = -0.75
    rows = [[0.0] * n_in for _ in range(n_out)]
    scale = n_in / n_out
    for i in range(n_out):
        x = (i + 0.5) * scale - 0.5
        ix = math.floor(x)
        t = x - ix
        w = (((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
             ((A + 2) * t - (A + 3)) * t * t + 1,
             ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
             ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A)
        for k in range(4):
            rows[i][min(max(ix - 1 + k, 0), n_in - 1)] += w[k]
    return rows


class PosInterpFn(torch.autograd.Function):
    """Position embedding of a crop whose patch grid differs from the model's (multi-crop extension; the reference adds
    ``pos_embed`` as is and cannot take another input size): the CLS entry is kept, the g x g patch grid is resized to
    g_out x g_out by bicubic interpolation, written as ONE fp32 product with the constant [g_out^2, g^2] weight matrix
    (kron of the two 1-D bicubic matrices); backward is the transposed product."""

    @staticmethod
    def forward(ctx, pos, g_out):
        _need_cuda(pos)
        P_in, D = pos.shape[1] - 1, pos.shape[2]
        g_in = int(round(math.sqrt(P_in)))
        assert g_in * g_in == P_in, "position grid must be square"
        key = (g_in, g_out, pos.device)
        W = _POS_W.get(key)
        if W is None:
            m = torch.tensor(_bicubic_matrix_1d(g_in, g_out), dtype=torch.float64)
            W = _POS_W[key] = torch.kron(m, m).float().to(pos.device).contiguous()
        out = torch.empty((1, 1 + g_out * g_out, D), dtype=torch.float32, device=pos.device)
        pf = _c(pos.detach().float())
        out[0, :1].copy_(pf[0, :1])
        gemm(W, pf[0, 1:], transB=True, out=out[0, 1:])
        ctx.W = W
        # A second path into pos_embed: it is counted like the direct one (TokensFn) and its gradient goes the same way -- into the
        # gradient arena, announced to the bucketer once per use.  (Round 2 returned dpos to autograd here: under data parallelism with
        # local crops the arena slice was then all-reduced when the GLOBAL pass had landed, before AccumulateGrad added this part.)
        ctx.pos = pos if isinstance(pos, torch.nn.Parameter) else None
        grad_sink.use(ctx.pos)
        return out

    @staticmethod
    def backward(ctx, g):
        W = ctx.W
        g = _c(g.float())
        dpos = torch.empty((1, 1 + W.shape[1], g.shape[2]), dtype=torch.float32, device=g.device)
        dpos[0, :1].copy_(g[0, :1])
        gemm(W, g[0, 1:], transA=True, transB=True, out=dpos[0, 1:])
        return small_grad(ctx.pos, dpos), None


def interp_pos(pos: Tensor, g_out: int) -> Tensor:
    return PosInterpFn.apply(pos, g_out)


def take_rows(src: Tensor, row: int, dt: torch.dtype, out: Optional[Tensor] = None, out_row0: int = 0, out_rows: int = 0) -> Tensor:
    """src [V,N,D] fp32 -> out[out_row0 + v] = src[v, row] in dt (``feats[:, 0]`` without a strided copy + cast)."""
    _need_cuda(src)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 3
    V, N, D = src.shape
    if out is None:
        out = torch.empty((out_rows or V, D), dtype=dt, device=src.device)
    check(lib.dinox_take_rows(src.data_ptr() + 4 * row * D, _p(out), V, N * D, D, out_row0, _code(dt), _stream()), "dinox_take_rows")
    return out


def put_rows_(dst: Tensor, row: int, src: Tensor, src_row0: int = 0, accumulate: bool = False) -> None:
    """dst[v, row] (+)= src[src_row0 + v]  (dst [V,N,D] fp32; src [*,D] fp32 or bf16)."""
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.dim() == 3 and src.is_contiguous()
    V, N, D = dst.shape
    check(lib.dinox_put_rows(_p(src), dst.data_ptr() + 4 * row * D, V, N * D, D, src_row0, _code(src.dtype), int(accumulate), _stream()),
          "dinox_put_rows")


def axpy_(y: Tensor, x: Tensor, alpha: float) -> None:
    assert y.dtype == torch.float32 and x.dtype == torch.float32 and y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel()
    check(lib.dinox_axpy(_p(y), _p(x), alpha, y.numel(), _stream()), "dinox_axpy")


def lincomb3(a: Tensor, b: Optional[Tensor], c: Optional[Tensor], wb: float, wc: float) -> Tensor:
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(lib.dinox_lincomb3(_p(a), _p(b), _p(c), wb, wc, _p(out), _stream()), "dinox_lincomb3")
    return out


def zero_(t: Tensor) -> None:
    assert t.is_contiguous()
    check(lib.dinox_zero(_p(t), t.numel() * t.element_size(), _stream()), "dinox_zero")


def colmean(t: Tensor) -> Tensor:
    t = _c(t.float())
    out = torch.empty(t.shape[1], dtype=torch.float32, device=t.device)
    check(lib.dinox_colmean(_p(t), _p(out), t.shape[0], t.shape[1], _stream()), "dinox_colmean")
    return out


def center_ema_(center: Tensor, batch_mean: Tensor, momentum: float) -> None:
    assert center.is_contiguous() and center.dtype == torch.float32
    check(lib.dinox_center_ema(_p(center), _p(batch_mean), momentum, center.numel(), _stream()), "dinox_center_ema")


def gram_loss_fwd(sf: Tensor, tf: Tensor, dt: torch.dtype):
    """Returns (loss[1], saved) with saved = (diff [V,T,T] fp32, shat, snorm) for backward."""
    _need_cuda(sf, tf)
    sf, tf = _c(sf.float()), _c(tf.float())
    V, N, D = sf.shape
    T = N - 1
    dev = sf.device
    cat = torch.empty((V, T, 2 * D), dtype=dt, device=dev)
    catneg = torch.empty((V, T, 2 * D), dtype=dt, device=dev)
    shat = torch.empty((V, T, D), dtype=dt, device=dev)
    snorm = torch.empty((V, T), dtype=torch.float32, device=dev)
    check(lib.dinox_gram_normalize(_p(sf), _p(tf), _p(cat), _p(catneg), _p(shat), _p(snorm), V, N, D, _code(dt), _stream()), "dinox_gram_normalize")
    diff = gemm(cat, catneg, out_dtype=torch.float32)          # [V,T,T] = Gs - Gt in one batched NT GEMM (K = 2D)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(1024, dtype=torch.float32, device=dev)
    check(lib.dinox_sqsum(_p(diff), diff.numel(), 1.0 / float(V * T * T), _p(loss), _p(ws), _stream()), "dinox_sqsum")
    return loss, (diff, shat, snorm)


def gram_loss_bwd(saved, sf_shape, scale: float, dfeats: Optional[Tensor] = None, accumulate=False) -> Tensor:
    diff, shat, snorm = saved
    V, N, D = sf_shape
    T = N - 1
    dt = shat.dtype
    d_in = diff if dt == torch.float32 else cast_bf16(diff)
    # dXh = (4*scale/(V T^2)) * diff . Xh ; diff is symmetric, so it is also the [K=u][M=t] operand of a TN product
    dxh = gemm(d_in, shat, transA=True, transB=True, out_dtype=torch.float32, alpha=4.0 * scale / float(V * T * T))
    if dfeats is None:
        dfeats = torch.zeros((V, N, D), dtype=torch.float32, device=diff.device)
    check(lib.dinox_gram_normalize_bwd(_p(dxh), _p(shat), _p(snorm), None, _p(dfeats), V, N, D, _code(dt), int(accumulate), _stream()),
          "dinox_gram_normalize_bwd")
    return dfeats


class GramLossFn(torch.autograd.Function):
    """Gram-anchoring loss (reference scripts/phase5_big_run.py:723-739)."""

    @staticmethod
    def forward(ctx, sf, tf):
        dt = current_dtype()
        loss, saved = gram_loss_fwd(sf, tf, dt)
        ctx.save_for_backward(*saved)
        ctx.shape = tuple(sf.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d = gram_loss_bwd(ctx.saved_tensors, ctx.shape, 1.0)
        return d * g, None


def koleo_begin(x: Tensor, group=None):
    """First half of koleo_fwd: normalise the rows and -- under data parallelism -- START the all-gather of the unit rows (V x out_dim fp32 per
    rank, 134 MB at 8 x 512 x 8192) without waiting for it, so that the caller can run independent work (the Gram loss) under the transfer."""
    import torch.distributed as dist
    _need_cuda(x)
    x = _c(x.float())
    V, D = x.shape
    dev = x.device
    from .dp import exchanging
    gather = exchanging(group)
    world = dist.get_world_size(group) if gather else 1
    rank = dist.get_rank(group) if gather else 0
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    xh, norm, sq = f(V, D), f(V), f(V)
    check(lib.dinox_koleo_normalize(_p(x), _p(xh), _p(norm), _p(sq), V, D, 1e-12, _stream()), "dinox_koleo_normalize")
    works = []
    if gather:
        xh_all, sq_all = f(world * V, D), f(world * V)
        works.append(dist.all_gather_into_tensor(xh_all, xh, group=group, async_op=True))
        works.append(dist.all_gather_into_tensor(sq_all, sq, group=group, async_op=True))
    else:
        xh_all, sq_all = xh, sq
    return (xh, xh_all, sq_all, norm, works, group, gather, world, rank, V, D)


def koleo_end(state, eps: float = 1e-8):
    """Second half: wait for the gathered rows, nearest neighbours over the global batch, the loss.  Returns (loss[1], saved)."""
    import torch.distributed as dist
    xh, xh_all, sq_all, norm, works, group, gather, world, rank, V, D = state
    for w in works:
        w.wait()
    dev = xh.device
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    Vg, row0 = world * V, rank * V
    G = gemm_nt_f32_splitk(xh, xh_all)                                  # [V, Vg] inner products, exact-fp32 MFMA, reduction split over the chip
    idx = torch.empty(V, dtype=torch.int32, device=dev)
    dmin = f(V)
    check(lib.dinox_koleo_nn(_p(G), Vg, _p(sq_all), _p(xh_all), row0, V, Vg, D, _p(idx), _p(dmin), _stream()), "dinox_koleo_nn")
    if gather:
        idx_all = torch.empty(Vg, dtype=torch.int32, device=dev)
        d_all = f(Vg)
        dist.all_gather_into_tensor(idx_all, idx, group=group)
        dist.all_gather_into_tensor(d_all, dmin, group=group)
    else:
        idx_all, d_all = idx, dmin
    loss = f(1)
    check(lib.dinox_koleo_loss(_p(dmin), V, eps, _p(loss), _stream()), "dinox_koleo_loss")
    return loss, (xh_all, idx_all, d_all, norm, (row0, V, Vg, D), eps)


def koleo_fwd(x: Tensor, eps: float = 1e-8, group=None):
    """KoLeo regulariser on the student head output (reference scripts/phase5_big_run.py:742-773, applied at :1764-1766):
    -mean_i log(min_{j != i} ||x^_i - x^_j|| + eps) with x^ = F.normalize(x).  fp32 in both modes, like cdist under autocast.
    Returns (loss[1], saved) -- no autograd, no framework kernel; koleo_bwd(saved, gscale) gives d(gscale * loss)/dx.

    Data parallel (``group`` with more than one rank): the neighbour of a row is searched over the GLOBAL batch, as the
    single-process reference would at that batch size.  Two all-gathers (unit rows; then index + distance per row) and no
    gradient collective: a row's gradient needs its own pair and the pairs that chose it, all of which are gathered data.
    The value returned is this rank's mean over its own rows, so the mean over ranks is the global loss.
    (koleo_begin / koleo_end are the two halves, for callers that have work to run under the first all-gather.)"""
    return koleo_end(koleo_begin(x, group), eps)


def koleo_bwd(saved, gscale: float = 1.0) -> Tensor:
    """dx [V, D] = d(gscale * koleo loss) / dx (the upstream factor is a host scalar here: loss weight / accumulation steps)."""
    xh_all, idx_all, d_all, norm, (row0, V, Vg, D), eps = saved
    dx = torch.empty((V, D), dtype=torch.float32, device=xh_all.device)
    check(lib.dinox_koleo_bwd(_p(xh_all), _p(idx_all), _p(d_all), _p(norm), row0, V, Vg, D, gscale / V, eps, 1e-12, _p(dx), _stream()),
          "dinox_koleo_bwd")
    return dx


class KoLeoFn(torch.autograd.Function):
    """koleo_fwd / koleo_bwd as an autograd node (the engine's fallback path and callers outside the engine)."""

    @staticmethod
    def forward(ctx, x, eps, group):
        loss, saved = koleo_fwd(x, eps, group)
        ctx.saved = saved
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        # the upstream gradient is a device scalar: run with gscale = 1 and scale the result (one small elementwise multiply, no host sync)
        return koleo_bwd(ctx.saved, 1.0) * g, None, None


def koleo_loss(x: Tensor, eps: float = 1e-8, group=None) -> Tensor:
    return KoLeoFn.apply(x, eps, group)


def adamw_hyper(lr: float, beta1: float, beta2: float, step_t: int) -> list:
    """The three per-step scalars of the optimiser pass -- [lr, 1/(1-beta1^t), 1/sqrt(1-beta2^t)] -- for dinox_adamw_ema_dev."""
    return [lr, 1.0 / (1.0 - beta1 ** step_t), 1.0 / math.sqrt(1.0 - beta2 ** step_t)]


def adamw_ema_(p: Tensor, g: Tensor, m: Tensor, v: Tensor, teacher: Optional[Tensor], *, lr: float, weight_decay: float,
               beta1: float, beta2: float, eps: float, step_t: int, ema: float, grad_scale: float = 1.0,
               hyper: Optional[Tensor] = None) -> Tensor:
    """Fused grad-norm + AdamW + EMA over flat fp32 arenas; returns gnorm_sq[1] (device).  With ``hyper`` (device float[3], see
    adamw_hyper) lr and the bias corrections are read from device memory: the launch is replayable from a captured hipGraph."""
    _need_cuda(p, g, m, v)
    for t in (p, g, m, v) + ((teacher,) if teacher is not None else ()):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() == p.numel()
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    ws = torch.empty(4096, dtype=torch.float32, device=p.device)
    if hyper is not None:
        assert hyper.is_cuda and hyper.dtype == torch.float32 and hyper.numel() == 3
        check(lib.dinox_adamw_ema_dev(_p(p), _p(g), _p(m), _p(v), _p(teacher), p.numel(), _p(hyper), weight_decay, beta1, beta2, eps, ema,
                                      grad_scale, _p(out), _p(ws), _stream()), "dinox_adamw_ema_dev")
        return out
    check(lib.dinox_adamw_ema(_p(p), _p(g), _p(m), _p(v), _p(teacher), p.numel(), lr, weight_decay, beta1, beta2, eps, step_t, ema,
                              grad_scale, _p(out), _p(ws), _stream()), "dinox_adamw_ema")
    return out


def sumsq(x: Tensor) -> Tensor:
    x = _c(x)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(4096, dtype=torch.float32, device=x.device)
    check(lib.dinox_sumsq(_p(x), x.numel(), _p(out), _p(ws), _stream()), "dinox_sumsq")
    return out


class GeluFn(torch.autograd.Function):
    """Stand-alone exact-erf GELU (fp32 math) for module paths that cannot use the fused epilogue."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        xf = _c(x.float())
        y = torch.empty_like(xf)
        check(lib.dinox_gelu_fwd(_p(xf), _p(y), xf.numel(), _stream()), "dinox_gelu_fwd")
        ctx.save_for_backward(xf)
        ctx.xdtype = x.dtype
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (xf,) = ctx.saved_tensors
        dyf = _c(dy.float())
        dx = torch.empty_like(xf)
        check(lib.dinox_gelu_bwd(_p(dyf), _p(xf), _p(dx), xf.numel(), _stream()), "dinox_gelu_bwd")
        return dx.to(ctx.xdtype)
