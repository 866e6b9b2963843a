#!/usr/bin/env python3
"""Correctness screen + A/B timing of the full-row NT kernel for N = 384 (csrc/gemm_bf16_pp384.hip, DINOX_NT_PP384=1) against the shipped
kernels (gemm_bf16_nt_pp128 / _areg).  The switch is read per call: both run interleaved in ONE process.  CHECK=0 / TIME=0 skip a part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
rf = lambda *s: torch.randn(*s, device=dev, generator=g)


def run(on, fn):
    os.environ["DINOX_NT_PP384"] = "1" if on else "0"
    ops.TRACE_KERNELS = []
    out = fn()
    names, ops.TRACE_KERNELS = ops.TRACE_KERNELS, None
    return out, names


if os.environ.get("CHECK", "1") != "0":
    bad = 0
    for (M, K) in [(4096, 128), (4096 + 17, 384), (5000, 1152), (208 * 20 + 207, 1536), (208 * 21 + 1, 160), (208 * 30, 384), (70000, 384), (102912, 1536)]:
        a, w, bias = rb(M, K), rb(384, K), rf(384)
        ref = a.float() @ w.float().t()
        res = rf(M, 384)
        for name, fn, want, tol in (("plain_bf16", lambda: ops.gemm(a, w), ref, 6e-3), ("bias_bf16", lambda: ops.gemm(a, w, bias=bias), ref + bias, 6e-3),
                                    ("plain_f32", lambda: ops.gemm(a, w, out_dtype=torch.float32), ref, 2e-5),
                                    ("bias_res_f32", lambda: ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32), ref + bias + res, 2e-5),
                                    ("res_f32", lambda: ops.gemm(a, w, residual=res, out_dtype=torch.float32), ref + res, 2e-5)):
            (o1, n1), (o2, _) = run(True, fn), run(True, fn)
            err = float((o1.float() - want).abs().max() / want.abs().max())
            ok = n1 == ["gemm_bf16_nt_pp384"] and torch.equal(o1, o2) and err < tol
            bad += not ok
            print(f"  M {M:6d} K {K:4d} {name:10s} {n1} rel max err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    print("check:", "FAILED" if bad else "all ok")
    if bad:
        sys.exit(1)

if os.environ.get("TIME", "1") != "0":
    M = int(os.environ.get("M", 102912))
    cases = {}
    for K in (384, 1152, 1536):
        a, w = rb(M, K), rb(384, K)
        cases[f"dX K {K} plain bf16"] = (lambda a=a, w=w: ops.gemm(a, w), 2 * M * K * 384)
        if K != 1152:
            bias, res = rf(384), rf(M, 384)
            cases[f"{'fc2' if K > 384 else 'proj'} K {K} bias+res f32"] = (lambda a=a, w=w, bias=bias, res=res: ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32), 2 * M * K * 384)
    res = {}
    for r in range(7):
        for on in (False, True):
            for name, (fn, fl) in cases.items():
                run(on, fn)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                os.environ["DINOX_NT_PP384"] = "1" if on else "0"
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((name, on), []).append(e0.elapsed_time(e1) / 4 * 1e3)
    for name, (fn, fl) in cases.items():
        t0, t1 = sorted(res[(name, False)])[3], sorted(res[(name, True)])[3]
        print(f"  {name:22s} shipped {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TFLOP/s) | full-row tile {t1:7.1f} us ({fl / t1 / 1e6:5.0f} TFLOP/s)")
