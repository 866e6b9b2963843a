#!/usr/bin/env python3
"""dX product + LayerNorm backward: two launches (dinox_gemm on the full-row kernel + dinox_layernorm_bwd) against dinox_linear_ln_bwd,
at the hot-path size, interleaved in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
dev = "cuda"
M, N = int(os.environ.get("M", 102912)), 384
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for K in (1152, 1536):
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    Wt = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    x = torch.randn(M, N, device=dev, generator=g)
    gam = torch.randn(N, device=dev, generator=g)
    mean, rstd = x.mean(-1), 1 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)
    gin = torch.randn(M, N, device=dev, generator=g)
    for r in range(7):
        for fused in (False, True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                if fused:
                    ops.linear_ln_bwd(A, Wt, x, gam, mean, rstd, dx_add=gin, want_lowp=True)
                else:
                    dy = ops.gemm(A, Wt)
                    ops.layernorm_bwd(dy, x, gam, mean, rstd, dx_add=gin, want_lowp=True)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res.setdefault((K, fused), []).append(e0.elapsed_time(e1) / 3 * 1e3)
    t0, t1 = sorted(res[(K, False)])[3], sorted(res[(K, True)])[3]
    print(f"K {K}: two launches {t0:.1f} us | fused {t1:.1f} us")
