#!/usr/bin/env python3
"""Repeatability probe of the register-prefetch NT kernel: launch N times, report where results differ from the first launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = int(os.environ.get("M", 512 * 201)), int(os.environ.get("N", 1152)), 384
A = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
B = (torch.randn(N, K, device=dev, generator=g) * 0.5).bfloat16()
bias = torch.randn(N, device=dev, generator=g) if not os.environ.get("NOBIAS") else None
ref = (A.float() @ B.float().t() + (bias if bias is not None else 0)).bfloat16()
first = ops.gemm(A, B, bias=bias)
bad_ref = (first.float() - ref.float()).abs() > 0.25
print("first vs torch: mismatching elements", int(bad_ref.sum()))
nbad = 0
for i in range(int(os.environ.get("REPS", 40))):
    c = ops.gemm(A, B, bias=bias)
    d = c != first
    if d.any():
        nbad += 1
        idx = d.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        tm, tn = (rows // 128).unique(), (cols // 128).unique()
        print(f"launch {i}: {int(d.sum())} elems differ; tiles_m {tm[:8].tolist()} tiles_n {tn[:8].tolist()} rows%128 {(rows % 128).unique()[:16].tolist()} n={len((rows%128).unique())} cols%128 n={len((cols % 128).unique())} maxdiff {float((c.float()-first.float()).abs().max()):.3f}")
        wrong_c = (c.float() - ref.float()).abs() > 0.25
        wrong_f = (first.float() - ref.float()).abs() > 0.25
        print("   wrong-vs-torch in this launch:", int(wrong_c.sum()), " in first:", int(wrong_f.sum()))
        w = wrong_c if wrong_c.any() else wrong_f
        bad = c if wrong_c.any() else first
        wi = w.nonzero()
        r0, cset = int(wi[0, 0]), wi[:, 1].unique()
        print("   cols%128:", (cset % 128).tolist(), " rows:", int(wi[:, 0].min()), "..", int(wi[:, 0].max()))
        for cc in cset.tolist()[:4]:
            rr = wi[wi[:, 1] == cc][:, 0][:3]
            print("   col", cc, "got", bad[rr, cc].float().tolist(), "ref", ref[rr, cc].float().tolist())
print("launches differing:", nbad)
