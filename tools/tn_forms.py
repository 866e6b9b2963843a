#!/usr/bin/env python3
"""The dW products (C[M,N] = A[K,M]^T B[K,N], K = tokens) on each tile shape of gemm_bf16_tn_big (DINOX_TN_FORM = 1: 256 x 192,
2: 384 x 128, 3: 256 x 256) and on the shape the plan picks by itself; interleaved rounds, HIP events.  MODEL=S|L|B|G."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

dev = "cuda"
model = os.environ.get("MODEL", "L")
D, H, K = {"S": (384, 1536, 102912), "B": (768, 3072, 51456), "L": (1024, 4096, 51456), "G": (1408, 6144, 12864)}[model]
K = int(os.environ.get("K", K))
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
x, xh, x3 = rb(K, D), rb(K, H), rb(K, 3 * D)
shapes = {"dW1  [H x D]": (xh, x), "dW2  [D x H]": (x, xh), "dWqkv[3D x D]": (x3, x), "dWp  [D x D]": (x, x)}
rounds = int(os.environ.get("ROUNDS", 5))
res = {}
ref = {}
for form in ("1", "2", "3", ""):
    for name, (a, b) in shapes.items():
        if form:
            os.environ["DINOX_TN_FORM"] = form
        else:
            os.environ.pop("DINOX_TN_FORM", None)
        db = torch.empty(a.shape[1], device=dev)
        out = ops.gemm(a, b, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
        if name not in ref:
            ref[name] = (out.clone(), db.clone())
        else:      # every shape computes the same sums in another order: equal to fp32 rounding of a K-long sum
            err = float((out - ref[name][0]).abs().max()) / float(ref[name][0].abs().max())
            assert err < 2e-5 and float((db - ref[name][1]).abs().max()) / float(ref[name][1].abs().max()) < 2e-5, (name, form, err)
for r in range(rounds):
    for form in ("1", "2", "3", ""):
        for name, (a, b) in shapes.items():
            if form:
                os.environ["DINOX_TN_FORM"] = form
            else:
                os.environ.pop("DINOX_TN_FORM", None)
            db = torch.empty(a.shape[1], device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, b, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, form), []).append(e0.elapsed_time(e1) / 4 * 1e3)
print(f"model {model}: D {D} H {H} K {K}; us per product incl. the split reduction (median of {rounds} rounds) | TFLOP/s")
for name, (a, b) in shapes.items():
    fl = 2 * K * a.shape[1] * b.shape[1]
    row = []
    for form in ("1", "2", "3", ""):
        t = sorted(res[(name, form)])[rounds // 2]
        row.append(f"{'auto' if not form else 'form' + form} {t:7.1f} ({fl / t / 1e6:5.0f})")
    print(f"  {name:14s} " + " | ".join(row))
