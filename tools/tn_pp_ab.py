#!/usr/bin/env python3
"""The dW products (C[M,N] = A[K,M]^T B[K,N], K = tokens) on gemm_bf16_tn_big with its two K loops: DINOX_TN_PP=0 (every wave in
step, one barrier per K-step) against the default (two wave groups in anti-phase).  Interleaved rounds, HIP events; the two loops add
the same products in the same order, so the results must be bit-equal.  MODEL=S|L|B|G."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

dev = "cuda"
for model in os.environ.get("MODEL", "S,L").split(","):
    D, H, K = {"S": (384, 1536, 102912), "B": (768, 3072, 51456), "L": (1024, 4096, 51456), "G": (1408, 6144, 12864)}[model]
    g = torch.Generator(device=dev).manual_seed(0)
    rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
    x, xh, x3 = rb(K, D), rb(K, H), rb(K, 3 * D)
    shapes = {"dW1  [H x D]": (xh, x), "dW2  [D x H]": (x, xh), "dWqkv[3D x D]": (x3, x), "dWp  [D x D]": (x, x)}
    rounds = int(os.environ.get("ROUNDS", 7))
    res, ref = {}, {}
    for pp in ("0", "1", "2"):
        os.environ["DINOX_TN_PP"] = pp
        for name, (a, b) in shapes.items():
            db = torch.empty(a.shape[1], device=dev)
            out = ops.gemm(a, b, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
            if name not in ref:
                ref[name] = (out.clone(), db.clone())
            elif pp == "1":
                assert torch.equal(out, ref[name][0]) and torch.equal(db, ref[name][1]), (model, name, "the two K loops differ")
            else:          # the other MFMA shape adds 32 products at a time instead of 16: equal to fp32 rounding of a K-long sum
                err = float((out - ref[name][0]).abs().max()) / float(ref[name][0].abs().max())
                errb = float((db - ref[name][1]).abs().max()) / float(ref[name][1].abs().max())
                assert err < 2e-5 and errb < 2e-5, (model, name, err, errb)
    for r in range(rounds):
        for pp in ("0", "1", "2"):
            os.environ["DINOX_TN_PP"] = pp
            for name, (a, b) in shapes.items():
                db = torch.empty(a.shape[1], device=dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(a, b, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((name, pp), []).append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f"model {model}: D {D} H {H} K {K}; us per product incl. the split reduction (median of {rounds} rounds) | TFLOP/s; results bit-equal")
    for name, (a, b) in shapes.items():
        fl = 2 * K * a.shape[1] * b.shape[1]
        t0, t1, t2 = (sorted(res[(name, pp)])[rounds // 2] for pp in ("0", "1", "2"))
        print(f"  {name:14s} in step {t0:7.1f} ({fl / t0 / 1e6:5.0f}) | anti-phase 32x32x16 {t1:7.1f} ({fl / t1 / 1e6:5.0f}) | anti-phase 16x16x32 {t2:7.1f} ({fl / t2 / 1e6:5.0f})")
