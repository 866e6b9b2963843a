#!/usr/bin/env python3
"""Producer -> consumer pairs of the hot path, timed separately with HIP events in one process: proj + LayerNorm -> fc1 and fc2 + LayerNorm -> qkv,
the producer on the 128 x 384 kernel (DINOX_ROWLN_PP=0) and on the full-row kernel's LayerNorm epilogue (=1).  What the consumer costs depends
on where the producer left y (memory-side cache): fc1 221 us behind the former, 247-263 behind the latter, 278 with the cache flushed between."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
dev = "cuda"
M, D, H = 102912, 384, 1536
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
rf = lambda *s: torch.randn(*s, device=dev, generator=g)
o, wp, bp, res, gam, bet = rb(M, D), rb(D, D), rf(D), rf(M, D), rf(D), rf(D)
w1, b1 = rb(H, D), rf(H)
wq, bq = rb(3 * D, D), rf(3 * D)
act, w2, b2 = rb(M, H), rb(D, H), rf(D)
aux = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
res_t = {}
for r in range(6):
    for pp in ("0", "1"):
        os.environ["DINOX_ROWLN_PP"] = pp
        for name in ("proj->fc1", "fc2->qkv"):
            tp, tc = 0.0, 0.0
            for _ in range(3):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                if name == "proj->fc1":
                    x, y, mu, rs = ops.linear_residual_ln(o, wp, bp, res, gam, bet, 1e-5, torch.bfloat16)
                    e[1].record()
                    out = ops.gemm(y, w1, bias=b1, gelu=True, aux=aux, auxgrad=True)
                else:
                    x, y, mu, rs = ops.linear_residual_ln(act, w2, b2, res, gam, bet, 1e-5, torch.bfloat16)
                    e[1].record()
                    out = ops.gemm(y, wq, bias=bq)
                e[2].record()
                torch.cuda.synchronize()
                tp += e[0].elapsed_time(e[1]) * 1e3 / 3
                tc += e[1].elapsed_time(e[2]) * 1e3 / 3
            if r:
                res_t.setdefault((name, pp), []).append((tp, tc))
for k, v in res_t.items():
    v = sorted(v, key=lambda t: t[0] + t[1])
    print(k, "producer %.1f us consumer %.1f us" % v[len(v) // 2])
