#!/usr/bin/env python3
"""Does the MLP pair fc1 -> fc2 run faster when the token dimension is split so that fc1's activation (316 MB at bs256) is still in
the last-level cache when fc2 reads it?  Times fc1 + fc2 over all M tokens against 2 and 4 row chunks (same kernels, same bytes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
M, D, H = 512 * 201, 384, 1536
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(M, D, device=dev, generator=g) * 0.5).bfloat16()
w1 = (torch.randn(H, D, device=dev, generator=g) * 0.05).bfloat16()
w2 = (torch.randn(D, H, device=dev, generator=g) * 0.05).bfloat16()
b1, b2 = torch.randn(H, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
res = torch.randn(M, D, device=dev, generator=g)
act = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
aux = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
out = torch.empty(M, D, device=dev)

def run(chunks, with_aux):
    step = (M // chunks + 127) // 128 * 128
    for r0 in range(0, M, step):
        r1 = min(M, r0 + step)
        ops.gemm(x[r0:r1], w1, bias=b1, gelu=True, aux=aux[r0:r1] if with_aux else None, auxgrad=with_aux, out=act[r0:r1])
        ops.gemm(act[r0:r1], w2, bias=b2, residual=res[r0:r1], out=out[r0:r1], out_dtype=torch.float32)

def t(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]

for with_aux in (False, True):
    print("student (side tensor written)" if with_aux else "teacher (no side tensor)")
    for chunks in (1, 2, 4, 8):
        print(f"  {chunks} chunk(s): fc1 + fc2 = {t(lambda: run(chunks, with_aux)):7.1f} us")
