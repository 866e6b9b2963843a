#!/usr/bin/env python3
"""What does the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reach on the hot-path shapes?  A yardstick for DESIGN.md, not a code
path of the product (the product never calls a GEMM library).  Plain products only: the library has no GELU' / side-tensor / LayerNorm epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
M = int(os.environ.get("M", 102912))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def rb(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
D, H = 384, 1536
x, xh, x3 = rb(M, D), rb(M, H), rb(M, 3 * D)
wqkv, w1, w2, w1T, wqT = rb(3 * D, D), rb(H, D), rb(D, H), rb(D, H), rb(D, 3 * D)
cases = {
    "qkv  [M,384]x[384,1152]": (lambda: torch.matmul(x, wqkv.t()), lambda: ops.gemm(x, wqkv), 2 * M * D * 3 * D),
    "fc1  [M,384]x[384,1536]": (lambda: torch.matmul(x, w1.t()), lambda: ops.gemm(x, w1), 2 * M * D * H),
    "fc2  [M,1536]x[1536,384]": (lambda: torch.matmul(xh, w2.t()), lambda: ops.gemm(xh, w2), 2 * M * D * H),
    "dxn1 [M,1152]x[1152,384]": (lambda: torch.matmul(x3, wqT.t()), lambda: ops.gemm(x3, wqT), 2 * M * D * 3 * D),
    "dW1  [1536,M]x[M,384] (vendor bf16 out, dinox f32 out + column sums)": (lambda: torch.matmul(xh.t(), x), lambda: ops.gemm(xh, x, transA=True, transB=True, out_dtype=torch.float32), 2 * M * D * H),
}
R = int(os.environ.get("ROUNDS", 12))
for name, (lib, ours, fl) in cases.items():
    out = {}
    for tag, fn in (("vendor", lib), ("dinox", ours)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(R):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        out[tag] = ts[len(ts) // 2]
    print(f"{name:70s} vendor {out['vendor']:7.1f} us ({fl / out['vendor'] / 1e6:6.1f} TF/s)   dinox {out['dinox']:7.1f} us ({fl / out['dinox'] / 1e6:6.1f} TF/s)")
