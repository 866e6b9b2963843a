import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
dev = "cuda"
os.environ["DINOX_TN_PP"] = "1"
D, H, K = 384, 1536, 102912
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
x, xh, x3 = rb(K, D), rb(K, H), rb(K, 3 * D)
shapes = {"dW1": (xh, x), "dW2": (x, xh), "dWqkv": (x3, x)}
names = {0: "full", 1: "no loop requests", 2: "no MFMAs", 4: "no fragment reads", 5: "no requests, no reads (MFMA + barriers)", 6: "no MFMA, no reads (requests + waits + barriers)", 7: "barriers only", 8: "no loop", 16: "no epilogue", 23: "barriers only, no epilogue", 24: "no loop, no epilogue"}
if os.environ.get("ONLY"):
    names = {int(os.environ["ONLY"]): names[int(os.environ["ONLY"])]}
res = {}
for r in range(5):
    for d in names:
        os.environ["DINOX_TN_DBG"] = str(d)
        for name, (a, b) in shapes.items():
            db = torch.empty(a.shape[1], device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, b, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, d), []).append(e0.elapsed_time(e1) / 4 * 1e3)
for name in shapes:
    print(name, " | ".join(f"{names[d]} {sorted(res[(name, d)])[2]:.1f}" for d in names))
