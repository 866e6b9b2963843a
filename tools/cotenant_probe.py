#!/usr/bin/env python3
"""What a co-tenant that holds CUs (RCCL's channel workgroups during a gradient all-reduce) does to the step's GEMMs: `HOG` workgroups of
tools/cu_hog.hip take one CU each on a second stream for ~2 ms while the product under test is launched beside them.
  hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/cu_hog.so tools/cu_hog.hip     (built by the caller; the .so travels with the snapshot)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "cu_hog.so"))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = "cuda"
M = 102912
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
x, xh, x3 = rb(M, 384), rb(M, 1536), rb(M, 1152)
wq, w2, wqT = rb(1152, 384), rb(384, 1536), rb(384, 1152)
res = torch.randn(M, 384, device=dev, generator=g)
db = torch.empty(1536, device=dev)
cases = {
    "qkv (nt_pp)": lambda: ops.gemm(x, wq),
    "fc2 (nt_pp128)": lambda: ops.gemm(xh, w2, residual=res, out_dtype=torch.float32),
    "dX K1152 (nt_pp128)": lambda: ops.gemm(x3, wqT),
    "dW1 (tn_big)": lambda: ops.gemm(xh, x, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db),
    "attention fwd": None,
}
qkv = rb(512, 201, 1152)
cases["attention fwd"] = lambda: ops.attention_fwd(qkv, 6)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for f in cases.values():
    for _ in range(3):
        f()
torch.cuda.synchronize()
hogs = [int(h) for h in os.environ.get("HOGS", "0,8,32").split(",")]
print(f"us per launch, alone and beside a co-tenant that holds H CUs (median of {int(os.environ.get('ROUNDS', 7))})")
for name, f in cases.items():
    line = f"  {name:22s}"
    for H in hogs:
        ts = []
        for r in range(int(os.environ.get("ROUNDS", 7))):
            torch.cuda.synchronize()
            if H:
                hog.cu_hog(H, 4_000_000, side.cuda_stream)            # ~2 ms of shader clocks
                torch.cuda._sleep(200_000)                            # let the co-tenant become resident first
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main); f(); e1.record(main)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        line += f" | H={H:3d}: {ts[len(ts) // 2]:7.1f}"
    print(line)
