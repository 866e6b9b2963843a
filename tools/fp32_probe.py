#!/usr/bin/env python3
"""fp32 (no --amp) step: this engine's exact-fp32 parity mode against the plain-PyTorch restatement of the reference loop on the same GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 64))
wl = bench.Workload(dev, 0, B=B, fp32=True)
for _ in range(2): wl.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): wl.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"dinox fp32 parity mode, bs {B}: {dt*1e3:.1f} ms/step = {B/dt:.0f} samples/s")
wl = None; torch.cuda.empty_cache()
from oracle import dinox_oracle as O      # (a measurement tool of the same kind as bench.py's baseline leg; not a product path)
cfg = O.VitCfg(out_dim=8192, img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
st = O.init_state(cfg, O.random_params(cfg, seed=0))
for name in ("student", "teacher", "adam_m", "adam_v"):
    setattr(st, name, {k: v.to(dev) for k, v in getattr(st, name).items()})
st.center = st.center.to(dev)
g = torch.Generator(device=dev).manual_seed(1)
batch = torch.randn(2 * B, 3, 224, 224, device=dev, generator=g); sp = torch.rand(B, 3, device=dev, generator=g) + 0.5
sp2 = torch.cat([sp, sp], 0); hp = O.HyperParams()
for _ in range(2): O.train_step(st, batch, sp2, hp)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): O.train_step(st, batch, sp2, hp)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"plain PyTorch-ROCm fp32 (oracle on cuda), bs {B}: {dt*1e3:.1f} ms/step = {B/dt:.0f} samples/s")
