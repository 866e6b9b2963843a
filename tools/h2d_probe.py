#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline config: the training step with its input batch handed over as a pinned HOST buffer
(serial copy, then step) and with the copy of batch i+1 overlapped on a side stream.  bench.py's `value` keeps inputs resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox.engine import StepHyperParams, TrainEngine
import zoo.arch as arch

B = 256
dev = torch.device("cuda", 0)
cfg_kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
torch.manual_seed(0)
student = arch.DinoStudentTeacher(arch.PatchViT(**cfg_kw), 8192)
torch.nn.init.xavier_uniform_(student.backbone.scale_embed.mlp[2].weight)
teacher = arch.DinoStudentTeacher(arch.PatchViT(**cfg_kw), 8192)
teacher.load_state_dict(student.state_dict())
eng = TrainEngine(student.to(dev), teacher.to(dev), 8192, StepHyperParams(max_steps=1000, warmup_steps=5), amp_dtype=torch.bfloat16)
g = torch.Generator().manual_seed(1234)
batch = torch.randn(2 * B, 3, 224, 224, generator=g).to(dev)
sp2 = (torch.rand(2 * B, 3, generator=g) + 0.5).to(dev)
host = batch.cpu().pin_memory()
dev_a, dev_b = torch.empty_like(batch), torch.empty_like(batch)
for _ in range(3):
    eng.step(batch, sp2)
torch.cuda.synchronize()
def timed(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = timed(lambda i: eng.step(batch, sp2))
def serial(i):
    dev_a.copy_(host, non_blocking=True)
    eng.step(dev_a, sp2)
ser = timed(serial)
side = torch.cuda.Stream()
bufs = [dev_a, dev_b]
ev = [torch.cuda.Event(), torch.cuda.Event()]
dev_a.copy_(host, non_blocking=True); torch.cuda.synchronize()
def overlapped(i):
    cur, nxt = bufs[i & 1], bufs[(i + 1) & 1]
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())     # nxt is free once the previous step that read it was enqueued before
        nxt.copy_(host, non_blocking=True)
        ev[(i + 1) & 1].record(side)
    eng.step(cur, sp2)
    torch.cuda.current_stream().wait_event(ev[(i + 1) & 1])
ovl = timed(overlapped)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dev_a.copy_(host, non_blocking=True); e1.record(); torch.cuda.synchronize()
gb = host.numel() * 4 / 1e9
print(f"batch {gb*1e3:.0f} MB pinned; H2D alone {e0.elapsed_time(e1):.2f} ms = {gb / e0.elapsed_time(e1) * 1e3:.1f} GB/s")
print(f"resident inputs : {res:.2f} ms/step = {B / res * 1e3:.0f} samples/s")
print(f"serial H2D+step : {ser:.2f} ms/step = {B / ser * 1e3:.0f} samples/s")
print(f"overlapped H2D  : {ovl:.2f} ms/step = {B / ovl * 1e3:.0f} samples/s")
