import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
import zoo.arch as arch
from dinox.engine import StepHyperParams, TrainEngine
dev = torch.device("cuda", 0)
kw = dict(img_size=224, patch=14, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
torch.manual_seed(0)
s = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192); t = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
t.load_state_dict(s.state_dict())
eng = TrainEngine(s.to(dev), t.to(dev), 8192, StepHyperParams(max_steps=100, warmup_steps=5), amp_dtype=torch.bfloat16)
B = 128
x = torch.randn(2 * B, 3, 224, 224, device=dev); sp = torch.rand(2 * B, 3, device=dev) + 0.5
for _ in range(5): eng.step(x, sp)
torch.cuda.synchronize()
