"""Worker of test_engine_data_parallel_two_ranks_match_single_process: two TrainEngine steps (fp32 parity mode, DINO + Gram +
KoLeo, whose nearest neighbours span the global batch) on this rank's shard of a fixed global batch.  Run with RANK/WORLD_SIZE/MASTER_* set (WORLD_SIZE=1: whole batch)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]

from dinox.dp import init_process_group, shard_range  # noqa: E402
from dinox.engine import StepHyperParams, TrainEngine  # noqa: E402
import zoo.arch as arch  # noqa: E402

rank, world, local = init_process_group()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
accum = int(os.environ.get("DINOX_TEST_ACCUM") or 1)
ckpt = bool(os.environ.get("DINOX_TEST_GRAD_CKPT"))
crops = int(os.environ.get("DINOX_TEST_LOCAL_CROPS") or 0)        # L local 28-px crops per sample (multi-crop extension)
scale_aware = os.environ.get("DINOX_TEST_SCALE_AWARE", "1") != "0"
kw = dict(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=4, scale_aware=scale_aware, use_grad_checkpoint=ckpt)
torch.manual_seed(100 + rank)                      # different init per rank: the engine must broadcast rank 0's weights
if world == 1:
    torch.manual_seed(100)
student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 256)
if scale_aware:
    torch.nn.init.xavier_uniform_(student.backbone.scale_embed.mlp[2].weight)
teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 256)
teacher.load_state_dict(student.state_dict())
eng = TrainEngine(student.to(dev), teacher.to(dev), 256, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, koleo_weight=0.1),
                  bucket_bytes=64 << 10, accumulation_steps=accum)
student.train()
assert world == 1 or len(eng.bucketer.buckets) >= 3
if os.environ.get("DINOX_DP_FORCE_COLLECTIVES"):
    assert eng.bucketer.exchange and torch.distributed.get_backend() == os.environ.get("DINOX_EXPECT_BACKEND", torch.distributed.get_backend())
g = torch.Generator().manual_seed(7)
B = 8
v1, v2 = torch.randn(B, 3, 56, 56, generator=g), torch.randn(B, 3, 56, 56, generator=g)
sp = torch.rand(B, 3, generator=g) * 2 + 0.4
lo, hi = shard_range(B, rank, world)
batch = torch.cat([v1[lo:hi], v2[lo:hi]], 0).to(dev)
sp2 = torch.cat([sp[lo:hi], sp[lo:hi]], 0).to(dev) if scale_aware else None
loc = lsp = None
if crops:                                           # view-major local crops of this rank's samples
    lv = torch.randn(crops, B, 3, 28, 28, generator=g)
    loc = lv[:, lo:hi].reshape(-1, 3, 28, 28).to(dev)
    lsp = sp[lo:hi].repeat(crops, 1).to(dev) if scale_aware else None
for _ in range(2 * accum):                          # two optimiser steps
    eng.step(batch, sp2, loc, lsp)
sc = eng.scalars()
torch.save({"flat_p": eng.flat_p.cpu(), "center": eng.center.cpu(), "loss": sc["loss"], "grad_norm": sc["grad_norm"],
            "buckets": len(eng.bucketer.buckets), "fired_in_backward": eng.bucketer.fired_in_backward}, sys.argv[1])
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
