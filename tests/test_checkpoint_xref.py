"""Cross-implementation checkpoint parity, direction engine -> reference.

tests/golden/engine_checkpoint_00000003.pth was written on an MI355X by the drop-in CLI's save_checkpoint after three engine
steps (tests/test_gpu_parity.py::test_engine_written_checkpoint_payload, from ckpt_tiny.npz's initial state and batches);
engine_step4.json / engine_student_after_step4.pth hold the engine's step 4 after that point.  Here -- in the build container,
where the real reference is importable -- the REFERENCE's own load_checkpoint (scripts/phase5_big_run.py:1128-1193) restores
that file into reference modules + a stock torch.optim.AdamW, takes step 4 in the reference's loop order, and must land where
the engine did.  Skipped where /root/reference does not exist (the GPU box)."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, sub, t

REF = os.environ.get("DINOX_REFERENCE", "/root/reference")
CKPT = os.path.join(GOLDEN, "engine_checkpoint_00000003.pth")

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF) and os.path.exists(CKPT)),
                                reason="needs the reference checkout (build container only) and the engine-written fixture")


def _reference():
    sys.dont_write_bytecode = True
    for p in (REF, os.path.join(REF, "scripts")):
        if p not in sys.path:
            sys.path.append(p)
    for name in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "zoo" or k.startswith("zoo.") or k == "phase5_big_run"}
    sys.path.insert(0, REF)
    sys.path.insert(1, os.path.join(REF, "scripts"))
    try:
        import phase5_big_run as P          # the reference's script (imports the reference's zoo.arch)
        assert P.__file__.startswith(REF), P.__file__
        return P
    finally:
        sys.path.remove(REF)
        sys.path.remove(os.path.join(REF, "scripts"))
        for k in [k for k in sys.modules if k == "zoo" or k.startswith("zoo.") or k == "phase5_big_run"]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_engine_written_checkpoint_resumes_in_the_reference():
    from pathlib import Path
    P = _reference()
    g = load_golden("ckpt_tiny.npz")
    lr, min_lr, warm, max_steps, wd, ema, ts, tt, cm, gw = [float(v) for v in g["hp"]]
    kw = dict(img_size=28, patch=14, dim=32, depth=2, heads=2, mlp_ratio=4.0, use_grad_checkpoint=False, scale_aware=True)
    student = P.DinoStudentTeacher(P.PatchViT(**kw), out_dim=64)
    teacher = P.DinoStudentTeacher(P.PatchViT(**kw), out_dim=64)
    for p in teacher.parameters():
        p.requires_grad_(False)
    opt = torch.optim.AdamW(student.parameters(), lr=lr, weight_decay=wd)
    scaler = torch.amp.GradScaler("cpu", enabled=False)
    L = P.DINOLoss(64, center_momentum=cm)
    step, cfg = P.load_checkpoint(Path(CKPT), student, teacher, opt, scaler, L, torch.device("cpu"), scale_aware=True)
    assert step == 3 and cfg.model.dim == 32 and cfg.scale_aware
    # the restored state is the reference's own state after the same three steps (engine steps == reference steps, 1e-3)
    for k, v in sub(g, "student3").items():
        # (key bias: numerically-zero gradient, Adam moves it by +-lr with a round-off sign -> bounded by 3 steps x lr)
        assert torch.allclose(student.state_dict()[k], v, rtol=1e-3, atol=3.1e-3 if k.endswith("attn.qkv.bias") else 2e-5), k
    assert torch.allclose(L.center, t(g["center3"]), rtol=1e-4, atol=1e-7)
    assert float(opt.state_dict()["state"][0]["step"]) == 3.0
    # step 4 in the reference's loop order (:1692-1802)
    cur_lr = P.get_lr(3, int(max_steps), int(warm), lr, min_lr)
    for pg in opt.param_groups:
        pg["lr"] = cur_lr
    batch, sp2 = t(g["batch3"]), t(g["spacing3"])
    s_feats = student.backbone(batch, spacing=sp2)
    with torch.no_grad():
        t_feats = teacher.backbone(batch, spacing=sp2)
    loss = L(student.head(s_feats[:, 0]), teacher.head(t_feats[:, 0]), ts, tt) + gw * P.compute_gram_anchoring_loss(s_feats, t_feats)
    loss.backward()
    gn = sum(p.grad.norm(2).item() ** 2 for p in student.parameters() if p.grad is not None) ** 0.5
    opt.step()
    eng = json.load(open(os.path.join(GOLDEN, "engine_step4.json")))
    assert loss.item() == pytest.approx(eng["loss4"], rel=1e-3)
    assert gn == pytest.approx(eng["grad_norm4"], rel=1e-3)
    assert cur_lr == pytest.approx(eng["lr4"], rel=1e-12)
    after = torch.load(os.path.join(GOLDEN, "engine_student_after_step4.pth"), map_location="cpu", weights_only=True)
    for k, v in student.state_dict().items():
        assert torch.allclose(v, after[k], rtol=1e-3, atol=1.1e-3 if k.endswith("attn.qkv.bias") else 2e-5), k
