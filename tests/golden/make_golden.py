#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REAL REFERENCE.

Run in the build container only (the reference checkout never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

It imports ``zoo.arch`` and ``scripts/phase5_big_run.py`` from ``/root/reference`` (the latter with
empty ``torchvision`` stub modules, because the image has no torchvision and the script only
touches it inside ``PngDataset``), runs them on CPU in fp32 with seeded inputs and writes *data
only* (inputs, weights, expected outputs, expected gradients) as ``.npz`` files next to this script.

Fixtures
  ops_attention.npz   reference Attention module   fwd + grads, N=201 tokens, d=64
  ops_mlp.npz         reference Mlp module         fwd + grads
  ops_scale_embed.npz reference ScaleEmbedding     fwd + grads (non-zero output projection)
  dino_loss.npz       DINOLoss (2 calls, centre carried), grad wrt student logits
  gram_loss.npz       compute_gram_anchoring_loss, grad wrt student feats
  koleo_loss.npz      KoLeoLoss (10 rows: cdist direct path; 40 rows: cdist matmul path), grad wrt input
  vit_tiny.npz        DinoStudentTeacher(PatchViT 56/14/64/2/2 regs4 scale-aware) fwd taps,
                      DINO+Gram loss, every parameter gradient
  vit_plain.npz       PatchViT 28/14/32/1/2 no registers, not scale-aware: fwd only
  step_tiny.npz       3 consecutive training steps of the reference loop order
                      (phase5_big_run.py:1741-1802) on a 28/14/32/2/2 model
  init_seed0.npz      raw initial state_dict under torch.manual_seed(0) (28/14/32/2/2 model)
  vit_tiny_autocast.npz  the vit_tiny case again with forward + losses under torch.autocast("cpu", bfloat16) -- the reference's
                      --amp arithmetic (phase5_big_run.py:1716-1717) -- feats, losses and every parameter gradient; two policies
                      for the DINO term (fp32 softmax like GPU autocast / CPU autocast as is).  Inputs + weights: vit_tiny.npz
  ref_checkpoint_00000003.pth  written by the REFERENCE's save_checkpoint (:1104-1125) after 3 steps of its loop order
  ckpt_tiny.npz       the batches of those steps, the state the checkpoint must restore, and step 4 as the reference runs it after
                      its own load_checkpoint (:1128-1193) into fresh modules + a fresh torch.optim.AdamW
  get_lr.npz          get_lr at a grid of (step,total)
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("DINOX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "scripts"))
for name in ("torchvision", "torchvision.transforms"):
    sys.modules.setdefault(name, types.ModuleType(name))

import zoo.arch as A            # noqa: E402  (the reference)
import phase5_big_run as P      # noqa: E402  (the reference)

torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(t) else t


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def perturb_(module: torch.nn.Module, g: torch.Generator) -> None:
    """Make every parameter non-trivial (reference init has zero biases, unit LN, zero-init
    scale-embed output projection) so the fixtures exercise every term."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("mlp.2.weight") and "scale_embed" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))


def sd_arrays(prefix: str, sd) -> dict:
    return {f"{prefix}/{k}": v.detach().clone() for k, v in sd.items()}  # clone: params mutate in place later


# ------------------------------------------------------------------------------------------
def ops_attention():
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    m = A.Attention(128, num_heads=2)
    perturb_(m, g)
    x = torch.randn(2, 201, 128, generator=g).requires_grad_(True)
    dy = torch.randn(2, 201, 128, generator=g)
    y = m(x)
    y.backward(dy)
    out = {"x": x, "dy": dy, "y": y, "dx": x.grad, "heads": np.int64(2)}
    out.update(sd_arrays("w", m.state_dict()))
    out.update({f"g/{n}": p.grad for n, p in m.named_parameters()})
    save("ops_attention.npz", **out)


def ops_mlp():
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)
    m = A.Mlp(64, 4.0)
    perturb_(m, g)
    x = (2.0 * torch.randn(3, 37, 64, generator=g)).requires_grad_(True)
    dy = torch.randn(3, 37, 64, generator=g)
    y = m(x)
    y.backward(dy)
    out = {"x": x, "dy": dy, "y": y, "dx": x.grad}
    out.update(sd_arrays("w", m.state_dict()))
    out.update({f"g/{n}": p.grad for n, p in m.named_parameters()})
    save("ops_mlp.npz", **out)


def ops_scale_embed():
    g = torch.Generator().manual_seed(13)
    torch.manual_seed(13)
    m = A.ScaleEmbedding(64)
    torch.nn.init.xavier_uniform_(m.mlp[2].weight)
    perturb_(m, g)
    sp = torch.tensor([[0.5, 0.5, 1.0], [1.5, 1.5, 5.0], [0.7, 0.9, 2.5], [0.46, 0.98, 0.625], [1.0, 1.0, 1.0]]).requires_grad_(True)
    dy = torch.randn(5, 1, 64, generator=g)
    y = m(sp)
    y.backward(dy)
    out = {"spacing": sp, "dy": dy, "y": y, "dspacing": sp.grad}
    out.update(sd_arrays("w", m.state_dict()))
    out.update({f"g/{n}": p.grad for n, p in m.named_parameters()})
    save("ops_scale_embed.npz", **out)


def dino_loss():
    g = torch.Generator().manual_seed(21)
    B2, K = 12, 512
    s = (3.0 * torch.randn(B2, K, generator=g)).requires_grad_(True)
    t = 1.5 * torch.randn(B2, K, generator=g)
    c0 = 0.1 * torch.randn(1, K, generator=g)
    L = P.DINOLoss(K, center_momentum=0.9)
    L.center.copy_(c0)
    l1 = L(s, t, 0.1, 0.04)
    l1.backward()
    ds1 = s.grad.clone()
    c1 = L.center.clone()
    t2 = 1.5 * torch.randn(B2, K, generator=g)
    l2 = L(s.detach(), t2, 0.1, 0.04)
    c2 = L.center.clone()
    save("dino_loss.npz", s=s, t=t, t2=t2, center0=c0, loss1=l1, ds1=ds1, center1=c1, loss2=l2, center2=c2,
         student_temp=np.float32(0.1), teacher_temp=np.float32(0.04), momentum=np.float32(0.9))


def gram_loss():
    g = torch.Generator().manual_seed(22)
    sf = torch.randn(3, 41, 32, generator=g).requires_grad_(True)
    tf = torch.randn(3, 41, 32, generator=g)
    with torch.no_grad():
        sf[1, 5] = 0.0          # a zero-norm token exercises the eps clamp of F.normalize
    l = P.compute_gram_anchoring_loss(sf, tf)
    l.backward()
    save("gram_loss.npz", sf=sf, tf=tf, loss=l, dsf=sf.grad, gram_s=P.compute_gram_matrix(sf.detach()[:, 1:]))


def koleo_loss():
    """KoLeoLoss (phase5_big_run.py:742-773) on the student head output: a 10-row case (torch.cdist's direct path) and a
    40-row case (> 25 rows: its matmul path), each with the gradient wrt the input."""
    g = torch.Generator().manual_seed(23)
    out = {}
    for tag, (V, K) in {"small": (10, 48), "mm": (40, 256)}.items():
        x = (2.0 * torch.randn(V, K, generator=g)).requires_grad_(True)
        with torch.no_grad():
            x[3] = x[7] + 0.05 * torch.randn(K, generator=g)      # a close pair: both rows pick each other
        l = P.KoLeoLoss()(x)
        l.backward()
        out.update({f"{tag}_x": x, f"{tag}_loss": l, f"{tag}_dx": x.grad})
    save("koleo_loss.npz", **out)


def _vit_tiny_setup():
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(0)
    cfg = dict(img_size=56, patch=14, dim=64, depth=2, heads=2, mlp_ratio=4.0, num_registers=4, scale_aware=True)
    student = A.DinoStudentTeacher(A.PatchViT(**cfg), out_dim=128)
    perturb_(student, g)
    teacher = A.DinoStudentTeacher(A.PatchViT(**cfg), out_dim=128)
    teacher.load_state_dict(student.state_dict())
    with torch.no_grad():
        for p in teacher.parameters():
            p.add_(0.01 * torch.randn(p.shape, generator=g))
            p.requires_grad_(False)
    x = torch.randn(4, 3, 56, 56, generator=g)
    sp = torch.tensor([[0.5, 0.5, 1.0], [1.5, 1.5, 5.0]])
    sp2 = torch.cat([sp, sp], 0)
    center = 0.05 * torch.randn(1, 128, generator=g)
    return student, teacher, x, sp2, center


def vit_tiny():
    student, teacher, x, sp2, center = _vit_tiny_setup()

    taps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=f"block{j}": taps.__setitem__(k, o.detach().clone()))
             for j, blk in enumerate(student.backbone.blocks)]
    s_feats = student.backbone(x, spacing=sp2)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        t_feats = teacher.backbone(x, spacing=sp2)
    s_out = student.head(s_feats[:, 0])
    t_out = teacher.head(t_feats[:, 0])
    L = P.DINOLoss(128, center_momentum=0.9)
    L.center.copy_(center)
    l_dino = L(s_out, t_out, 0.1, 0.04)
    l_gram = P.compute_gram_anchoring_loss(s_feats, t_feats)
    loss = l_dino + 1.0 * l_gram
    loss.backward()
    # forward with spacing=None on the scale-aware model is a no-op for the embedding (arch.py:224)
    with torch.no_grad():
        feats_nospacing = student.backbone(x, spacing=None)
    out = dict(x=x, spacing=sp2, center=center, s_feats=s_feats, t_feats=t_feats, s_out=s_out, t_out=t_out,
               loss_dino=l_dino, loss_gram=l_gram, loss=loss, center_after=L.center, feats_nospacing=feats_nospacing,
               cfg=np.array([56, 14, 64, 2, 2, 4, 1, 128], dtype=np.int64))
    out.update({f"tap/{k}": v for k, v in taps.items()})
    out.update(sd_arrays("student", student.state_dict()))
    out.update(sd_arrays("teacher", teacher.state_dict()))
    out.update({f"grad/{n}": p.grad for n, p in student.named_parameters()})
    out["param_order"] = np.array([n for n, _ in student.named_parameters()])
    save("vit_tiny.npz", **out)


def vit_tiny_autocast():
    """The vit_tiny case with the forward and the losses under torch.autocast(bfloat16), as the reference's --amp path runs them
    (phase5_big_run.py:1716-1717, backward outside the context :1772).  The reference enables autocast on its GPU only; the
    context is device-generic and this container has no GPU, so it is entered for "cpu".  CPU and GPU autocast share the policy
    for every op of the model (bf16 linear / conv / SDPA / bmm with bf16 results, GELU in the dtype it is handed, LayerNorm of
    the fp32 residual stream in fp32, mse_loss fp32) and differ in the DINO term: GPU autocast runs softmax / log_softmax in fp32,
    CPU autocast leaves them in the head's bf16.  Both are recorded: "f32loss/" hands the loss fp32 copies of the head outputs
    (= the GPU policy, the tighter reference for the bf16 gates), "cpu/" is CPU autocast as is."""
    out = {}
    for tag, loss_fp32 in (("f32loss", True), ("cpu", False)):
        student, teacher, x, sp2, center = _vit_tiny_setup()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            s_feats = student.backbone(x, spacing=sp2)
            with torch.no_grad():
                t_feats = teacher.backbone(x, spacing=sp2)
            s_out = student.head(s_feats[:, 0])
            t_out = teacher.head(t_feats[:, 0])
            L = P.DINOLoss(128, center_momentum=0.9)
            L.center.copy_(center)
            l_dino = L(s_out.float(), t_out.float(), 0.1, 0.04) if loss_fp32 else L(s_out, t_out, 0.1, 0.04)
            l_gram = P.compute_gram_anchoring_loss(s_feats, t_feats)
            loss = l_dino + 1.0 * l_gram
        loss.backward()
        out.update({f"{tag}/s_feats": s_feats.float(), f"{tag}/t_feats": t_feats.float(), f"{tag}/s_out": s_out.float(),
                    f"{tag}/t_out": t_out.float(), f"{tag}/loss_dino": l_dino.float(), f"{tag}/loss_gram": l_gram.float(),
                    f"{tag}/loss": loss.float()})
        out.update({f"{tag}/grad/{n}": p.grad.float() for n, p in student.named_parameters()})
        out[f"{tag}/dtypes"] = np.array([str(s_feats.dtype), str(s_out.dtype), str(l_dino.dtype), str(l_gram.dtype)])
    save("vit_tiny_autocast.npz", **out)


def ckpt_tiny():
    """Cross-implementation checkpoint parity.  Three steps in the reference's loop order, then the REFERENCE's
    save_checkpoint writes ref_checkpoint_00000003.pth; the reference's load_checkpoint restores it into fresh modules and a
    fresh AdamW, and step 4 runs from there.  The engine must restore the same file through the CLI's load_checkpoint and
    reproduce step 4.  Model: what the reference's main() builds for --vit-patch 14 --vit-dim 32 --vit-depth 2 --vit-heads 2
    --out-dim 64 --img-size 28 --scale-aware (:1594-1608: registers default 4)."""
    import tempfile
    from pathlib import Path
    g = torch.Generator().manual_seed(51)
    torch.manual_seed(0)
    mc = P.ModelConfig(name="custom", patch=14, dim=32, depth=2, heads=2, mlp_ratio=4.0, out_dim=64)
    hp = dict(lr=1e-3, min_lr=1e-5, warmup=2, max_steps=10, wd=0.04, ema=0.9, ts=0.1, tt=0.04, cm=0.9, gw=1.0)
    tc = P.TrainingConfig(model=mc, img_size=28, batch_size=3, lr=hp["lr"], min_lr=hp["min_lr"], warmup_steps=hp["warmup"],
                          weight_decay=hp["wd"], max_steps=hp["max_steps"], ema=hp["ema"], teacher_temp=hp["tt"], student_temp=hp["ts"],
                          center_momentum=hp["cm"], gram_weight=hp["gw"], scale_aware=True, created_at="2026-01-01 00:00:00 UTC")

    def build():
        kw = dict(img_size=28, patch=mc.patch, dim=mc.dim, depth=mc.depth, heads=mc.heads, mlp_ratio=mc.mlp_ratio,
                  use_grad_checkpoint=False, scale_aware=True)
        s = P.DinoStudentTeacher(P.PatchViT(**kw), out_dim=mc.out_dim)
        t = P.DinoStudentTeacher(P.PatchViT(**kw), out_dim=mc.out_dim)
        return s, t

    def one_step(step, student, teacher, opt, L, batch, sp2):
        lr = P.get_lr(step, hp["max_steps"], hp["warmup"], hp["lr"], hp["min_lr"])
        for pg in opt.param_groups:
            pg["lr"] = lr
        s_feats = student.backbone(batch, spacing=sp2)
        with torch.no_grad():
            t_feats = teacher.backbone(batch, spacing=sp2)
        s_out = student.head(s_feats[:, 0])
        t_out = teacher.head(t_feats[:, 0])
        loss = L(s_out, t_out, hp["ts"], hp["tt"]) + hp["gw"] * P.compute_gram_anchoring_loss(s_feats, t_feats)
        loss.backward()
        gn = sum(p.grad.detach().norm(2).item() ** 2 for p in student.parameters() if p.grad is not None) ** 0.5
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for ps, pt in zip(student.parameters(), teacher.parameters()):
                pt.data.mul_(hp["ema"]).add_(ps.data, alpha=1.0 - hp["ema"])
        return loss.item(), gn, lr

    student, teacher = build()
    perturb_(student, g)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad_(False)
    opt = torch.optim.AdamW(student.parameters(), lr=hp["lr"], weight_decay=hp["wd"])
    scaler = torch.amp.GradScaler("cpu", enabled=False)
    L = P.DINOLoss(mc.out_dim, center_momentum=hp["cm"])
    out = dict(hp=np.array([hp[k] for k in ("lr", "min_lr", "warmup", "max_steps", "wd", "ema", "ts", "tt", "cm", "gw")], dtype=np.float64))
    out.update(sd_arrays("init", student.state_dict()))
    batches = []
    for step in range(4):
        v = torch.randn(2 * 3, 3, 28, 28, generator=g)
        sp = torch.rand(3, 3, generator=g) * 2 + 0.4
        batches.append((v, torch.cat([sp, sp], 0)))
        out[f"batch{step}"], out[f"spacing{step}"] = batches[-1]
    losses = [one_step(i, student, teacher, opt, L, *batches[i]) for i in range(3)]
    path = Path(HERE) / "ref_checkpoint_00000003.pth"
    P.save_checkpoint(path, 3, student, teacher, opt, scaler, L, tc)                      # <- the reference writes the file
    out.update(sd_arrays("student3", student.state_dict()))
    out.update(sd_arrays("teacher3", teacher.state_dict()))
    out["center3"] = L.center.clone()
    # resume in the reference: fresh modules, fresh optimiser, its own load_checkpoint, then step 4 (index 3)
    s2, t2 = build()
    for p in t2.parameters():
        p.requires_grad_(False)
    opt2 = torch.optim.AdamW(s2.parameters(), lr=hp["lr"], weight_decay=hp["wd"])
    L2 = P.DINOLoss(mc.out_dim, center_momentum=hp["cm"])
    step0, cfg_back = P.load_checkpoint(path, s2, t2, opt2, scaler, L2, torch.device("cpu"), scale_aware=True)
    assert step0 == 3 and cfg_back.model.dim == 32
    l4 = one_step(3, s2, t2, opt2, L2, *batches[3])
    # (the uninterrupted run gives the same step 4: resume is exact in the reference)
    l4b = one_step(3, student, teacher, opt, L, *batches[3])
    assert abs(l4[0] - l4b[0]) < 1e-6 * abs(l4b[0]), (l4, l4b)
    out.update(sd_arrays("student4", s2.state_dict()))
    out.update(sd_arrays("teacher4", t2.state_dict()))
    out["center4"] = L2.center.clone()
    out["losses"] = np.array([l[0] for l in losses] + [l4[0]], dtype=np.float64)
    out["grad_norms"] = np.array([l[1] for l in losses] + [l4[1]], dtype=np.float64)
    out["lrs"] = np.array([l[2] for l in losses] + [l4[2]], dtype=np.float64)
    out["adam_step4"] = np.float64(float(opt2.state_dict()["state"][0]["step"]))
    np.savez_compressed(os.path.join(HERE, "ckpt_tiny.npz"), **{k: (npy(v) if torch.is_tensor(v) else v) for k, v in out.items()})
    print(f"ckpt_tiny.npz + {path.name} ({path.stat().st_size / 1024:.1f} KiB): losses={out['losses']}")


def vit_plain():
    g = torch.Generator().manual_seed(32)
    torch.manual_seed(1)
    m = A.PatchViT(img_size=28, patch=14, dim=32, depth=1, heads=2, mlp_ratio=2.0, num_registers=0, scale_aware=False)
    perturb_(m, g)
    x = torch.randn(3, 3, 28, 28, generator=g)
    with torch.no_grad():
        y = m(x)
    out = dict(x=x, y=y, cfg=np.array([28, 14, 32, 1, 2, 0, 0, 0], dtype=np.int64))
    out.update(sd_arrays("w", m.state_dict()))
    save("vit_plain.npz", **out)


def step_tiny():
    """Three steps in the exact order of phase5_big_run.py:1692-1802 (loss_type=dino, accumulation 1,
    koleo 0, AMP off -> pure fp32 CPU), driven by this harness because main() needs a dataset."""
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(0)
    cfg = dict(img_size=28, patch=14, dim=32, depth=2, heads=2, mlp_ratio=4.0, num_registers=2, scale_aware=True)
    out_dim, B = 64, 3
    student = A.DinoStudentTeacher(A.PatchViT(**cfg), out_dim=out_dim)
    perturb_(student, g)
    teacher = A.DinoStudentTeacher(A.PatchViT(**cfg), out_dim=out_dim)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad_(False)
    hp = dict(lr=1e-3, min_lr=1e-5, warmup=2, max_steps=10, wd=0.04, ema=0.9, ts=0.1, tt=0.04, cm=0.9, gw=1.0)
    opt = torch.optim.AdamW(student.parameters(), lr=hp["lr"], weight_decay=hp["wd"])
    L = P.DINOLoss(out_dim, center_momentum=hp["cm"])
    out = dict(cfg=np.array([28, 14, 32, 2, 2, 2, 1, out_dim], dtype=np.int64),
               hp=np.array([hp[k] for k in ("lr", "min_lr", "warmup", "max_steps", "wd", "ema", "ts", "tt", "cm", "gw")], dtype=np.float64))
    out.update(sd_arrays("init", student.state_dict()))
    losses, gns, lrs, dinos, grams = [], [], [], [], []
    for step in range(3):
        lr = P.get_lr(step, hp["max_steps"], hp["warmup"], hp["lr"], hp["min_lr"])
        for pg in opt.param_groups:
            pg["lr"] = lr
        v1 = torch.randn(B, 3, 28, 28, generator=g)
        v2 = torch.randn(B, 3, 28, 28, generator=g)
        sp = torch.rand(B, 3, generator=g) * 2 + 0.4
        batch = torch.cat([v1, v2], 0)
        sp2 = torch.cat([sp, sp], 0)
        out[f"batch{step}"] = batch
        out[f"spacing{step}"] = sp2
        s_feats = student.backbone(batch, spacing=sp2)
        with torch.no_grad():
            t_feats = teacher.backbone(batch, spacing=sp2)
        s_out = student.head(s_feats[:, 0])
        t_out = teacher.head(t_feats[:, 0])
        l_dino = L(s_out, t_out, hp["ts"], hp["tt"])
        l_gram = P.compute_gram_anchoring_loss(s_feats, t_feats)
        loss = l_dino + hp["gw"] * l_gram
        loss.backward()
        tot = 0.0
        for p in student.parameters():
            if p.grad is not None:
                tot += p.grad.detach().norm(2).item() ** 2
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for ps, pt in zip(student.parameters(), teacher.parameters()):
                pt.data.mul_(hp["ema"]).add_(ps.data, alpha=1.0 - hp["ema"])
        losses.append(loss.item()); gns.append(tot ** 0.5); lrs.append(lr)
        dinos.append(l_dino.item()); grams.append(l_gram.item())
        if step == 0:
            out.update(sd_arrays("student1", student.state_dict()))
    out.update(sd_arrays("student3", student.state_dict()))
    out.update(sd_arrays("teacher3", teacher.state_dict()))
    out["center3"] = L.center
    out["losses"] = np.array(losses, dtype=np.float64)
    out["dinos"] = np.array(dinos, dtype=np.float64)
    out["grams"] = np.array(grams, dtype=np.float64)
    out["grad_norms"] = np.array(gns, dtype=np.float64)
    out["lrs"] = np.array(lrs, dtype=np.float64)
    path = os.path.join(HERE, "step_tiny.npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else v) for k, v in out.items()})
    print(f"step_tiny.npz: {os.path.getsize(path) / 1024:.1f} KiB; losses={losses} gn={gns}")


def init_seed0():
    """Raw reference initialisation under torch.manual_seed(0): pins the drop-in's init call sequence."""
    torch.manual_seed(0)
    m = A.DinoStudentTeacher(A.PatchViT(img_size=28, patch=14, dim=32, depth=2, heads=2, num_registers=2, scale_aware=True), out_dim=64)
    save("init_seed0.npz", **sd_arrays("sd", m.state_dict()))


def get_lr_grid():
    rows = []
    for total in (None, 10, 5000):
        for step in (0, 1, 2, 3, 9, 10, 11, 2499, 2500, 3750, 4999, 5000, 9999):
            for warm in (2, 2500):
                rows.append((step, -1 if total is None else total, warm, P.get_lr(step, total, warm, 1e-4, 1e-6)))
    np.savez_compressed(os.path.join(HERE, "get_lr.npz"), rows=np.array(rows, dtype=np.float64))
    print(f"get_lr.npz: {len(rows)} rows")


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    ops_attention()
    ops_mlp()
    ops_scale_embed()
    dino_loss()
    gram_loss()
    koleo_loss()
    vit_tiny()
    vit_plain()
    step_tiny()
    init_seed0()
    get_lr_grid()
    vit_tiny_autocast()
    ckpt_tiny()
