"""Training engine: one DINO-X optimiser step on the HIP kernels, optionally data-parallel.

Follows the order of the reference loop (scripts/phase5_big_run.py:1692-1802) for
``--loss-type dino`` with ``accumulation_steps == 1``:

    lr = get_lr(step)                                              :1692-1700
    student fwd, teacher fwd (no grad), heads on CLS                :1741-1747
    DINO loss with the PRE-update centre, then centre EMA           :1749-1755 -> :692-720
    + gram_weight * Gram anchoring loss                             :1758-1761
    + koleo_weight * KoLeo regulariser on the student head output   :1764-1766
    backward                                                        :1772
    global grad-norm, AdamW (wd on every parameter), EMA teacher    :1781-1802

What is different from the reference, by design:
  * parameters, gradients, Adam moments and teacher weights live in flat fp32 arenas, so the grad-norm,
    AdamW and EMA are ONE kernel pass (dinox_adamw_ema) instead of 161 x (.item() + 2 EMA launches); the dW products
    accumulate straight into the gradient arena (ops._GradSink), which is zeroed once per optimiser step;
  * no host synchronisation inside a step: loss and grad-norm stay on the device until asked for;
  * data parallel: bucketed RCCL all-reduce of the gradient arena overlapped with backward, centre
    batch-mean all-reduced, 1/world folded into the AdamW kernel (dinox/dp.py).
Documented deviation: the reference's AdamW skips parameters whose ``.grad`` is None (only possible for
``scale_embed.*`` when a scale-aware model is stepped with ``spacing=None``); the arena pass applies
weight decay to them.  The reference loop always passes spacing for scale-aware models (:1713).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops
from .dp import GradBucketer, exchanging
from .schedule import get_lr


@dataclass
class StepHyperParams:
    """Defaults are the reference CLI defaults (scripts/phase5_big_run.py:1264-1285)."""
    lr: float = 1e-4
    min_lr: float = 1e-6
    warmup_steps: int = 2500
    max_steps: Optional[int] = None
    weight_decay: float = 0.04
    ema: float = 0.996
    teacher_temp: float = 0.04
    student_temp: float = 0.1
    center_momentum: float = 0.9
    gram_weight: float = 1.0
    koleo_weight: float = 0.0
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8


def flatten_parameters(module: torch.nn.Module, align: int = 8) -> Tuple[torch.Tensor, List[torch.nn.Parameter], List[int]]:
    """Move every parameter of ``module`` into one flat fp32 arena (each at an offset of a multiple of 8 elements, so that
    both the fp32 slice and the same slice of a bf16 image of the arena are 16-byte aligned) and re-point ``p.data`` at its slice.  Returns (arena, params in arena order, element offsets)."""
    params = list(module.parameters())
    if not params:
        raise ValueError("module has no parameters")
    dev = params[0].device
    offsets, total = [], 0
    for p in params:
        if p.dtype != torch.float32:
            raise TypeError("master parameters must be fp32")
        offsets.append(total)
        total += (p.numel() + align - 1) // align * align
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    for p, off in zip(params, offsets):
        flat[off:off + p.numel()].copy_(p.data.reshape(-1))
        p.data = flat[off:off + p.numel()].view(p.shape)
    return flat, params, offsets


class TrainEngine:
    """Owns student/teacher arenas, the DINO centre and the optimiser state; ``step()`` runs one update."""

    def __init__(self, student: torch.nn.Module, teacher: torch.nn.Module, out_dim: int, hp: StepHyperParams,
                 amp_dtype: Optional[torch.dtype] = None, process_group=None, bucket_bytes: int = 32 << 20,
                 accumulation_steps: int = 1, use_graph: bool = False) -> None:
        """``use_graph``: after two eager steps the whole optimiser step (forward, backward, optimiser tail) is captured ONCE into
        a hipGraph and every later step is one graph launch -- for the launch-bound small-batch regime (at bs 64 the host needs
        10-14 ms to enqueue the ~700 launches of a step that the GPU finishes in 11).  The C ABI was designed for it: no
        allocation, no synchronisation, no host-dependent scalar inside a launch (lr and the Adam bias corrections come from
        device memory, dinox_adamw_ema_dev).  Single rank, accumulation_steps == 1, fixed batch shape."""
        self.student, self.teacher, self.hp = student, teacher, hp
        if accumulation_steps < 1:
            raise ValueError("accumulation_steps must be >= 1")
        self.accum = accumulation_steps
        self.compute_dtype = amp_dtype or torch.float32
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        for p in teacher.parameters():
            p.requires_grad_(False)
        self.flat_p, self.params, self.offsets = flatten_parameters(student)
        self.flat_t, t_params, t_off = flatten_parameters(teacher)
        if t_off != self.offsets or self.flat_t.numel() != self.flat_p.numel():
            raise ValueError("student and teacher must have identical parameter layouts")
        if exchanging(process_group):                        # identical start on every rank
            dist.broadcast(self.flat_p, src=0, group=process_group)
            dist.broadcast(self.flat_t, src=0, group=process_group)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.adam_m = torch.zeros_like(self.flat_p)
        self.adam_v = torch.zeros_like(self.flat_p)
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat_g[off:off + p.numel()].view(p.shape)
        dev = self.flat_p.device
        self.center = torch.zeros(1, out_dim, dtype=torch.float32, device=dev)
        self.bucketer = GradBucketer(self.params, self.offsets, self.flat_g, bucket_bytes=bucket_bytes, group=process_group)
        self.bucketer.pre_exchange = ops.dw_stream.join
        # the teacher forward has no data dependence on the student forward: it runs on its own HIP stream so the two
        # kernel chains fill each other's tails (every launch ends with a partial last round of workgroups)
        self.side_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.shadows = [ops.ArenaShadow(self.flat_p, self.params, self.offsets), ops.ArenaShadow(self.flat_t, t_params, t_off)]
        self.use_graph = bool(use_graph)
        if self.use_graph and (self.accum != 1 or exchanging(process_group)):
            raise ValueError("use_graph: single rank and accumulation_steps == 1 only")
        self._graph = None
        self._static: Optional[list] = None
        self._eager_steps = 0
        self._hyper_dev = torch.zeros(3, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(3)
        import zoo.arch as _arch
        self.manual_top = all(type(m.head) is _arch.DinoHead and type(m.head[0]) is _arch.Linear and type(m.head[2]) is _arch.Linear
                              and m.head[0].bias is not None and m.head[2].bias is not None for m in (student, teacher)) \
            and not os.environ.get("DINOX_AUTOGRAD_TOP")
        self._zero1 = torch.zeros(1, dtype=torch.float32, device=dev)
        self.marks = None            # bench.py: a list -> (phase name, HIP event on the launch stream) at every phase boundary of step()
        self.step_count = 0          # micro-batches seen (drives the LR schedule, like the reference)
        self.opt_steps = 0           # optimiser steps taken (AdamW bias correction)
        self.last = {}

    def _mark(self, name: str) -> None:
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    # -- one optimiser step ---------------------------------------------------------------------
    def step(self, batch: torch.Tensor, spacing2b: Optional[torch.Tensor] = None, local_batch: Optional[torch.Tensor] = None,
             local_spacing: Optional[torch.Tensor] = None) -> dict:
        if not self.use_graph:
            return self._step_eager(batch, spacing2b, local_batch, local_spacing)
        return self._step_graph([batch, spacing2b, local_batch, local_spacing])

    def _step_graph(self, inputs: list) -> dict:
        hp = self.hp
        if self._graph is None and self._eager_steps < 2:       # eager first: LDS limits granted, operand images and tables built
            self._eager_steps += 1
            return self._step_eager(*inputs)
        lr = get_lr(self.step_count, hp.max_steps, hp.warmup_steps, hp.lr, hp.min_lr)
        # a FRESH page-locked staging tensor per step (the host allocator recycles it only after the copy has run): with one reused
        # buffer a host running two steps ahead would overwrite the scalars of a copy that is still queued
        self._hyper_host = torch.tensor(ops.adamw_hyper(lr, hp.beta1, hp.beta2, self.opt_steps + 1), dtype=torch.float32).pin_memory()
        self._hyper_dev.copy_(self._hyper_host, non_blocking=True)
        if self._graph is None:
            self._static = [None if t is None else t.clone() for t in inputs]
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            count, opt = self.step_count, self.opt_steps
            with torch.cuda.graph(self._graph):
                self._captured = self._step_eager(*self._static, hyper=self._hyper_dev)
            self.step_count, self.opt_steps = count, opt          # capturing enqueued nothing: the replay below IS this step
        else:
            for dst, src in zip(self._static, inputs):
                if (dst is None) != (src is None) or (dst is not None and dst.shape != src.shape):
                    raise ValueError("use_graph: the batch layout must not change after capture")
                if dst is not None:
                    dst.copy_(src, non_blocking=True)
        self._graph.replay()
        self.step_count += 1
        self.opt_steps += 1
        self.last = dict(self._captured, lr=lr)
        return self.last

    def _step_eager(self, batch: torch.Tensor, spacing2b: Optional[torch.Tensor] = None, local_batch: Optional[torch.Tensor] = None,
                    local_spacing: Optional[torch.Tensor] = None, hyper: Optional[torch.Tensor] = None) -> dict:
        """batch: (2B,3,H,W) = [view1; view2] on the device; spacing2b: (2B,3) or None.
        local_batch (L*B,3,s,s), view-major, with local_spacing (L*B,3): the multi-crop extension (not in the reference) --
        the student also sees L smaller crops per sample, which enter the DINO term only (every (teacher view, other student
        view) pair, averaged); Gram and KoLeo stay on the global views.
        Returns device tensors {loss, dino, gram, koleo, grad_norm_sq} and the python float lr (no sync)."""
        hp = self.hp
        lr = get_lr(self.step_count, hp.max_steps, hp.warmup_steps, hp.lr, hp.min_lr)
        # gradient accumulation with the reference's semantics (phase5_big_run.py:1769-1796): `step` counts micro-batches,
        # loss/accum is back-propagated every micro-batch, the optimiser (and EMA) run when (step+1) % accum == 0 with the LR
        # of that micro-batch, the centre moves every micro-batch.  Gradients are exchanged once, on the last micro-batch.
        first = self.step_count % self.accum == 0
        last = (self.step_count + 1) % self.accum == 0
        self._mark("start")
        if first:
            ops.zero_(self.flat_g)
        self.bucketer.active = last
        self.bucketer.arm()
        if ops.grad_sink.owner is not self:      # weight gradients accumulate straight into flat_g (ops._GradSink)
            ops.grad_sink.register(self, self.params, self.bucketer.grad_ready if self.bucketer.exchange else None)
            ops.weight_cache.shadows = self.shadows
        ops.grad_sink.uses.clear()
        # (the unfolded batch is shared by student and teacher WITHIN this scope, never carried across steps)
        with ops.compute_dtype(self.compute_dtype), ops.unfold_share():
            main = torch.cuda.current_stream()
            # opt-in (DINOX_SIDE_STREAM=1): +1.3 % measured, but concurrent chains blur per-kernel timings, so bench/profiles keep it off
            side = self.side_stream if os.environ.get("DINOX_SIDE_STREAM") else None
            if side is not None:
                ops.patch_unfold(batch, self.student.backbone.patch, self.compute_dtype)     # shared by both nets: before the fork
                side.wait_stream(main)
                with torch.cuda.stream(side), torch.no_grad():
                    t_feats = self.teacher.backbone(batch, spacing=spacing2b)
                s_feats = self.student.backbone(batch, spacing=spacing2b)
                main.wait_stream(side)
                t_feats.record_stream(main)
            else:
                s_feats = self.student.backbone(batch, spacing=spacing2b)
                self._mark("fwd_student")
                with torch.no_grad():
                    t_feats = self.teacher.backbone(batch, spacing=spacing2b)
                self._mark("fwd_teacher")
            if self.manual_top:
                loss, l_dino, l_gram, l_koleo, bm, bm_work = self._losses_and_backward(s_feats, t_feats, batch, local_batch, local_spacing)
            else:
                with torch.no_grad():
                    t_out = self.teacher.head(t_feats[:, 0])
                loss, l_dino, l_gram, l_koleo, bm, bm_work = self._losses_and_backward_autograd(s_feats, t_feats, t_out, batch, local_batch,
                                                                                                local_spacing)
        ops.dw_stream.join()              # (weight-gradient products enqueued on the dW stream, when DINOX_DW_STREAM is set)
        self._mark("bwd")
        if bm_work is not None:
            bm_work.wait()
            bm.div_(self.world)           # (data parallel only)
        ops.center_ema_(self.center.view(-1), bm, hp.center_momentum)
        self.bucketer.finish()
        self._mark("comm_exposed")        # what of the exchanges did not fit under backward (+ the centre EMA launch)
        if last:
            self.opt_steps += 1
            gsq = ops.adamw_ema_(self.flat_p, self.flat_g, self.adam_m, self.adam_v, self.flat_t, lr=lr,
                                 weight_decay=hp.weight_decay, beta1=hp.beta1, beta2=hp.beta2, eps=hp.adam_eps,
                                 step_t=self.opt_steps, ema=hp.ema, grad_scale=1.0 / self.world, hyper=hyper)
            ops.weight_cache.clear()     # master weights changed under the bf16 copies
            if self.compute_dtype == torch.bfloat16:
                for sh in self.shadows:  # one cast launch per arena (+ one for every transposed matrix backward uses)
                    sh.refresh()
        else:
            gsq = self._zero1                                # the reference logs grad-norm 0 between optimiser steps
        self._mark("optimiser_tail")
        self.step_count += 1
        self.last = {"loss": loss.detach(), "dino": l_dino.detach(), "gram": l_gram.detach(), "koleo": l_koleo.detach(),
                     "grad_norm_sq": gsq, "lr": lr}
        return self.last

    # -- everything above the backbones, without the framework's elementwise kernels ---------------------------------
    def _head_forward(self, head, cls_op: torch.Tensor, train: bool):
        """DinoHead = Linear(D,D) -> GELU -> Linear(D,out) (zoo/arch.py:252-256) on the CLS rows: two products, GELU (and GELU' for the
        backward) in the first one's epilogue.  Returns (logits fp32, saved)."""
        dt = self.compute_dtype
        l0, l2 = head[0], head[2]
        pre0 = torch.empty((cls_op.shape[0], l0.weight.shape[0]), dtype=dt, device=cls_op.device) if train else None
        h0 = ops.gemm(cls_op, ops.weight_operand(l0.weight, dt), bias=l0.bias, gelu=True, aux=pre0, auxgrad=True, out_dtype=dt)
        out = ops.gemm(h0, ops.weight_operand(l2.weight, dt), bias=l2.bias, out_dtype=torch.float32)
        if train:
            ops.grad_sink.use(l0.weight, l0.bias, l2.weight, l2.bias)
        return out, (cls_op, h0, pre0)

    def _head_backward(self, head, saved, ds: torch.Tensor) -> torch.Tensor:
        """d logits [V,out] fp32 -> d CLS rows [V,D]; the four parameter gradients go straight into the gradient arena."""
        dt = self.compute_dtype
        l0, l2 = head[0], head[2]
        cls_op, h0, pre0 = saved
        dy = ops.to_mode(ds, dt)
        if dt == torch.float32:
            dpre = ops.gemm(dy, l2.weight.detach(), transB=True, dgelu=True, aux=pre0, auxgrad=True, out_dtype=dt)
        else:
            dpre = ops.gemm(dy, ops.weight_operand(l2.weight, dt, transposed=True), dgelu=True, aux=pre0, auxgrad=True, out_dtype=dt)
        g2 = ops.weight_grad(dy, h0, l2.weight, l2.bias, True)
        if dt == torch.float32:
            dcls = ops.gemm(dpre, l0.weight.detach(), transB=True, out_dtype=dt)
        else:
            dcls = ops.gemm(dpre, ops.weight_operand(l0.weight, dt, transposed=True), out_dtype=dt)
        g0 = ops.weight_grad(dpre, cls_op, l0.weight, l0.bias, True)
        assert g2 == (None, None) and g0 == (None, None), "the head's parameters must live in the engine's gradient arena"
        return dcls

    def _losses_and_backward(self, s_feats, t_feats, batch, local_batch, local_spacing):
        """Heads, DINO CE (pre-update centre), Gram, KoLeo and the gradient of their weighted sum w.r.t. the student features, written
        out by hand -- every step is one of the library's kernels -- then ONE autograd backward from the features down.  (Through
        autograd the same thing costs a strided CLS copy + cast per head, `ds * g` / `d * g` multiplies, a zero fill + slice copy +
        158 MB add to merge the two feature gradients, and a handful of scalar kernels: ~0.35 ms of framework kernels per step.)"""
        hp, dt = self.hp, self.compute_dtype
        scale = 1.0 / self.accum
        V = s_feats.shape[0]
        with torch.no_grad():
            sf = s_feats.detach()
            t_out, _ = self._head_forward(self.teacher.head, ops.take_rows(t_feats, 0, dt), train=False)
            if local_batch is None:
                l_feats = None
                cls = ops.take_rows(sf, 0, dt)
            else:
                with torch.enable_grad():
                    l_feats = self.student.backbone(local_batch, spacing=local_spacing)
                lf = l_feats.detach()
                cls = ops.take_rows(sf, 0, dt, out_rows=V + lf.shape[0])
                ops.take_rows(lf, 0, dt, out=cls, out_row0=V)
            s_all, saved = self._head_forward(self.student.head, cls, train=True)
            if local_batch is None:
                l_dino, ds = ops.dino_ce(s_all, t_out, self.center, hp.student_temp, hp.teacher_temp, True, grad_scale=scale)
            else:
                l_dino, ds = ops.dino_ce_multi(s_all, t_out, self.center, hp.student_temp, hp.teacher_temp, 2, grad_scale=scale)
            # centre EMA after the loss used the old centre; batch mean is global under DP (the centre itself moves after backward,
            # so the exchange runs under the backward pass)
            bm = ops.colmean(t_out)
            bm_work = dist.all_reduce(bm, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if exchanging(self.group) else None
            # KoLeo (:1764-1766; nearest neighbours over the global batch under DP): its all-gather of the unit rows is started here and
            # awaited after the Gram loss, which has no data in common with it (nor with the head's backward)
            kstate = ops.koleo_begin(s_all[:V], group=self.group) if hp.koleo_weight > 0.0 else None
            dfeats = torch.empty_like(sf)
            l_gram = None
            if hp.gram_weight != 0.0:
                l_gram, gsaved = ops.gram_loss_fwd(sf, t_feats, dt)
                ops.gram_loss_bwd(gsaved, tuple(sf.shape), hp.gram_weight * scale, dfeats=dfeats, accumulate=False)   # rows 1..N-1
            else:
                ops.zero_(dfeats)
            l_koleo = None
            if kstate is not None:
                l_koleo, ksaved = ops.koleo_end(kstate)
                ops.axpy_(ds[:V], ops.koleo_bwd(ksaved, hp.koleo_weight * scale), 1.0)      # (ds[:V]: the leading rows, contiguous)
            dcls = self._head_backward(self.student.head, saved, ds)
            ops.put_rows_(dfeats, 0, dcls)                                                                            # row 0 (CLS)
            roots, grads = [s_feats], [dfeats]
            if l_feats is not None:
                dl = torch.empty_like(lf)
                ops.zero_(dl)
                ops.put_rows_(dl, 0, dcls, src_row0=V)
                roots.append(l_feats)
                grads.append(dl)
            loss = ops.lincomb3(l_dino, l_gram, l_koleo, hp.gram_weight, hp.koleo_weight)
        self._mark("loss")
        torch.autograd.backward(roots, grads)
        z = self._zero1
        return loss.reshape(()), l_dino.reshape(()), (l_gram if l_gram is not None else z).reshape(()), \
            (l_koleo if l_koleo is not None else z).reshape(()), bm, bm_work

    def _losses_and_backward_autograd(self, s_feats, t_feats, t_out, batch, local_batch, local_spacing):
        """The same through the per-op autograd nodes (a head whose layers were replaced, e.g. LoRA-wrapped)."""
        hp = self.hp
        if local_batch is None:
            s_out = self.student.head(s_feats[:, 0])
            l_dino = ops.DinoCEFn.apply(s_out, t_out, self.center, hp.student_temp, hp.teacher_temp)
        else:
            l_feats = self.student.backbone(local_batch, spacing=local_spacing)
            s_all = self.student.head(torch.cat([s_feats[:, 0], l_feats[:, 0]], 0))       # one head product for all views
            s_out = s_all[:s_feats.shape[0]]
            l_dino = ops.DinoCEMultiFn.apply(s_all, t_out, self.center, hp.student_temp, hp.teacher_temp, 2)
        bm = ops.colmean(t_out)
        bm_work = dist.all_reduce(bm, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if exchanging(self.group) else None
        if hp.gram_weight != 0.0:
            l_gram = ops.GramLossFn.apply(s_feats, t_feats)
            loss = l_dino + hp.gram_weight * l_gram
        else:
            l_gram = torch.zeros((), device=batch.device)
            loss = l_dino
        if hp.koleo_weight > 0.0:
            l_koleo = ops.koleo_loss(s_out, group=self.group)
            loss = loss + hp.koleo_weight * l_koleo
        else:
            l_koleo = torch.zeros((), device=batch.device)
        self._mark("loss")            # (local-crop forward, student head, DINO CE, Gram, KoLeo forward)
        (loss if self.accum == 1 else loss / self.accum).backward()
        return loss.detach(), l_dino.detach(), l_gram.detach(), l_koleo.detach(), bm, bm_work

    # -- convenience ------------------------------------------------------------------------------
    def scalars(self) -> dict:
        """Host copies of the last step's scalars (this is the only place that synchronises)."""
        r = self.last
        return {"loss": float(r["loss"]), "dino": float(r["dino"]), "gram": float(r["gram"]), "koleo": float(r["koleo"]),
                "grad_norm": float(r["grad_norm_sq"]) ** 0.5, "lr": r["lr"]}
