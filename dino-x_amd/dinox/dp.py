"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference is single-process; data parallelism is this engine's addition.
The global batch is partitioned by source sample (both views of a sample stay on one rank, so the
cross-view split of DINOLoss stays local).  Per optimiser step there are exactly two exchanges:

1. sum all-reduce of the student gradient arena, in buckets that are contiguous slices of ONE flat fp32
   buffer.  Buckets are cut in *reverse* parameter order (the order backward produces gradients) and each
   is launched asynchronously from a post-accumulate-grad hook as soon as its last gradient lands, so
   the collective overlaps the rest of backward on RCCL's own stream.  xGMI is point-to-point
   (7 links x ~153 GB/s per GPU) and ring collectives are per-link bound, so buckets are few and large
   (default 32 MiB: ViT-S = 4 buckets, ViT-L = ~38) rather than NVSwitch-style small ones.
2. sum all-reduce of the 8192-float teacher batch mean before the centre EMA, so that every rank holds
   the centre of the *global* batch (scripts/phase5_big_run.py:689-690).

The 1/world scaling of (1) is folded into the fused AdamW kernel (``grad_scale``); no extra pass.
Everything here is device-agnostic: the same code runs under ``gloo`` on CPU tensors in the tests.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def exchanging(group=None) -> bool:
    """True when collectives have to run: a process group with more than one rank -- or any initialised group when
    DINOX_DP_FORCE_COLLECTIVES is set, which sends every collective of the step through the backend even at world size 1
    (how the RCCL call surface is exercised on a one-GPU box: tests/test_gpu_parity.py::test_rccl_call_surface_world1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or bool(os.environ.get("DINOX_DP_FORCE_COLLECTIVES"))


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, world, local = env_rank_world()
    if (world > 1 or os.environ.get("DINOX_DP_FORCE_COLLECTIVES")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        if backend is None:     # DINOX_DIST_BACKEND=gloo lets several ranks rehearse the DP path on ONE GPU (RCCL refuses that)
            backend = os.environ.get("DINOX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        # A rank that dies mid-step must take the job down, not leave its peers waiting in finish(): every collective carries this
        # time-out (DINOX_DIST_TIMEOUT_S, default 30 min = torch's; the tests use a few seconds), after which the waiting ranks raise.
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("DINOX_DIST_TIMEOUT_S") or 1800))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


def shard_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, equal sample shard of a global batch (global batch must divide evenly)."""
    if n_samples % world:
        raise ValueError(f"global batch {n_samples} is not divisible by world size {world}")
    per = n_samples // world
    return rank * per, (rank + 1) * per


@dataclass
class _Bucket:
    lo: int            # element range [lo, hi) of the flat gradient arena
    hi: int
    pending: int       # parameters whose gradient has not landed yet (this step)
    n_params: int
    work: Optional[object] = None


class GradBucketer:
    """Bucketed asynchronous all-reduce over a flat gradient arena.

    ``params`` are the parameters in arena order; ``offsets[i]`` is the element offset of params[i] in
    ``flat_grad``.  Call ``arm()`` before backward, ``finish()`` after it (waits for every bucket)."""

    def __init__(self, params: Sequence[torch.nn.Parameter], offsets: Sequence[int], flat_grad: torch.Tensor,
                 bucket_bytes: int = 32 << 20, group=None, tail_bytes: int = 4 << 20) -> None:
        """``tail_bytes``: the bucket that fills LAST (the first parameters of the arena: tokens, patch embedding, the first
        blocks) cannot hide under backward -- nothing is left to run -- so it is kept small: its all-reduce is the only exposed one."""
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchange = exchanging(group)
        self.buckets: List[_Bucket] = []
        self.bucket_of: Dict[int, int] = {}
        cap = max(1, bucket_bytes // flat_grad.element_size())
        tail = max(1, min(tail_bytes, bucket_bytes) // flat_grad.element_size())
        # walk parameters from last to first; a bucket is a contiguous slice [lo, hi)
        hi = None
        lo = None
        members: List[int] = []
        for i in range(len(params) - 1, -1, -1):
            p_lo, p_hi = offsets[i], offsets[i] + params[i].numel()
            if hi is None:
                hi, lo, members = p_hi, p_lo, [i]
            else:
                lo = p_lo
                members.append(i)
            # close the bucket when it is full -- or when what is left below it fits the small tail bucket
            if hi - lo >= cap or i == 0 or (p_lo <= tail and hi - lo > 0 and p_lo > 0 and hi > tail):
                b = len(self.buckets)
                self.buckets.append(_Bucket(lo=lo, hi=hi, pending=len(members), n_params=len(members)))
                for j in members:
                    self.bucket_of[j] = b
                hi = None
        self._seen: set = set()
        self.active = True          # False on all but the last micro-batch of a gradient-accumulation group
        self.fired_in_backward = 0  # buckets of the last step launched from grad_ready (i.e. overlapped), not by finish()
        self.pre_exchange = lambda: None   # run on the launch stream before a bucket is exchanged (engine: joins the dW stream)
        self._hooks = []
        if self.exchange:
            for i, p in enumerate(params):
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i: int):
        def hook(_param):
            self.grad_ready(i)
        return hook

    def arm(self) -> None:
        self._seen = set()
        self.fired_in_backward = 0
        for b in self.buckets:
            b.pending = b.n_params
            b.work = None

    def grad_ready(self, i: int) -> None:
        if i in self._seen:         # autograd runs the post-accumulate hook even for a parameter whose gradient went straight into
            return                  # the arena (ops._GradSink announced it already): count every parameter once per step
        self._seen.add(i)
        b = self.buckets[self.bucket_of[i]]
        b.pending -= 1
        if b.pending == 0 and self.exchange and self.active and not NO_OVERLAP:
            self.fired_in_backward += 1
            self.pre_exchange()
            b.work = dist.all_reduce(self.flat[b.lo:b.hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        """Launch any bucket that never filled (parameters without gradient) and wait for all."""
        if not self.exchange or not self.active:
            return
        self.pre_exchange()
        for b in self.buckets:
            if b.work is None:
                b.work = dist.all_reduce(self.flat[b.lo:b.hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for i, b in enumerate(self.buckets):
            try:
                b.work.wait()
            except RuntimeError as e:          # a peer went away (or the time-out of init_process_group ran out): fail loudly, on every surviving rank
                raise RuntimeError(f"gradient exchange failed in bucket {i} of {len(self.buckets)} (elements {b.lo}..{b.hi}): a peer rank is gone or "
                                   f"stalled -- {e}") from e
            b.work = None

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


# DINOX_DP_OVERLAP=0: every bucket is exchanged from finish(), AFTER backward (one burst on all links, nothing beside the backward kernels).
# The default overlaps the exchange with backward; on MI355X the persistent GEMM / attention kernels then run beside RCCL's channel
# workgroups and pay +35..55 % while they do (tools/cotenant_probe.py) -- which of the two is cheaper on a given node is
# a measurement, and this switch is how to take it: bench.py --gpus N times a few warm-up steps both ways and keeps the faster (it sets
# NO_OVERLAP on every rank alike; DINOX_DP_OVERLAP=0|1 pins it).
NO_OVERLAP = os.environ.get("DINOX_DP_OVERLAP", "1") == "0"


def all_reduce_mean_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks (no-op for world size 1)."""
    if exchanging(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t
