#!/usr/bin/env python3
"""bench.py -- training-step throughput of the DINO-X hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL)

Started plainly with ``--gpus N`` (N > 1) it launches that second form itself: N fresh child processes, spawned BEFORE this
process has made any GPU call, whose rank-0 line it relays (it never re-executes a process that touched the GPU).

A "step" is one full optimiser step of the reference loop (scripts/phase5_big_run.py:1692-1802):
student forward, teacher forward (no grad), DINO centring/sharpening CE + Gram-anchoring loss,
backward, global grad-norm, AdamW, EMA teacher, centre update -- nothing skipped.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ViT-Small/16,
224x224x3 synthetic 2.5D slice batches, --scale-aware, 256 source samples per GPU = 512 views through
each network (2 global views per sample: the reference has no multi-crop),
out_dim 8192, bf16 MFMA operands with fp32 accumulation / residual stream / losses / optimiser.
Inputs are resident in HBM before the timed region.  Weak scaling: per-GPU batch fixed.

Output: ONE JSON line on rank 0 (see the contract in the task description), with
  roofline       the dominant kernel (the bf16 MFMA GEMM family that takes most of the step): algorithmic bytes / FLOPs of its
                 launches / their summed duration, measured live with HIP events on the launch stream over the timed
                 region; "step" adds the whole-step figure of BASELINE.md section 2;
  step_ms_split  HIP-event time between the phase boundaries of the step (fwd_student, fwd_teacher, loss, bwd,
                 comm_exposed, optimiser_tail), averaged over a few extra steps after the timed region
                 (protocol of the reference's scripts/tune_throughput.py:640-668: one sync per step);
  secondary      (N = 1 only) short measurements: bs256_side_stream / bs256_dw_stream (the headline workload with the opt-in second
                 stream for the teacher's forward / for the weight-gradient products) and the other single-GPU BASELINE configs: bs64_scale_off (configs[1]),
                 multicrop_2g8l (the literal "2 global + 8 local crops" reading of configs[2]; an extension, the reference
                 has two views), vit_large_bs128 (the per-GPU shape of configs[4]);
  cpu_baseline   the CPU oracle's same training step timed on this box's host cores (rank 0, N=1 only);
  framework_baseline  the same oracle step with every tensor on this GPU under torch.autocast(bfloat16): what the reference's
                 own loop does on this MI355X through plain PyTorch-ROCm, at the headline batch size (+ "speedup" = value / it).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dino-x_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PMC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json")  # profiles/: per-kernel HBM bytes from the rocprofv3 --pmc passes of this same command, newest first


def fwd_flops_per_image(img=224, patch=16, dim=384, depth=12, out_dim=8192, regs=4, gram=True):
    """BASELINE.md section 2 / SURVEY.md 8d: F_fwd per image (gram=False: a view that does not enter the Gram term)."""
    P = (img // patch) ** 2
    N = 1 + P + regs
    blk = N * dim * 3 * dim + 2 * N * N * dim + N * dim * dim + 8 * N * dim * dim
    return 2.0 * (P * 3 * patch * patch * dim + depth * blk + dim * dim + dim * out_dim + ((N - 1) ** 2 * dim if gram else 0))


def cpu_baseline(cfg_kw, out_dim, seconds_budget=25.0):
    """The oracle's training step (oracle/dinox_oracle.py: fp32 torch-CPU restatement of the reference
    loop) on a bounded sample of the same workload: same model, B=8 source samples per step."""
    import torch
    from oracle import dinox_oracle as O
    from dinox.hostinfo import usable_cpus
    cores = usable_cpus()                                  # cpuset + cgroup quota, not os.cpu_count()
    torch.set_num_threads(cores)
    B = 8
    cfg = O.VitCfg(out_dim=out_dim, **cfg_kw)
    sd = O.random_params(cfg, seed=0)
    st = O.init_state(cfg, sd)
    g = torch.Generator().manual_seed(1234)
    batch = torch.randn(2 * B, 3, cfg.img_size, cfg.img_size, generator=g)
    sp = torch.rand(B, 3, generator=g) * 0.5 + 0.5
    sp2 = torch.cat([sp, sp], 0)
    hp = O.HyperParams()
    O.train_step(st, batch, sp2, hp)                      # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        O.train_step(st, batch, sp2, hp)
        n += 1
        dt = time.perf_counter() - t0
        if n >= 3 and (dt > seconds_budget or n >= 8):
            break
    return {"value": round(B * n / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle train_step (fp32 torch CPU), ViT-S/16 224 scale-aware, B={B} samples/step, {n} timed steps after 1 warm-up"}


def framework_baseline(cfg_kw, out_dim, dev, B=256, steps=5, amp=True):
    """The SAME oracle step (the reference's loop restated in plain torch: its modules' functionals, its per-parameter grad-norm /
    AdamW / EMA loops) with every tensor on this GPU and the forward under torch.autocast(bfloat16) -- i.e. what the reference
    itself would do on this MI355X through PyTorch-ROCm (hipBLASLt GEMMs, the framework's SDPA / LayerNorm / optimizer kernels),
    at the headline batch size.  A baseline beside cpu_baseline, measured in the same leg; never a code path of the product."""
    import torch
    from oracle import dinox_oracle as O
    cfg = O.VitCfg(out_dim=out_dim, **cfg_kw)
    st = O.init_state(cfg, O.random_params(cfg, seed=0))
    for name in ("student", "teacher", "adam_m", "adam_v"):
        setattr(st, name, {k: v.to(dev) for k, v in getattr(st, name).items()})
    st.center = st.center.to(dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    batch = torch.randn(2 * B, 3, cfg.img_size, cfg.img_size, device=dev, generator=g)
    sp = torch.rand(B, 3, device=dev, generator=g) * 0.5 + 0.5
    sp2 = torch.cat([sp, sp], 0)
    hp = O.HyperParams()
    with O.autocast_device("cuda"):
        for _ in range(2):
            O.train_step(st, batch, sp2, hp, amp=amp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            O.train_step(st, batch, sp2, hp, amp=amp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / steps, 2), "kind": "port", "device": "this GPU",
            "sample": f"oracle train_step on cuda {'under torch.autocast(bfloat16)' if amp else 'in fp32 (no autocast)'} (torch {torch.__version__}: plain PyTorch-ROCm kernels), ViT-S/16 224 "
                      f"scale-aware, B={B} samples/step, {steps} timed steps after 2 warm-up"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # SURVEY 8d protocol: >= 5 warm-up, >= 20 timed
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=256, help="source samples per GPU")
    ap.add_argument("--fp32", action="store_true", help="parity mode (exact-fp32 MFMA) instead of bf16")
    ap.add_argument("--model", choices=["vit-small", "vit-large"], default="vit-small",
                    help="vit-small = BASELINE configs[2] (the metric's config); vit-large = configs[4] shape (not the headline metric)")
    ap.add_argument("--no-scale-aware", action="store_true", help="configs[1]: scale embedding off")
    ap.add_argument("--local-crops", type=int, default=0,
                    help="multi-crop extension (not in the reference, whose loop has 2 global views): L extra student-only local views per sample")
    ap.add_argument("--local-size", type=int, default=96)
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (launch-bound small-batch regime)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short measurements of the other single-GPU configs")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--time-every", type=int, default=16, help="bracket one GEMM launch in this many with HIP events (1: all, which costs the timed region ~2.3 %%)")
    ap.add_argument("--no-step-split", action="store_true", help="skip the extra steps that time the phases of the step (profiler runs)")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` outside torchrun: start N ranks as CHILD processes of a parent that has not touched the GPU
    (``torch.cuda.device_count()`` does not initialise it on this image), relay their output, return their exit code."""
    import torch
    have = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if have < args.gpus and env.get("DINOX_DIST_BACKEND") != "gloo":
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible (set DINOX_DIST_BACKEND=gloo to rehearse several ranks per GPU)",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class Workload:
    """One configuration resident on the device: models, engine, synthetic batch."""

    def __init__(self, dev, rank, *, model="vit-small", B=256, scale_aware=True, L=0, local_size=96, fp32=False, steps_hint=30, graph=False, koleo=0.0,
                 accum=1):
        import torch
        import zoo.arch as arch
        from dinox.engine import StepHyperParams, TrainEngine
        self.cfg_kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=scale_aware)
        if model == "vit-large":
            self.cfg_kw.update(dim=1024, depth=24, heads=16)
        self.out_dim, self.B, self.L, self.local_size, self.model = 8192, B, L, local_size, model
        torch.manual_seed(0)
        student = arch.DinoStudentTeacher(arch.PatchViT(**self.cfg_kw), self.out_dim)
        if scale_aware:      # exercise the scale branch: the reference zero-inits it, training moves it away from zero
            torch.nn.init.xavier_uniform_(student.backbone.scale_embed.mlp[2].weight)
        teacher = arch.DinoStudentTeacher(arch.PatchViT(**self.cfg_kw), self.out_dim)
        teacher.load_state_dict(student.state_dict())
        self.eng = TrainEngine(student.to(dev), teacher.to(dev), self.out_dim, StepHyperParams(max_steps=steps_hint + 10, warmup_steps=5, koleo_weight=koleo),
                               amp_dtype=None if fp32 else torch.bfloat16, accumulation_steps=accum, **({"use_graph": True} if graph else {}))
        g = torch.Generator().manual_seed(1234 + rank)       # per-rank shard of the synthetic global batch
        self.batch = torch.randn(2 * B, 3, 224, 224, generator=g).to(dev)
        sp = (torch.rand(B, 3, generator=g) * torch.tensor([0.52, 0.52, 4.375]) + torch.tensor([0.46, 0.46, 0.625]))
        self.sp2 = torch.cat([sp, sp], 0).to(dev) if scale_aware else None
        self.loc = torch.randn(L * B, 3, local_size, local_size, generator=g).to(dev) if L else None
        self.spl = torch.cat([sp] * L, 0).to(dev) if (L and scale_aware) else None

    def step(self):
        return self.eng.step(self.batch, self.sp2, self.loc, self.spl)

    def gflop_per_sample(self) -> float:
        gf = 8.0 * fwd_flops_per_image(dim=self.cfg_kw["dim"], depth=self.cfg_kw["depth"]) / 1e9
        if self.L:       # student fwd + bwd (3x fwd) of every local view; the Gram term does not see them
            gf += 3.0 * self.L * fwd_flops_per_image(img=self.local_size, dim=self.cfg_kw["dim"], depth=self.cfg_kw["depth"], gram=False) / 1e9
        return gf

    def step_split(self, n=4) -> dict:
        """HIP-event milliseconds between the phase boundaries TrainEngine.step marks on its launch stream."""
        import torch
        acc: dict = {}
        for _ in range(n):
            self.eng.marks = []
            self.step()
            torch.cuda.synchronize()
            m = self.eng.marks
            for (_, e0), (name, e1) in zip(m[:-1], m[1:]):
                acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
        self.eng.marks = None
        return {k: round(v / n, 3) for k, v in acc.items()}


def timed(wl, steps, warmup, barrier, note=None, timer=None):
    import torch
    from dinox import ops
    for i in range(warmup):
        wl.step()
        torch.cuda.synchronize()
        if note:
            note(f"warm-up step {i} done")
    if timer is not None:
        timer.start()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    t_enq = time.perf_counter() - t0          # host side done enqueueing; the device may still be running
    barrier()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.stop()
    if note:
        note(f"host enqueue {1e3 * t_enq / max(steps, 1):.2f} ms/step of {1e3 * dt / max(steps, 1):.2f} ms/step (with the launch queue full)")
    timed.host_enqueue_in_flight_ms = 1e3 * t_enq / max(steps, 1)
    return dt


def host_enqueue_ms(wl, n=5) -> float:
    """What the host needs to ISSUE one step: each of n steps starts on an empty launch queue (device synchronised first) and is timed
    to the return of step() -- inside the timed region the same call mostly waits for room in the queue of a GPU-bound step (round 2
    reported that figure, 24 ms; the host's own share was 8.8 ms, tools/host_overhead.py)."""
    import torch
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.step()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return 1e3 * sorted(ts)[len(ts) // 2]


def secondary(dev, note) -> dict:
    """Short (3 warm-up + 2 x 8 timed steps) measurements of the other single-GPU BASELINE configs, so that the driver's record -- not a
    builder log -- holds them.  `value` / `ms_per_step` are the MEAN of the two rounds (the protocol the headline and the baselines can be
    compared with); the better round rides along as `best_ms_per_step`.  Each frees its memory before the next."""
    import gc
    import torch
    out = {}
    from dinox import ops
    runs = [("bs256_side_stream", dict(B=256, side_stream=True),
             "the headline workload with DINOX_SIDE_STREAM=1 (the teacher's forward on a second HIP stream beside the student's: the step's idle phases fill in, +2.7 % on ViT-S -- and -5 % on ViT-L, whose kernels leave none; overlapping kernels cannot be priced one by one, so the headline line keeps it off)"),
            ("bs256_dw_stream", dict(B=256, dw_stream=True),
             "the headline workload with DINOX_DW_STREAM=1 (weight-gradient products on a second HIP stream: faster, but overlapping kernels cannot be priced one by one, so the headline line keeps it off)"),
            ("bs256_koleo_accum4", dict(B=256, koleo=0.1, accum=4),
             "the headline workload the way the reference's production runs use it (docs/EXPERIMENTS.md): --koleo-weight 0.1, --accumulation-steps 4 (samples/s counts micro-batches)"),
            ("bs64_fp32", dict(B=64, fp32=True),
             "the headline model without --amp: fp32 on the exact-fp32 matrix instructions, bs 64 (python bench.py --fp32 --batch-size 64 adds the plain-PyTorch fp32 figure)"),
            ("bs64_scale_off", dict(B=64, scale_aware=False), "BASELINE configs[1]: ViT-S/16 224, bs 64, scale-aware off, 2 views/sample"),
            ("bs64_scale_off_graph", dict(B=64, scale_aware=False, graph=True), "configs[1] with the step replayed as one captured hipGraph"),
            ("multicrop_2g8l", dict(B=256, L=8), "configs[2] read literally: 2 global + 8 local 96px views/sample (extension: the reference has 2 views)"),
            ("vit_large_bs128", dict(model="vit-large", B=128), "per-GPU shape of configs[4]: ViT-L/16 224, bs 128, scale-aware, Gram on")]

    def sync():
        torch.cuda.synchronize()

    for name, kw, what in runs:
        was = ops.dw_stream.enabled
        side_was = os.environ.get("DINOX_SIDE_STREAM")
        try:
            ops.dw_stream.enabled = was or bool(kw.pop("dw_stream", False))
            if kw.pop("side_stream", False):
                os.environ["DINOX_SIDE_STREAM"] = "1"
            wl = Workload(dev, 0, **kw)
            rounds = [timed(wl, 8, 3, sync), timed(wl, 8, 0, sync)]    # two rounds of eight (a single stall -- an allocation of a new size, a
            dt = sum(rounds) / 2                                        # clock ramp -- moves one 8-step figure by up to 30 %: both are reported)
            scal = wl.eng.scalars()
            sps = wl.B * 8 / dt
            out[name] = {"workload": what, "value": round(sps, 1), "unit": "samples/s", "ms_per_step": round(1e3 * dt / 8, 3),
                         "best_ms_per_step": round(1e3 * min(rounds) / 8, 3), "protocol": "mean of 2 rounds x 8 steps after 3 warm-up steps", "steps": 16, "warmup": 3,
                         "views_per_s": round((2 + wl.L) * sps, 1), "step_tflops": round(sps * wl.gflop_per_sample() / 1e3, 1),
                         "step_frac_of_mfma_peak": round(sps * wl.gflop_per_sample() / 1e3 / PEAK_BF16_DENSE_TFLOPS, 4), "loss": round(scal["loss"], 4)}
            note(f"secondary {name}: {out[name]['value']} samples/s, {out[name]['ms_per_step']} ms/step")
        except Exception as e:                                # a secondary line must never cost the headline line
            out[name] = {"workload": what, "error": f"{type(e).__name__}: {e}"[:300]}
            note(f"secondary {name} failed: {out[name]['error']}")
        ops.dw_stream.enabled = was
        if side_was is None:
            os.environ.pop("DINOX_SIDE_STREAM", None)
        else:
            os.environ["DINOX_SIDE_STREAM"] = side_was
        wl = None
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    from dinox import ops
    from dinox.dp import init_process_group
    from dinox.hostinfo import usable_cpus

    torch.set_num_threads(max(1, usable_cpus() // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    rank, world, local = init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    local = local % torch.cuda.device_count()        # (rehearsals may put several gloo ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    T_START = time.perf_counter()

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    B, L = args.batch_size, args.local_crops
    wl = Workload(dev, rank, model=args.model, B=B, scale_aware=not args.no_scale_aware, L=L, local_size=args.local_size, fp32=args.fp32,
                  steps_hint=args.steps + args.warmup + 8 + (16 if world > 1 else 0), graph=args.graph)
    note(f"model + data resident (world {world}, B {B}/GPU, {'fp32' if args.fp32 else 'bf16'})")
    timer = None if (args.no_kernel_timing or args.graph) else ops.GemmTimer(every=args.time_every)
    # Data parallel: should the gradient buckets be exchanged FROM backward (overlapped: RCCL's channel workgroups then hold CUs beside the
    # persistent GEMM / attention kernels, which cost them +35..55 % while they do) or AFTER it (one exposed burst)?  Which is cheaper is a
    # property of the node, so a few extra warm-up steps time both and every rank keeps the faster -- unless
    # DINOX_DP_OVERLAP pins it.  Nothing of this runs inside the timed region.
    dp_probe = None
    if world > 1 and "DINOX_DP_OVERLAP" not in os.environ:
        from dinox import dp as _dp
        probe = {}
        for mode, flag in (("overlapped", False), ("after_backward", True)):
            _dp.NO_OVERLAP = flag
            for _ in range(2):
                wl.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(4):
                wl.step()
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # (the same number on every rank: the same choice)
            probe[mode] = float(t) / 4 * 1e3
        _dp.NO_OVERLAP = probe["after_backward"] < probe["overlapped"]
        dp_probe = {"ms_per_step_overlapped": round(probe["overlapped"], 3), "ms_per_step_after_backward": round(probe["after_backward"], 3),
                    "chosen": "after_backward" if _dp.NO_OVERLAP else "overlapped"}
        note(f"gradient exchange: {dp_probe}")
    dt = timed(wl, args.steps, args.warmup, barrier, note, timer)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    note(f"timed region: {args.steps} steps in {dt:.3f}s")
    scal = wl.eng.scalars()
    if not (scal["loss"] == scal["loss"]):
        raise SystemExit("non-finite loss in the timed region")
    kernels = timer.summary() if timer else {}
    host_ms = host_enqueue_ms(wl)                     # (every rank: the step holds collectives)
    split = wl.step_split() if not (args.graph or args.no_step_split) else None     # (every rank: the step holds collectives)
    overlapped = getattr(wl.eng.bucketer, "fired_in_backward", None)
    # what the collectives really ran on: the backend of the process group, the number of ranks an all-reduce of ones sees, and the
    # slowest rank's exposed communication (the `comm_exposed` phase of step_ms_split: what did not fit under backward)
    dist_info = None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        exposed = torch.tensor([split["comm_exposed"] if split else -1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(exposed, op=dist.ReduceOp.MAX)
        bk = wl.eng.bucketer
        dist_info = {"dist_backend": dist.get_backend(), "rccl_ranks_seen": int(ones.item()), "comm_exposed_ms_max_over_ranks": round(float(exposed), 3),
                     "grad_bucket_bytes": [4 * (b_.hi - b_.lo) for b_ in bk.buckets], "grad_bytes_per_step": 4 * int(wl.eng.flat_g.numel()),
                     "centre_allreduce_bytes": 4 * wl.out_dim}

    if rank == 0:
        samples_s = world * B * args.steps / dt
        gf_sample = wl.gflop_per_sample()
        step_tflops = samples_s * gf_sample / 1e3 / world          # per GPU
        roof = {"bound": "mfma", "achieved": None, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None}
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms"])
            d = kernels[dom]
            sec = d["ms"] * 1e-3
            tf, gbs = d["flops"] / sec / 1e12, d["bytes"] / sec / 1e9
            # Which roof bounds the dominant kernel is decided from its own launches: the time the MFMA pipe needs for their
            # algorithmic FLOPs at the dense bf16 peak against the time HBM needs for their algorithmic bytes at its peak.
            # With K = 384..1536 and M ~ 1e5 the products of this path sit at or under the ridge (2.5 PF / 8 TB/s = 312 FLOP/B;
            # qkv 288, fc1 with its GELU' side tensor 171, proj 77): the larger floor is the HBM one.
            t_mfma, t_hbm = d["flops"] / (PEAK_BF16_DENSE_TFLOPS * 1e12), d["bytes"] / (PEAK_HBM_GBS * 1e9)
            mfma = {"achieved": round(tf, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_DENSE_TFLOPS, 4),
                    "floor_ms_per_step": round(1e3 * t_mfma / args.steps, 3)}
            hbm = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                   "floor_ms_per_step": round(1e3 * t_hbm / args.steps, 3), "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"])}
            if t_hbm >= t_mfma:
                roof.update(bound="hbm", achieved=hbm["achieved"], peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm["frac"], mfma=mfma, hbm=hbm)
            else:
                roof.update(bound="mfma", achieved=mfma["achieved"], frac=mfma["frac"], mfma=mfma, hbm=hbm)
            roof.update(kernel=dom, launches_per_step=d["launches"] // args.steps, timed_launches=d["timed"], avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2),
                        kernel_share_of_step=round(d["ms"] / (dt * 1e3), 4),
                        all_gemm_kernels={k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                              "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in kernels.items()})
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and, in a
        # separate run, WRITE_SIZE of this same command; FETCH doubled per the gfx950 correction) -- bench.py cannot run the
        # profiler on itself, so the figure is read from profiles/.  It is reported only while the kernel sources are the ones the
        # profile was taken on (its "_source_fingerprint" stamp, dinox.hostinfo.source_fingerprint): a stale profile gives null.
        if args.model == "vit-small" and B == 256 and not L and not args.no_scale_aware:
            from dinox.hostinfo import source_fingerprint
            fp_now = source_fingerprint()
            for pmc_file in PMC_FILES:
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                except (OSError, ValueError):
                    continue
                fam = "dinox::" + roof.get("kernel", "")
                if pmc.get("_source_fingerprint") != fp_now:
                    roof["traffic_note"] = (f"profiles/{pmc_file} was taken on other kernel sources (stamp {pmc.get('_source_fingerprint')}, now {fp_now}): "
                                            "traffic withheld until tools/refresh_profiles.sh is re-run")
                elif fam in pmc:
                    roof["traffic"] = pmc[fam]["hbm_bytes_per_launch"]
                    roof["traffic_note"] = f"measured avg HBM bytes/launch (PMC, profiles/{pmc_file}, same kernel sources) beside hbm.algorithmic_bytes_per_launch"
                break
        roof["step"] = {"gflop_per_sample": round(gf_sample, 2), "achieved": round(step_tflops, 2),
                        "frac": round(step_tflops / PEAK_BF16_DENSE_TFLOPS, 4)}
        views = "2 views/sample" if not L else f"2 global + {L} local {args.local_size}px views/sample"
        if args.model == "vit-small":
            metric = (f"training images/sec (source samples; 2 global + {L} local {args.local_size}px views each; multi-crop extension, the reference has 2 global views) ViT-S/16 224px bs{B}/GPU"
                      if L else f"training images/sec (source samples; 2 global views each) ViT-S/16 224px bs{B}/GPU")
        else:
            metric = "training images/sec (source samples; 2 global views each) ViT-L/16 224px (configs[4] shape, not the headline metric)"
        line = {
            "metric": metric, "value": round(samples_s, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": ("ViT-S" if args.model == "vit-small" else "ViT-L") + f"/16 224x224x3 2.5D slice stacks, "
                                   f"{'scale-aware' if not args.no_scale_aware else 'scale-aware off'}, {views}, DINO+Gram loss, AdamW+EMA",
                       "per_gpu_batch": B, "global_batch": B * world, "views_per_step": (2 + L) * B * world, "local_crops": L, "tokens": 201, "out_dim": wl.out_dim,
                       "parallelism": f"dp{world}", "views_per_s": round((2 + L) * samples_s, 2), "hipgraph": bool(args.graph), "host_enqueue_ms_per_step": round(host_ms, 2),
                       "host_enqueue_note": "median host time to issue one step on an empty launch queue; host_enqueue_ms_in_flight = the same call inside the timed region, where it also waits for queue room",
                       "host_enqueue_ms_in_flight": round(timed.host_enqueue_in_flight_ms, 2),
                       "loss": round(scal["loss"], 5), "grad_norm": round(scal["grad_norm"], 5)},
            "roofline": roof,
            "step_ms_split": split,
        }
        if world > 1 and overlapped is not None:
            line["config"]["grad_buckets_launched_during_backward"] = f"{overlapped}/{len(wl.eng.bucketer.buckets)}"
        if dist_info is not None:
            line["config"].update(dist_info)
        if dp_probe is not None:
            line["config"]["grad_exchange_probe"] = dp_probe
        default_cfg = args.model == "vit-small" and B == 256 and not L and not args.no_scale_aware and not args.fp32
        if world == 1 and not args.no_secondary and default_cfg:
            wl = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            line["secondary"] = secondary(dev, note)
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle (cpu_baseline) ...")
            vit_s = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
            line["cpu_baseline"] = cpu_baseline(vit_s, 8192) if args.model == "vit-small" else None
            if args.model == "vit-small" and (default_cfg or (args.fp32 and not L and not args.no_scale_aware)):     # (--fp32: the no-"--amp" comparison, any batch size)
                try:
                    wl = None
                    torch.cuda.empty_cache()
                    note("timing the oracle step on this GPU through plain PyTorch-ROCm (framework_baseline) ...")
                    line["framework_baseline"] = framework_baseline(vit_s, 8192, dev, B=B, steps=5 if not args.fp32 else 3, amp=not args.fp32)
                    line["framework_baseline"]["speedup"] = round(line["value"] / line["framework_baseline"]["value"], 2)
                except Exception as e:                            # a baseline must never cost the headline line
                    line["framework_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
