"""Worker side of tests/test_dp_gloo.py: run as `python _dp_workers.py <which> <rank> <world> <port> <out.npz>`.
One OS process per rank, `gloo` backend on 127.0.0.1 (CPU tensors).  TEST CODE: it may import oracle/."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "dino-x_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def toy():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(12, 33), torch.nn.Tanh(), torch.nn.Linear(33, 7), torch.nn.Tanh(), torch.nn.Linear(7, 5))


def toy_data():
    g = torch.Generator().manual_seed(0)
    return torch.randn(8, 12, generator=g), torch.randn(8, 5, generator=g)


def bucketer(rank, world, port, out):
    init(rank, world, port)
    from dinox.dp import GradBucketer, shard_range
    from dinox.engine import flatten_parameters
    model = toy()
    flat_p, params, offs = flatten_parameters(model)
    flat_g = torch.zeros_like(flat_p)
    for p, o in zip(params, offs):
        p.grad = flat_g[o:o + p.numel()].view(p.shape)
    bk = GradBucketer(params, offs, flat_g, bucket_bytes=160)        # tiny buckets -> several collectives
    assert len(bk.buckets) >= 3
    X, Y = toy_data()
    lo, hi = shard_range(8, rank, world)
    fired = []
    for _ in range(2):                                               # two steps: arm() must reset state
        flat_g.zero_()
        bk.arm()
        ((model(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
        fired.append(bk.fired_in_backward)
        bk.finish()
    # DINOX_DP_OVERLAP=0 (A/B on real nodes): nothing leaves from backward, finish() exchanges every bucket -- same sums either way
    assert fired == ([0, 0] if os.environ.get("DINOX_DP_OVERLAP") == "0" else [len(bk.buckets)] * 2), fired
    if rank == 0:
        np.savez(out, flat=(flat_g / world).numpy(), nbuckets=len(bk.buckets))
    dist.barrier()
    dist.destroy_process_group()


def bucketer_accum(rank, world, port, out):
    """Gradient accumulation x data parallel as TrainEngine.step drives it: the arena is zeroed on the first micro-batch, every
    micro-batch back-propagates loss / accum into it, the bucketer is armed each time but ACTIVE only on the last one."""
    init(rank, world, port)
    from dinox.dp import GradBucketer, shard_range
    from dinox.engine import flatten_parameters
    model = toy()
    flat_p, params, offs = flatten_parameters(model)
    flat_g = torch.zeros_like(flat_p)
    for p, o in zip(params, offs):
        p.grad = flat_g[o:o + p.numel()].view(p.shape)
    bk = GradBucketer(params, offs, flat_g, bucket_bytes=160)
    X, Y = toy_data()
    lo, hi = shard_range(8, rank, world)
    mid = (lo + hi) // 2
    accum, fired = 2, []
    for opt_step in range(2):
        for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
            if k == 0:
                flat_g.zero_()
            bk.active = k == accum - 1
            bk.arm()
            (((model(X[a:b]) - Y[a:b]) ** 2).mean() / accum).backward()
            fired.append(bk.fired_in_backward)
            bk.finish()
    assert fired == [0, len(bk.buckets)] * 2, fired           # nothing is exchanged on the first micro-batch, everything from backward on the last
    if rank == 0:
        np.savez(out, flat=(flat_g / world).numpy(), nbuckets=len(bk.buckets))
    dist.barrier()
    dist.destroy_process_group()


def oracle_setup():
    from oracle import dinox_oracle as O
    cfg = O.VitCfg(img_size=28, patch=14, dim=32, depth=1, heads=2, num_registers=2, scale_aware=True, out_dim=48)
    sd = O.random_params(cfg, seed=5)
    g = torch.Generator().manual_seed(9)
    B = 4
    v1, v2 = torch.randn(B, 3, 28, 28, generator=g), torch.randn(B, 3, 28, 28, generator=g)
    sp = torch.rand(B, 3, generator=g) + 0.5
    hp = O.HyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.9)
    return O, cfg, sd, v1, v2, sp, hp


def dp_oracle(rank, world, port, out):
    init(rank, world, port)
    from dinox.dp import all_reduce_mean_, shard_range
    O, cfg, sd, v1, v2, sp, hp = oracle_setup()
    st = O.init_state(cfg, sd)
    st.center = 0.05 * torch.ones(1, cfg.out_dim)
    lo, hi = shard_range(v1.shape[0], rank, world)
    batch = torch.cat([v1[lo:hi], v2[lo:hi]], 0)                    # both views of a sample on one rank
    sp2 = torch.cat([sp[lo:hi], sp[lo:hi]], 0)
    loss, l_dino, l_gram, grads, t_out, _ = O.losses_and_grads(st, batch, sp2, hp)
    bm = all_reduce_mean_(t_out.mean(0))                            # exchange (2): centre batch mean
    center = st.center * hp.center_momentum + bm * (1 - hp.center_momentum)
    flat = torch.cat([grads[k].reshape(-1) for k in grads])
    dist.all_reduce(flat)                                           # exchange (1); 1/world applied after
    flat /= world
    lsum = loss.clone()
    dist.all_reduce(lsum)
    if rank == 0:
        np.savez(out, flat=flat.numpy(), center=center.numpy(), loss=float(lsum) / world)
    dist.barrier()
    dist.destroy_process_group()


def rank_death(rank, world, port, out):
    """Rank 1 dies in the middle of a step (after arming, before its gradients exist); rank 0 must come out of finish() with an error
    within the process group's time-out instead of waiting for ever."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      DINOX_DIST_BACKEND="gloo", DINOX_DIST_TIMEOUT_S="8")
    torch.set_num_threads(2)
    from dinox.dp import GradBucketer, init_process_group, shard_range
    from dinox.engine import flatten_parameters
    init_process_group()
    model = toy()
    flat_p, params, offs = flatten_parameters(model)
    flat_g = torch.zeros_like(flat_p)
    for p, o in zip(params, offs):
        p.grad = flat_g[o:o + p.numel()].view(p.shape)
    bk = GradBucketer(params, offs, flat_g, bucket_bytes=160)
    X, Y = toy_data()
    lo, hi = shard_range(8, rank, world)
    bk.arm()
    ((model(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
    bk.finish()                                                      # step 1: both ranks alive
    bk.arm()
    if rank == 1:
        os._exit(3)                                                  # no clean-up, no goodbye: what a crashed rank looks like
    ((model(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
    bk.finish()                                                      # must raise
    np.savez(out, survived=1)                                        # (never reached)


if __name__ == "__main__":
    which, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    {"bucketer": bucketer, "bucketer_accum": bucketer_accum, "dp_oracle": dp_oracle, "rank_death": rank_death}[which](rank, world, port, out)
