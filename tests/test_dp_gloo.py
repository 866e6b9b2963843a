"""Data-parallel path on CPU: world_size-2 `gloo` process groups exercising dinox/dp.py.

(1) GradBucketer: hooks + bucketed async all-reduce over a flat gradient arena reproduce the
    single-process gradient of the global batch.
(2) The DP recipe of SURVEY.md section 8e (shard by sample, sum all-reduce of gradients scaled by
    1/world, all-reduce of the teacher batch mean for the centre) with the CPU oracle as the compute:
    the 2-rank result equals the single-process oracle at the global batch.
Workers are separate OS processes (tests/_dp_workers.py); the oracle is the checker/stand-in compute."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import _dp_workers as W

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(which, tmp_path, world=2, env=None):
    port, out = _free_port(), str(tmp_path / "out.npz")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_workers.py"), which, str(r), str(world), str(port), out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **(env or {}))) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with np.load(out) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_bucketed_allreduce_matches_global_batch(tmp_path, overlap):
    """(DINOX_DP_OVERLAP=0: the buckets are exchanged after backward instead of from it -- the same gradient)"""
    got = _run("bucketer", tmp_path, env={"DINOX_DP_OVERLAP": overlap})
    from dinox.engine import flatten_parameters
    model = W.toy()
    flat_p, params, offs = flatten_parameters(model)
    X, Y = W.toy_data()
    ((model(X) - Y) ** 2).mean().backward()
    want = torch.zeros_like(flat_p)
    for p, o in zip(params, offs):
        want[o:o + p.numel()] = p.grad.reshape(-1)
    assert int(got["nbuckets"]) >= 3
    np.testing.assert_allclose(got["flat"], want.numpy(), rtol=1e-5, atol=1e-7)


def test_accumulation_times_data_parallel_exchanges_once_and_matches_global_batch(tmp_path):
    """accumulation_steps = 2 under 2 ranks: four quarter-batches (2 ranks x 2 micro-batches) of equal size, each scaled by
    1/accum, summed over micro-batches locally and over ranks ONCE (on the last micro-batch), then / world = the gradient of
    the mean loss over the whole global batch."""
    got = _run("bucketer_accum", tmp_path)
    from dinox.engine import flatten_parameters
    model = W.toy()
    flat_p, params, offs = flatten_parameters(model)
    X, Y = W.toy_data()
    ((model(X) - Y) ** 2).mean().backward()
    want = torch.zeros_like(flat_p)
    for p, o in zip(params, offs):
        want[o:o + p.numel()] = p.grad.reshape(-1)
    np.testing.assert_allclose(got["flat"], want.numpy(), rtol=1e-5, atol=1e-7)


def test_two_rank_step_equals_global_batch_oracle(tmp_path):
    got = _run("dp_oracle", tmp_path)
    O, cfg, sd, v1, v2, sp, hp = W.oracle_setup()
    st = O.init_state(cfg, sd)
    st.center = 0.05 * torch.ones(1, cfg.out_dim)
    l, _, _, grads, t_out, _ = O.losses_and_grads(st, torch.cat([v1, v2], 0), torch.cat([sp, sp], 0), hp)
    want = torch.cat([grads[k].reshape(-1) for k in grads]).numpy()
    assert float(got["loss"]) == pytest.approx(float(l), rel=1e-5)
    np.testing.assert_allclose(got["flat"], want, rtol=2e-4, atol=2e-5 * float(np.abs(want).max()))   # fp32 summation order
    np.testing.assert_allclose(got["center"], O.center_update(st.center, t_out, hp.center_momentum).numpy(), rtol=1e-5, atol=1e-7)


def test_a_dying_rank_takes_the_job_down(tmp_path):
    """A rank that crashes mid-step (tests/_dp_workers.py::rank_death: os._exit between arm() and backward) must not leave its peer
    waiting in GradBucketer.finish(): the survivor raises within the process group's time-out (dinox.dp.init_process_group,
    DINOX_DIST_TIMEOUT_S) and exits non-zero, so a launcher (torchrun) tears the job down.  Fresh child processes only."""
    import time
    port, out = _free_port(), str(tmp_path / "out.npz")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_workers.py"), "rank_death", str(r), "2", str(port), out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += b"\n[killed by the test: still running after 90 s]"
        logs.append(o.decode(errors="replace")[-1500:])
    assert procs[1].returncode == 3, logs[1]
    assert procs[0].returncode not in (0, None, -9), "the surviving rank did not fail:\n" + logs[0]
    assert time.time() - t0 < 80 and not os.path.exists(out)
    assert "gradient exchange failed" in logs[0] or "Connection" in logs[0] or "timed out" in logs[0].lower(), logs[0]


def test_shard_range_rejects_ragged():
    from dinox.dp import shard_range
    assert shard_range(8, 1, 2) == (4, 8)
    with pytest.raises(ValueError):
        shard_range(7, 0, 2)
