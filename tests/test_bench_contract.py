"""The bench.py output contract, checked on the line recorded in profiles/ (the last `python bench.py` run on an MI355X):
one JSON object with the driver's keys, the roofline and cpu_baseline objects, and internally consistent numbers."""
import json
import os
import re

import pytest

from conftest import ROOT


def _recorded_line():
    import glob
    md = open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_bs256_summary.md")))[-1]).read()      # the latest round's record
    sec = md[md.index("Default bench line"):]
    m = re.search(r"```\n(\{.*?\})\n```", sec, re.S)
    assert m, "no default bench line recorded"
    return json.loads(m.group(1))


def test_recorded_bench_line_meets_the_contract():
    d = _recorded_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = samples per step / time per step
    assert d["value"] == pytest.approx(d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3), rel=1e-3)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3) and 0 < r["frac"] < 1
    assert (r["bound"] == "hbm") == (r["unit"] == "GB/s")
    # the two floors are both reported and the larger one names the bound
    assert (r["hbm"]["floor_ms_per_step"] >= r["mfma"]["floor_ms_per_step"]) == (r["bound"] == "hbm")
    # achieved = algorithmic bytes per launch / average launch time
    if r["bound"] == "hbm":
        assert r["achieved"] == pytest.approx(r["hbm"]["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9, rel=2e-3)
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["hbm"]["algorithmic_bytes_per_launch"]
    if "step_ms_split" in d:                       # round 2 on: phase split (HIP events) and the other single-GPU configs in the same line
        sp = d["step_ms_split"]
        assert set(sp) == {"fwd_student", "fwd_teacher", "loss", "bwd", "comm_exposed", "optimiser_tail"} and all(v >= 0 for v in sp.values())
        assert sum(sp.values()) == pytest.approx(d["ms_per_step"], rel=0.15)
        sec = d["secondary"]
        assert {"bs64_scale_off", "multicrop_2g8l", "vit_large_bs128"} <= set(sec)
        for k, v in sec.items():
            assert "error" not in v, (k, v)
            assert v["unit"] == "samples/s" and v["value"] > 0 and v["ms_per_step"] > 0 and v["workload"]
    if "framework_baseline" in d:                   # the same oracle step through plain PyTorch-ROCm on the same GPU (round 2 on)
        f = d["framework_baseline"]
        assert "error" not in f, f
        assert f["unit"] == d["unit"] and f["kind"] == "port" and f["value"] > 0 and f["sample"]
        assert f["speedup"] == pytest.approx(d["value"] / f["value"], rel=1e-2)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
