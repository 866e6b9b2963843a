"""pytest configuration: markers, import paths, fixture loaders."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dino-x_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The C-ABI library is a build artefact (git-ignored).  A fresh checkout gets it from __graft_entry__.build(); if the tests
# are started before that, build it here rather than fail at import (hipcc cross-compiles gfx950 without a GPU).
if not os.path.exists(os.path.join(PKG, "dinox", "libdinox_hip.so")):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j", "8"], check=True, stdout=subprocess.DEVNULL)

from dinox.hostinfo import usable_cpus  # noqa: E402

torch.set_num_threads(min(8, usable_cpus()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(d: dict, prefix: str) -> dict:
    """{'w/a.b': x} -> {'a.b': tensor(x)} for keys under prefix."""
    pre = prefix + "/"
    return {k[len(pre):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(pre)}


def t(a) -> torch.Tensor:
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def cli():
    """The drop-in training script (dino-x_amd/scripts/phase5_big_run.py) as a module."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("phase5_big_run", os.path.join(ROOT, "dino-x_amd", "scripts", "phase5_big_run.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod            # dataclasses resolve string annotations through sys.modules
    spec.loader.exec_module(mod)
    return mod
