"""Host CPU budget: honour cpuset affinity AND the cgroup CPU quota (a GPU box can expose 256 logical
CPUs with a 16-CPU quota; sizing thread pools from os.cpu_count() there throttles everything)."""
from __future__ import annotations

import math
import os


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:              # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.floor(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def source_fingerprint() -> str:
    """16 hex digits over the kernel sources (csrc/*.hip, *.h, the C-ABI header) this library is built from.  The GPU box's snapshot
    carries no .git, so profiles/*.json are stamped with this instead of a commit hash: bench.py reports a profile's PMC figures only
    while the kernels are the ones the profile was taken on."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files.append(os.path.join(os.path.dirname(os.path.dirname(here)), "include", "dinox.h"))
    hsh = hashlib.sha256()
    for f in files:
        try:
            with open(f, "rb") as fh:
                hsh.update(os.path.basename(f).encode() + b"\0" + fh.read())
        except OSError:
            pass
    return hsh.hexdigest()[:16]
