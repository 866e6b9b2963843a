#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs) into per-kernel HBM traffic per launch.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half the bytes of wide coalesced 16-B/lane
streaming reads -> doubled here; WRITE_SIZE is exact for 16-B/lane streaming stores.  Counter unit: KiB.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def _fingerprint():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dino-x_amd", "dinox", "hostinfo.py")
    spec = importlib.util.spec_from_file_location("_dinox_hostinfo", path)      # (plain Python: no GPU, no library load)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_fingerprint()


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        fam = name.split("<")[0]
        agg[fam].append(float(r["Counter_Value"]))
    return agg


fe, wr = load(sys.argv[1]), load(sys.argv[2])
out = {}
for fam in sorted(fe, key=lambda k: -sum(fe[k])):
    n = len(fe[fam])
    f = 2.0 * 1024.0 * sum(fe[fam]) / n
    w = 1024.0 * sum(wr.get(fam, [0.0])) / max(1, len(wr.get(fam, [0.0])))
    out[fam] = {"launches": n, "fetch_bytes_per_launch_x2_corrected": round(f), "write_bytes_per_launch": round(w),
                "hbm_bytes_per_launch": round(f + w)}
out["_source_fingerprint"] = _fingerprint()      # the kernel sources these figures belong to (bench.py checks it)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in [kv for kv in out.items() if not kv[0].startswith("_")][:12]:
    print(f"{k:40s} launches {v['launches']:5d}  fetch {v['fetch_bytes_per_launch_x2_corrected'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")
