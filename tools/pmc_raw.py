#!/usr/bin/env python3
"""Per-kernel-family sums of whatever counters a rocprofv3 --pmc pass collected, per launch (plus the dispatch duration).
usage: pmc_raw.py <counter_collection.csv> [<more.csv> ...] <out.json>"""
import collections, csv, json, sys


def _fingerprint():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dino-x_amd", "dinox", "hostinfo.py")
    spec = importlib.util.spec_from_file_location("_dinox_hostinfo", path)      # (plain Python: no GPU, no library load)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_fingerprint()


agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(collections.Counter)
for path in sys.argv[1:-1]:
    seen = set()
    for r in csv.DictReader(open(path)):
        fam = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), fam)
        if key not in seen:
            seen.add(key)
            launches[path][fam] += 1
            agg[fam]["_ns:" + path] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
first = sys.argv[1]
for fam in sorted(agg, key=lambda f: -agg[f].get("_ns:" + first, 0)):
    n = max(launches[first][fam], 1)
    row = {"launches": n, "us_per_launch_under_pmc": round(agg[fam].get("_ns:" + first, 0) / n / 1e3, 1)}
    for k, v in agg[fam].items():
        if not k.startswith("_ns"):
            row[k] = round(v / n, 1)
    out[fam] = row
out["_source_fingerprint"] = _fingerprint()      # the kernel sources these figures belong to (bench.py checks it)
json.dump(out, open(sys.argv[-1], "w"), indent=1)
for fam, row in list(out.items())[:10]:
    print(fam, row)
