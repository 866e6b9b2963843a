#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel family from one rocprofv3 SQ counter pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (32 per v_mfma_f32_32x32x16_bf16); 1024 SIMDs (256 CUs x 4).
The denominator is the dispatch's duration (its timestamps in the same CSV) times the shader clock under this load, 2.03 GHz, measured
inside the GEMM with s_memtime against s_memrealtime -- GRBM_GUI_ACTIVE over-counts short dispatches and is not used.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave:
their ratios say where resident waves spend their time (parked on s_waitcnt/barrier vs issue-stalled vs issuing).
usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import collections, csv, json, sys


def _fingerprint():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dino-x_amd", "dinox", "hostinfo.py")
    spec = importlib.util.spec_from_file_location("_dinox_hostinfo", path)      # (plain Python: no GPU, no library load)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_fingerprint()


SIMDS = 1024
CLOCK_GHZ = 2.03
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    fam = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), fam)
    if key not in seen:
        seen.add(key)
        launches[fam] += 1
        agg[fam]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for fam, c in sorted(agg.items(), key=lambda kv: -kv[1].get("_ns", 0)):
    act = c.get("_ns", 0.0) * CLOCK_GHZ
    if act <= 0:
        continue
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out[fam] = {"launches": launches[fam], "us_per_launch_under_pmc": round(c["_ns"] / launches[fam] / 1e3, 1), "assumed_clock_ghz": CLOCK_GHZ,
                "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * SIMDS), 4),
                "wave_wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wave_wait_inst": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                "wave_active_inst": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)}
out["_source_fingerprint"] = _fingerprint()      # the kernel sources these figures belong to (bench.py checks it)
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in [kv for kv in out.items() if not kv[0].startswith("_")][:12]:
    print(f"{k:36s} launches {v['launches']:5d}  MFMA util {100 * v['mfma_util']:5.1f} %   waves: parked {v['wave_wait_any']:.2f}  issue-stalled {v['wave_wait_inst']:.2f}  issuing {v['wave_active_inst']:.2f}")
