#!/usr/bin/env python3
"""Turn gpurun_out/refresh/ (tools/refresh_profiles.sh) into the tracked profile artefacts of a round:
profiles/rNN_bench_bs256_kernel_stats.csv, profiles/rNN_pmc_traffic.json and profiles/rNN_bench_bs256_summary.md.
usage: make_profile_summary.py [round tag, default r01]"""
import csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "refresh")
prof = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)       # (gpurun MERGES: older runs' files stay)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[-1]


def json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    raise SystemExit(f"no JSON line in {path}")


stats = one("stats/**/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(prof, f"{tag}_bench_bs256_kernel_stats.csv"))
pmc_json = os.path.join(prof, f"{tag}_pmc_traffic.json")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), one("pmc_fetch/**/*_counter_collection.csv"),
                one("pmc_write/**/*_counter_collection.csv"), pmc_json], check=True, stdout=subprocess.DEVNULL)
pmc = json.load(open(pmc_json))
rows = list(csv.DictReader(open(stats)))
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
prof_line = json.loads(json_line(os.path.join(src, "bench_prof.log")))
default_line = json_line(os.path.join(src, "bench_default.log"))
steps = prof_line["steps"] + prof_line["warmup"]
# every optimiser step of the process launches adamw_ema_kernel once -- the timed + warm-up steps AND the few extra ones bench.py runs on an
# empty queue for host_enqueue_ms_per_step: count what the trace holds
opt_calls = sum(int(r["Calls"]) for r in rows if "adamw_ema" in r["Name"])
steps = opt_calls or steps
fam = {}
for r in rows:
    f = r["Name"].replace("void ", "").split("(")[0].split("<")[0]
    d = fam.setdefault(f, [0, 0.0])
    d[0] += int(r["Calls"])
    d[1] += float(r["TotalDurationNs"])
dom = prof_line["roofline"]["kernel"]
dom_key = next(k for k in fam if k.endswith(dom))
md = [f"# rocprofv3 --kernel-trace --stats summary, round {tag[1:]}", "",
      "Command (MI355X, 1 GPU): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-step-split`",
      f"({steps} optimiser steps in the trace ({prof_line['warmup']} warm-up + {prof_line['steps']} timed + the empty-queue host-enqueue steps) of ViT-S/16 224 bs256 = 512 views, bf16 mode, everything on one stream; HIP-event kernel timing active as in the "
      f"default bench run).  Raw CSV: `{tag}_bench_bs256_kernel_stats.csv`.  Regenerate: `tools/refresh_profiles.sh` on the GPU box, then "
      "`tools/make_profile_summary.py`.", "",
      "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:30]:
    md.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
              f"{100 * float(r['TotalDurationNs']) / total_ns:.1f} |")
md += ["", f"Total kernel time {total_ns / 1e6:.1f} ms over {steps} steps = {total_ns / 1e6 / steps:.1f} ms/step.", "",
       "bench.py line of the same (profiled) run:", "", "```", json.dumps(prof_line), "```", "",
       f"Dominant kernel family `{dom}` (all epilogue/tile variants): {fam[dom_key][0]} launches, {fam[dom_key][1] / 1e6:.1f} ms, average "
       f"**{fam[dom_key][1] / fam[dom_key][0] / 1e3:.1f} us** per launch in the rocprofv3 trace; bench.py's own HIP-event figure over its timed steps: "
       f"**{prof_line['roofline']['avg_launch_us']} us** (`roofline.avg_launch_us` above).", "",
       "Default bench line (`python bench.py`, unprofiled, with the CPU baseline leg) of the same box:", "", "```", default_line, "```", "",
       "## HBM traffic (separate PMC passes)", "",
       "`rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step-split --no-kernel-timing` and the same with `--pmc WRITE_SIZE` "
       f"(never combined with a trace domain); summary by `tools/pmc_summary.py` in `{tag}_pmc_traffic.json` (FETCH_SIZE doubled: gfx950 reports half the "
       "bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md; sanity check: `ln_fwd_kernel` must read its 158 MB fp32 input once).", "",
       "| kernel family | launches | HBM read / launch | HBM write / launch |", "|---|---|---|---|"]
for k, v in [kv for kv in pmc.items() if not kv[0].startswith("_")][:14]:
    md.append(f"| `{k}` | {v['launches']} | {v['fetch_bytes_per_launch_x2_corrected'] / 1e6:.1f} MB | {v['write_bytes_per_launch'] / 1e6:.1f} MB |")
md.append("")
sq = sorted(glob.glob(os.path.join(src, "pmc_sq/**/*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
mfma_json = os.path.join(prof, f"{tag}_pmc_mfma.json")
if sq:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_mfma.py"), sq[-1], mfma_json], check=True, stdout=subprocess.DEVNULL)
if os.path.exists(mfma_json):
    mf = json.load(open(mfma_json))
    md += ["## MFMA-pipe utilisation and wave states (separate SQ counter pass)", "",
           "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py "
           f"--steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step-split --no-kernel-timing`; summary by `tools/pmc_mfma.py` in `{tag}_pmc_mfma.json`.  MFMA utilisation = "
           "SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.03 GHz x 1024 SIMDs); wave states are fractions of SQ_WAVE_CYCLES (parked = s_waitcnt / "
           "barrier, issue-stalled = an instruction is ready but cannot issue, issuing = an instruction issues).", "",
           "| kernel family | launches | MFMA utilisation | parked | issue-stalled | issuing |", "|---|---|---|---|---|---|"]
    for k, v in [kv for kv in mf.items() if not kv[0].startswith("_")][:8]:
        md.append(f"| `{k}` | {v['launches']} | {100 * v['mfma_util']:.1f} % | {v['wave_wait_any']:.2f} | {v['wave_wait_inst']:.2f} | {v['wave_active_inst']:.2f} |")
    md.append("")
open(os.path.join(prof, f"{tag}_bench_bs256_summary.md"), "w").write("\n".join(md))
print("\n".join(md[-20:]))
