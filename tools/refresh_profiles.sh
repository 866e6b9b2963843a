#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, the rocprofv3 kernel-trace summary of the same command, and the two
# separate PMC passes (FETCH_SIZE / WRITE_SIZE) that price HBM traffic.  Everything lands under gpurun_out/refresh/;
# tools/make_profile_summary.py turns it into the tracked files under profiles/.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/refresh
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_default.log" 2> "$OUT/bench_default.err"
echo "[refresh] default bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-step-split > "$OUT/bench_prof.log" 2> "$OUT/bench_prof.err"
echo "[refresh] kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step-split --no-kernel-timing > "$OUT/pmc_fetch.log" 2>&1
echo "[refresh] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step-split --no-kernel-timing > "$OUT/pmc_write.log" 2>&1
echo "[refresh] WRITE_SIZE pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step-split --no-kernel-timing > "$OUT/pmc_sq.log" 2>&1
echo "[refresh] SQ pass done"
find "$OUT" -name "*_kernel_trace.csv" -delete        # large; the stats CSV is what gets committed
find "$OUT" -name "*.db" -delete
