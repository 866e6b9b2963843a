#!/bin/bash
# Real-data rate of the training script on one MI355X: PNG files decoded per sample (the reference's way) against the decoded-slice
# cache (--stack-cache).  Writes gpurun_out/png_*.log; the summary lines go to profiles/rNN_cli_end_to_end.txt by hand.
#   tools/png_end_to_end.sh [series=64] [slices=64] [size=512] [workers=12]
set -e
SERIES=${1:-64}; SLICES=${2:-64}; SIZE=${3:-512}; WORKERS=${4:-12}
D=/tmp/dinox_png_$$
mkdir -p gpurun_out
python tools/make_png_dataset.py $D/data --series $SERIES --slices $SLICES --size $SIZE --workers 16 | tee gpurun_out/png_dataset.log
COMMON="--config vit-small --vit-patch 16 --img-size 224 --batch-size 256 --scale-aware --amp --index-csv $D/data/index.csv --num-workers $WORKERS
        --warmup-steps 5 --koleo-weight 0.1 --gpu-views --run-dir $D/runs --ckpt-every 100000"
if [ -z "$SKIP_NOCACHE" ]; then
DINOX_CLI_PROFILE=40 timeout -k 10 400 python dino-x_amd/scripts/phase5_big_run.py $COMMON --max-steps 120 > gpurun_out/png_nocache.log 2>&1
tail -4 gpurun_out/png_nocache.log
fi
DINOX_CLI_PROFILE=${PROFILE_FROM:-300} timeout -k 10 500 python dino-x_amd/scripts/phase5_big_run.py $COMMON --max-steps ${STEPS:-900} --stack-cache $D/cache --stack-cache-prefill \
    > gpurun_out/png_cache.log 2>&1
grep stack_cache gpurun_out/png_cache.log
tail -4 gpurun_out/png_cache.log
rm -rf $D
