#!/bin/bash
# per-kernel, per-grid breakdown of one UNet forward + input-VJP at batch 8 (FFHQ-256 architecture by default)
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ARCH=${1:-ffhq}
OUT=gpurun_out/unet_trace_$ARCH${TAG:-}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 profiles/tools/unet_once.py $ARCH 8 > $OUT/run.log 2>&1
python3 profiles/tools/trace_by_grid.py $(find $OUT -name '*kernel_trace.csv' | head -1) 3 90 > $OUT/by_grid.txt
cat $OUT/by_grid.txt
