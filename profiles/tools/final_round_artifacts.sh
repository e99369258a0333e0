#!/bin/bash
# Round artifacts: the default bench command (JSON line), its rocprofv3 kernel-trace summary, the bf16-mode line, the N = 2
# self-launch rehearsal on one device.  Everything lands under gpurun_out/ and is copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err
tail -c 600 gpurun_out/r03_bench_line.json; echo
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-cpu-baseline > gpurun_out/r03_bench_profiled_line.json 2>/dev/null
cp gpurun_out/prof_bench/*/*kernel_stats.csv gpurun_out/r03_bench_ffhq_gblur_heun30_b8_kernel_stats.csv
python3 bench.py --no-cpu-baseline --unet-dtype fp16x3 > gpurun_out/r03_bench_line_fp16x3_mode.json 2>/dev/null
python3 bench.py --no-cpu-baseline --unet-dtype fp16 > gpurun_out/r03_bench_line_fp16_mode.json 2>/dev/null
python3 bench.py --no-cpu-baseline --unet-dtype bf16 > gpurun_out/r03_bench_line_bf16_mode.json 2>/dev/null
python3 bench.py --no-cpu-baseline --unet-dtype bf16x3 > gpurun_out/r03_bench_line_bf16x3_mode.json 2>/dev/null
FH_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --batch 4 --no-cpu-baseline > gpurun_out/r03_bench_line_gpus2_rehearsal.json 2>/dev/null
head -c 300 gpurun_out/r03_bench_line_bf16_mode.json; echo; head -c 300 gpurun_out/r03_bench_line_gpus2_rehearsal.json; echo
head -12 gpurun_out/r03_bench_ffhq_gblur_heun30_b8_kernel_stats.csv | cut -c1-160
