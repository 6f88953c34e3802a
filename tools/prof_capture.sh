#!/bin/bash
# GPU box: kernel trace + stats of the capture harness (one warm-up round, one timed round of 1024 streams, files in /dev/shm).
# usage: tools/prof_capture.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_capture.py --frames 1024 --streams 1024 --warmup 1024 --cpu-frames 0 --out /dev/shm/rh_cap_prof > $OUT/trace.log 2>&1 || exit 1
rm -rf /dev/shm/rh_cap_prof
tail -1 $OUT/trace.log > $OUT/capture_line_under_tracer.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
