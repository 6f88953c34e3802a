#!/bin/bash
# GPU box: kernel trace + stats of the default bench (no capture leg); prints the kernel stats table.
# usage: tools/prof_trace.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --capture-frames 0 "$@" > $OUT/trace.log 2>&1 || exit 1
tail -1 $OUT/trace.log > $OUT/bench_line_under_tracer.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
