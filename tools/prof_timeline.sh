#!/bin/bash
# GPU box: kernel trace of the default bench (no capture leg, no untimed leg); the kernel stats and the timeline of three
# frames of the timed region go to gpurun_out/<tag>/.   usage: tools/prof_timeline.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-alone --capture-frames 0 "$@" > $OUT/trace.log 2>&1 || exit 1
tail -1 $OUT/trace.log > $OUT/bench_line_under_tracer.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 tools/trace_frames.py $OUT/trace 20 3 > $OUT/frames_timeline.txt
rm -rf $OUT/trace
cat $OUT/frames_timeline.txt; head -12 $OUT/kernel_stats.csv
