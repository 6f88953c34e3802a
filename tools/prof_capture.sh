#!/bin/bash
# GPU box: kernel trace + stats of the capture harness (one warm-up round, one timed round of 1024 streams, files in
# /dev/shm), then one counter pass each for FETCH_SIZE and WRITE_SIZE over the same command; the per-launch averages of
# rh_lz4f_compress go to <out>/capture_pmc_avg_per_launch.json (both counters in KiB, uncorrected).
# usage: tools/prof_capture.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python3 tools/bench_capture.py --frames 1024 --streams 1024 --warmup 1024 --cpu-frames 0 --out /dev/shm/rh_cap_prof"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
rm -rf /dev/shm/rh_cap_prof
tail -1 $OUT/trace.log > $OUT/capture_line_under_tracer.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
rm -rf /dev/shm/rh_cap_prof
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
rm -rf /dev/shm/rh_cap_prof
python3 - <<PY
import csv, glob, json
out = {"streams": 1024}
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)[0]
    rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "lz4f_compress" in r["Kernel_Name"] and r["Counter_Name"] == name]
    out[name] = sum(rows) / max(1, len(rows)); out["launches_" + name] = len(rows)      # (one row per launch and counter)
json.dump({"rh_lz4f_compress": out}, open("$OUT/capture_pmc_avg_per_launch.json", "w"))
print(out)
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
