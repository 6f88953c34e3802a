#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + two SQ counter passes + FETCH_SIZE and WRITE_SIZE passes (one run each); summaries under gpurun_out/<tag>/.
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-alone "$@" > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone "$@" > $OUT/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone "$@" > $OUT/pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone "$@" > $OUT/pmc3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone "$@" > $OUT/pmc4.log 2>&1
python3 tools/prof_summary.py $OUT
# keep the summaries, drop the raw traces (what comes back from the box is limited to 64 MiB)
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
grep -h '^{"metric"' $OUT/trace.log > $OUT/bench_line_under_tracer.json || true
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
