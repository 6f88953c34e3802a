#!/bin/bash
# GPU box: SQ counter passes of the default bench (no trace), per-launch averages printed for the tile kernel.
# usage: tools/prof_sq.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --capture-frames 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -- $B "$@" > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -- $B "$@" > $OUT/pmc2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH --output-format csv -d $OUT/pmc5 -- $B "$@" > $OUT/pmc5.log 2>&1 || true
python3 tools/prof_summary.py $OUT | grep -v "^\"" | tail -60
