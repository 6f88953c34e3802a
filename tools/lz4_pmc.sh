#!/bin/bash
# Counter passes over one round of the GPU LZ4F kernel (64 streams, one 4K frame each); per-kernel sums are printed.
# usage (GPU box): tools/lz4_pmc.sh <tag>
TAG=${1:-lz4pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/$TAG/p1 -- python3 tools/bench_capture.py --frames 64 --streams 64 --warmup 2 --cpu-frames 0 > gpurun_out/$TAG.1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/$TAG/p2 -- python3 tools/bench_capture.py --frames 64 --streams 64 --warmup 2 --cpu-frames 0 > gpurun_out/$TAG.2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    f=glob.glob("gpurun_out/$TAG/%s/*/*counter_collection.csv"%p)[0]
    acc=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(f)):
        if "lz4f_compress" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print({k:"%.3g"%v for k,v in sorted(acc.items())})
PY
