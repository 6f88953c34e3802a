#!/bin/bash
# Parity slice + timing of the GPU LZ4F kernel: codec tests, 100 s of the codec fuzz sweep, then the kernel's average
# duration per round of 64 4K frames under rocprofv3 --kernel-trace --stats.  usage (GPU box): tools/lz4_time.sh <tag> [fuzz seed]
set -o pipefail
TAG=${1:-lz4}; SEED=${2:-80000}
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_codec.py tests/test_gpu_fuzz.py -m gpu -x -q -k "lz4 or streams or roundtrip" --timeout 300 --timeout-method thread 2>&1 | tail -2 &&
timeout -k 10 300 python3 -u tests/fuzz_codec.py --rounds 3000 --seed $SEED --max-seconds 100 2>&1 | tail -1 &&
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 tools/bench_capture.py --frames 64 --streams 64 --cpu-frames 0 > gpurun_out/$TAG.log 2>&1
grep -h lz4f gpurun_out/$TAG/*/*kernel_stats.csv | cut -c1-120
