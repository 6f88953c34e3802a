#!/bin/bash
# Duration of one GPU LZ4F round against the number of streams in it (one 4K frame per stream, nothing else on the GPU
# while it runs): rocprofv3 --kernel-trace --stats of tools/bench_capture.py with frames == streams.
# usage (GPU box): tools/lz4_scale.sh [streams...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in ${@:-64 256 512}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lz4_scale_$S -- python3 tools/bench_capture.py --frames $S --streams $S --cpu-frames 0 > gpurun_out/lz4_scale_$S.log 2>&1 || exit 1
  echo "streams $S: $(grep -h lz4f gpurun_out/lz4_scale_$S/*/*kernel_stats.csv | cut -c1-100)"
done
