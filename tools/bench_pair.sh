#!/bin/bash
# C1 (default) and C2 (10 M tiny quads) tile-kernel timings of the current build, one line each.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python3 bench.py --capture-frames 0 --no-cpu-baseline > gpurun_out/pair_c1.json 2> gpurun_out/pair_c1.err &&
timeout -k 10 200 python3 bench.py --workload c2 --quads 10000000 --capture-frames 0 --no-cpu-baseline > gpurun_out/pair_c2.json 2> gpurun_out/pair_c2.err &&
cat gpurun_out/pair_c1.json gpurun_out/pair_c2.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['metric'], '%.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'tile kernel ms %.3f' % d['roofline']['kernel_ms'])"
