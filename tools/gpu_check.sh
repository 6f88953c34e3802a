#!/bin/bash
# GPU box: parity tests of the rasteriser, then the headline bench line (no capture leg, no CPU baseline).
# usage: tools/gpu_check.sh <tag> [quick]
TAG=${1:-check}; mkdir -p gpurun_out/$TAG
if [ "$2" = "quick" ]; then SEL="tests/test_gpu_parity.py tests/test_gpu_binning_load.py"; else SEL="tests -m gpu"; fi
timeout -k 10 900 python -u -m pytest $SEL -x -q --timeout 180 --timeout-method thread > gpurun_out/$TAG/tests.log 2>&1
rc=$?; tail -3 gpurun_out/$TAG/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --capture-frames 0 --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || exit 1
python - <<PY
import json; d=json.load(open("gpurun_out/$TAG/bench.json")); r=d["roofline"]
print("ms_per_step %.4f tile_ms %.4f setup_ms %.4f bin_ms %.4f frac %.4f value %.1fM" % (d["ms_per_step"], r["kernel_ms"], r["setup_ms"] or 0, r["binning_ms"] or 0, r["frac"], d["value"]/1e6))
PY
