#!/bin/bash
# Experiment: how much of the C1 tile kernel's time is texture fetching?  Same scene, smaller textures / nearest filter.
for args in "" "--texture-size 64" "--texture-size 16" "--filter nearest" "--filter nearest --texture-size 16"; do
  echo "== $args"
  timeout -k 10 120 python bench.py --capture-frames 0 --no-cpu-baseline --steps 10 --warmup 3 $args | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'tile_ms', round(d['roofline']['kernel_ms'],4), 'bin_ms', round(d['roofline']['binning_ms'],4))" || exit 1
done
