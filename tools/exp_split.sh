#!/bin/bash
# GPU box: the C1 bench line for each way of cutting the last tiles into bands (RH_TAIL_SPLIT values given as arguments).
mkdir -p gpurun_out/split
for eb in 1 0; do for m in "$@"; do
  RH_EARLY_BINNING=$eb RH_TAIL_SPLIT=$m timeout -k 10 200 python bench.py --capture-frames 0 --no-cpu-baseline $BENCH_ARGS > gpurun_out/split/b_${eb}_$m.json 2> gpurun_out/split/b_${eb}_$m.err || exit 1
  python - <<PY
import json; d=json.load(open("gpurun_out/split/b_${eb}_$m.json")); r=d["roofline"]
print("early $eb split $m: ms_per_step %.4f tile_ms %.4f frac %.4f" % (d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
done; done
