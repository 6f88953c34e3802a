#!/bin/bash
# GPU box: profile + default bench line of C1, then the same for C2; summaries under gpurun_out/<tag>/ and gpurun_out/<tag>c2/.
# usage: tools/final_lines.sh <tag>
set -e
TAG=$1; mkdir -p gpurun_out/$TAG gpurun_out/${TAG}c2
show() { python3 - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(round(d["value"] / 1e6, 1), round(d["ms_per_step"], 4), round(r["kernel_ms"], 4), round(r["frac"], 4), r.get("alone"), d.get("parity"), (d.get("capture") or {}).get("value"))
PY
}
bash tools/prof.sh $TAG
timeout -k 10 400 python3 bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench.err
show gpurun_out/$TAG/bench_line.json
bash tools/prof.sh ${TAG}c2 --workload c2
timeout -k 10 300 python3 bench.py --workload c2 --capture-frames 0 > gpurun_out/${TAG}c2/bench_line.json 2> gpurun_out/${TAG}c2/bench.err
show gpurun_out/${TAG}c2/bench_line.json
