#!/bin/bash
# GPU box: times the C1 frame with fragment code objects built with different flags.  usage: tools/exp_variants.sh "name:flags" ...
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  tools/build_frag_variant.sh $name $flags || exit 1
  echo "== $name ($flags)"
  RH_FRAG_CO_OVERRIDE=/tmp/$name.frag.co timeout -k 10 120 python bench.py --capture-frames 0 --no-cpu-baseline --steps 10 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'tile_ms', round(d['roofline']['kernel_ms'],4), 'bin_ms', round(d['roofline']['binning_ms'],4))" || exit 1
done
