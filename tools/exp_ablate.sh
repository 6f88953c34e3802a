#!/bin/bash
# GPU box: times the C1 tile kernel (every launch on one stream) with fragment code objects built with different flags,
# e.g. the diagnostic builds -DRH_EXP=1 (no texture read), 2 (read, no filter), 3 (no blend arithmetic).  All variants are
# built first and then timed in turn, ROUNDS times over (clocks drift and boxes differ: only figures of one run compare).
# A name may end in @VAR=VALUE: that variable is set for the variant's timed runs (e.g. cur@RH_TAIL_SPLIT=3).
# usage: [ROUNDS=2] tools/exp_ablate.sh "name:flags" ...
export RH_EARLY_BINNING=${RH_EARLY_BINNING:-0}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; name=${name%%@*}
  # (a variant whose name is a directory under tools/data/ holding rh_stage_fragment.hip is built from that file: an
  # earlier or experimental version of the source put there by hand; prev* = tools/data/prev/)
  case $name in prev*) export FRAG_SRC=tools/data/prev/rh_stage_fragment.hip;; *) unset FRAG_SRC;; esac
  [ -f tools/data/$name/rh_stage_fragment.hip ] && export FRAG_SRC=tools/data/$name/rh_stage_fragment.hip
  tools/build_frag_variant.sh $name $flags 2>&1 | grep -v "warning\|^$" ; [ -f /tmp/$name.frag.co ] || exit 1
done
for round in $(seq ${ROUNDS:-2}); do
  for spec in "$@"; do
    name=${spec%%:*}; setting=; case $name in *@*) setting=${name#*@}; name=${name%%@*};; esac
    env $setting RH_FRAG_CO_OVERRIDE=/tmp/$name.frag.co timeout -k 10 120 python bench.py --capture-frames 0 --no-cpu-baseline --no-alone --steps 20 --warmup 5 ${BENCH_ARGS} | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-18s' % ('$name', '$setting'), 'ms_per_step', round(d['ms_per_step'],4), 'tile_ms', round(d['roofline']['kernel_ms'],4), 'bin_ms', round(d['roofline']['binning_ms'] or 0,4), 'setup_ms', round(d['roofline']['setup_ms'] or 0,4))" || exit 1
  done
done
