#!/bin/bash
# Diagnostic: builds the hello fragment code object with -DRH_WG_TIMES and runs two C1 frames with it; every fourth
# workgroup of the tile kernel has its start and end time printed (100 MHz clock; lines on stderr).  tools/wg_times.py turns the lines of the
# last launch into a summary: when the whole tiles and the bands start and end, how many workgroups run at a time.
# usage (GPU box, repo root): tools/wg_times.sh [bench args] > lines.txt
set -e
tools/build_frag_variant.sh wgtimes -DRH_WG_TIMES > /dev/null
RH_EARLY_BINNING=${RH_EARLY_BINNING:-0} RH_FRAG_CO_OVERRIDE=/tmp/wgtimes.frag.co timeout -k 10 120 python bench.py --capture-frames 0 --no-cpu-baseline --no-alone --steps 2 --warmup 1 "$@"
