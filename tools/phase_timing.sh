#!/bin/bash
# Diagnostic: builds the hello fragment code object with -DRH_PHASE_TIMING and runs one C1 frame with it; sample
# workgroups print the cycles their waves spent per phase (see RH_PHASE_* in csrc/shader/rh_stage_fragment.hip).
# usage (GPU box, repo root): tools/phase_timing.sh [bench args]
set -e
D=renderer_amd/csrc/shader; S=renderer_amd/shaders
/opt/rocm/bin/hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed \
  -I $D -DRH_PHASE_TIMING "-DRH_VERTEX_SOURCE=\"$PWD/$S/hello.vert.hip\"" '-DRH_VERTEX_TAG="hello"' "-DRH_FRAGMENT_SOURCE=\"$PWD/$S/hello.frag.hip\"" \
  $D/rh_stage_fragment.hip -o /tmp/hello_phase.frag.co
RH_FRAG_CO_OVERRIDE=/tmp/hello_phase.frag.co timeout -k 10 120 python bench.py --capture-frames 0 --no-cpu-baseline --steps 1 --warmup 0 "$@"
