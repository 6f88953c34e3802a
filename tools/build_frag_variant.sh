#!/bin/bash
# Builds the fragment code object of shader $SHADER (default: hello) with extra compiler flags into /tmp/<name>.frag.co (kernel experiments; run
# with RH_FRAG_CO_OVERRIDE=/tmp/<name>.frag.co).  usage: tools/build_frag_variant.sh <name> [flags...]
set -e
NAME=$1; shift
D=renderer_amd/csrc/shader; S=renderer_amd/shaders
# (FRAG_SRC: another version of the fragment stage's source, e.g. the previous commit's, to time against this tree's)
SRC=${FRAG_SRC:-$D/rh_stage_fragment.hip}
/opt/rocm/bin/hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed \
  -I $D "$@" "-DRH_VERTEX_SOURCE=\"$PWD/$S/${SHADER:-hello}.vert.hip\"" "-DRH_VERTEX_TAG=\"${SHADER:-hello}\"" "-DRH_FRAGMENT_SOURCE=\"$PWD/$S/${SHADER:-hello}.frag.hip\"" \
  $SRC -o /tmp/$NAME.frag.co
