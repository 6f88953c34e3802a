#!/bin/bash
# Host-side AddressSanitizer + UBSan check of the C ABI: builds the C++ quads example together with the library sources
# into ONE binary with the HOST code instrumented (device code is not: GPU sanitizers are unavailable), to be run on a
# GPU box:   tools/asan_example.sh build            (here, no GPU needed)
#            tools/asan_example.sh run [msaa]       (on the GPU box, from the repo root)
set -e
cd "$(dirname "$0")/.."
case "$1" in
build)
    mkdir -p build_asan
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -Wno-unused-result \
        -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -I include -I renderer_amd/cpp \
        examples/quads_main_thread.cpp renderer_amd/csrc/rh_api.cpp renderer_amd/csrc/rh_kernels.hip \
        renderer_amd/csrc/rh_codec.hip renderer_amd/csrc/rh_png.hip -lz -o build_asan/quads_asan
    ;;
run)
    out=$(mktemp -d)
    ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 timeout -k 10 200 \
        build_asan/quads_asan renderer_amd/shaders tests/golden/letter_f.png "$out" 20 480 270 ${2:-1}
    ;;
*) echo "usage: $0 build | run [msaa samples]"; exit 2 ;;
esac
