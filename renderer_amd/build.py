"""Builds librenderer_hip.so (host C ABI + shader-independent gfx950 kernels) and the bundled shader code objects.
Run as `python -m renderer_amd.build`.  hipcc cross-compiles for gfx950 without a GPU present."""
import os
import subprocess
import sys

from .compiler import HIPCC, Compiler

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librenderer_hip.so")
SOURCES = ["rh_api.cpp", "rh_kernels.hip", "rh_codec.hip", "rh_png.hip"]


def build_library(force=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(HERE, "..", "include", "renderer_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
           "-Wno-unused-result"] + srcs + ["-o", LIB]
    subprocess.check_call(cmd)
    return LIB


def build_examples(force=False):
    """The C++ host mirror's examples: examples/quads_main_thread.cpp (the reference's quads example call for call) and
    examples/quads_render_thread.cpp (the same through the RenderThread proxy)."""
    root = os.path.join(HERE, "..")
    built = []
    for name, extra in (("quads_main_thread", []), ("quads_render_thread", ["-pthread"])):
        src = os.path.join(root, "examples", name + ".cpp")
        exe = os.path.join(root, "examples", name)
        deps = [src, LIB, os.path.join(root, "include", "renderer_hip.h")] + \
            [os.path.join(HERE, "cpp", f) for f in os.listdir(os.path.join(HERE, "cpp"))]
        if force or not os.path.exists(exe) or any(os.path.getmtime(exe) < os.path.getmtime(d) for d in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(root, "include"), "-I", os.path.join(HERE, "cpp"),
                                   src, "-o", exe, "-L", HERE, "-lrenderer_hip", "-lz", "-Wl,-rpath,$ORIGIN/../renderer_amd"] + extra)
        built.append(exe)
    return built


def build_all(force=False):
    lib = build_library(force)
    shaders = Compiler.compile_shaders(os.path.join(HERE, "shaders"), force=force)
    return lib, shaders + build_examples(force)


if __name__ == "__main__":
    lib, shaders = build_all(force="--force" in sys.argv)
    print(lib)
    for s in shaders:
        print(s)
