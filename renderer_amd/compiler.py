"""Offline shader build (reference src/compiler.rs:7-33): recursively compiles `<name>.vert.hip` / `<name>.frag.hip`
pairs to gfx950 code objects `<name>.vert.co` / `<name>.frag.co` next to the sources, with hipcc instead of shaderc.

A fragment object embeds the vertex function of the same stem, because triangle setup re-runs the vertex function
inside the tile kernel rather than reading transformed vertices back from HBM."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SHADER_INCLUDE = os.path.join(HERE, "csrc", "shader")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--genco", "--no-gpu-bundle-output", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
         # packed-f32 instructions issue at half the rate of scalar f32 ones on gfx950, and forming the pairs costs
         # moves: keep the SLP vectoriser from creating them
         "-fno-slp-vectorize", "-Wno-pass-failed", "-I", SHADER_INCLUDE]


class Compiler:
    @staticmethod
    def compile_shaders(directory, force=False, jobs=None):
        """Compiles every shader source under `directory`; out-of-date ones are built by a few hipcc processes at once
        (a fragment object holds a dozen kernels and takes the better part of a minute)."""
        from concurrent.futures import ThreadPoolExecutor
        sources = []
        for root, _dirs, files in os.walk(directory):
            for name in sorted(files):
                if name.endswith(".vert.hip") or name.endswith(".frag.hip"):
                    sources.append(os.path.join(root, name))
        jobs = jobs or max(1, min(8, (os.cpu_count() or 2) // 2, len(sources)))
        with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
            built = list(pool.map(lambda path: Compiler.compile_shader(path, force=force), sources))
        return [b for b in built if b]

    @staticmethod
    def compile_shader(path, force=False):
        path = os.path.abspath(path)
        stem, stage = path[:-len(".vert.hip")], path[-len(".vert.hip"):-len(".hip")]     # ".vert" | ".frag"
        vert = stem + ".vert.hip"
        out = path[:-len(".hip")] + ".co"
        if not os.path.exists(vert):
            raise FileNotFoundError(f"{path}: no vertex source {vert} with the same stem")
        deps = [path, vert] + [os.path.join(SHADER_INCLUDE, f) for f in os.listdir(SHADER_INCLUDE)] + \
               [os.path.join(HERE, "csrc", "rh_abi.h")]
        if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
            return out
        tag = os.path.basename(stem)
        defs = [f'-DRH_VERTEX_SOURCE="{vert}"', f'-DRH_VERTEX_TAG="{tag}"']
        if stage == ".vert":
            unit = os.path.join(SHADER_INCLUDE, "rh_stage_vertex.hip")
        else:
            unit = os.path.join(SHADER_INCLUDE, "rh_stage_fragment.hip")
            defs.append(f'-DRH_FRAGMENT_SOURCE="{path}"')
        subprocess.check_call([HIPCC] + FLAGS + defs + [unit, "-o", out])
        return out
