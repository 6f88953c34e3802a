#!/usr/bin/env python3
"""BASELINE.json configs[3] on one GPU: render-to-texture two-pass at 16384 x 16384.  Pass 1 draws 4 M quads of 64x64 px
(pre-multiplied alpha, linear filter) into texture A; pass 2 draws one quad covering target B that samples A (linear).
A is bound to pass 2 as it was rendered (no copy).  Prints one JSON line with the HIP-event time of each pass."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--quads", type=int, default=4_000_000)
    ap.add_argument("--side", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import scenes
    from renderer_amd import Renderer
    S = args.size
    s1 = scenes.random_quads_scene(args.quads, args.side, S, S, seed=3)
    r = Renderer(64, 64)
    h1 = scenes.HipScene(s1, renderer=r)                    # pass 1 -> its own RGBA8 texture target A (S x S)
    a_tex = h1.target_texture
    # pass 2: the hello program with A as its texture; A gets a linear sampler by re-wrapping the same device texture
    vert, frag = scenes.shader_code("hello")
    a_position, a_tex_coord, i_offset = r.attribute(0, 2), r.attribute(1, 2), r.instanced()
    from renderer_amd.texture import Texture
    a_view = Texture.__new__(Texture)                       # same rh_texture handle, plus a sampler (a view of A)
    a_view.__dict__.update(a_tex.__dict__)
    from renderer_amd import _ffi
    import ctypes as C
    smp = C.c_void_p()
    _ffi.check(_ffi.lib().rh_sampler_create(r.device, int(Renderer.linear_filtering()), C.byref(smp)))
    a_view.sampler = smp
    _ffi.lib().rh_texture_retain(a_view.handle)
    b_tex = r.texture(S, S, 1, Renderer.nearest_filtering(), Renderer.rgba_u8(), True, True, False)
    program = r.program(vert, frag, [a_position, a_tex_coord], [i_offset], [], [(a_view, Renderer.visible_to_fragment_shader())])
    p2 = r.pipeline(program, Renderer.pre_multiplied_blend(), Renderer.triangle_strip_primitive(), 1, [Renderer.texture_target(b_tex)])
    r.set_attribute(p2, 0, scenes.UNIT_POSITION)
    r.set_attribute(p2, 1, scenes.QUAD_UV)
    r.set_instanced(p2, (0, 0), np.zeros((1, 2), dtype=np.float32))
    clear = Renderer.clear_color(0., 0., 0., 1.)

    def frame():
        r.render(h1.pipeline, h1.clear, None, h1.count)
        r.render(p2, clear, None, (1, 4))
        r.finish_frame()

    frame(); r.synchronize()
    r.set_timing(True); r.drain_timings()
    import time
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    r.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    t = r.drain_timings()
    n_frag1 = args.quads * args.side * args.side
    out = {"workload": f"two-pass {S}x{S}: {args.quads} quads {args.side}x{args.side} px into A, then one quad sampling A into B",
           "ms_per_frame": wall * 1e3, "tile_ms_both_passes": t["tile_ms"] / args.steps, "binning_ms_both_passes": t["bin_ms"] / args.steps, "setup_ms_both_passes": t["setup_ms"] / args.steps,
           "fragments": n_frag1 + S * S, "fragments_per_s": (n_frag1 + S * S) / wall}
    print(json.dumps(out), flush=True)
    h1.close(); r.close()


if __name__ == "__main__":
    main()
