#!/usr/bin/env python3
"""Sort-first split of one very large target over the GPUs of a node (BASELINE.json configs[3]).

Launch one process per GPU (python -m torch.distributed.run --nproc-per-node N tools/render_strips.py ...) or a single
process.  The target's rows are dealt to the ranks in strips aligned to the tile height; every rank holds the whole
instance buffer and renders (and clears) only its strip, so per-pixel primitive order is untouched.  Pass 1 renders the
quads into texture A; the strips of A are all-gathered (RCCL) into every rank's copy of A because pass 2 samples A at
arbitrary coordinates; pass 2 renders each rank's strip of B; rank 0 gathers the strips of B.

usage: render_strips.py --size 16384 --quads 4000000 --side 64 [--out b.png]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def render_two_pass(size, quads, side, rank=0, world=1, local_rank=0, device="cpu", seed=3):
    """Returns (full image B on rank 0 as uint8 [size, size, 4] or None, seconds for the two passes and exchanges)."""
    import ctypes as C
    import numpy as np
    import scenes
    from renderer_amd import Renderer, _ffi, partition
    S = size
    s1 = scenes.random_quads_scene(quads, side, S, S, seed=seed)
    r = Renderer(64, 64, hip_ordinal=local_rank)
    h1 = scenes.HipScene(s1, renderer=r)                    # pass 1 -> texture A (S x S, RGBA8)
    a_tex = h1.target_texture
    a_view = type(a_tex).__new__(type(a_tex))               # A again, with a linear sampler, as pass 2's texture
    a_view.__dict__.update(a_tex.__dict__)
    smp = C.c_void_p()
    _ffi.check(_ffi.lib().rh_sampler_create(r.device, int(Renderer.linear_filtering()), C.byref(smp)))
    a_view.sampler = smp
    _ffi.lib().rh_texture_retain(a_view.handle)
    b_tex = r.texture(S, S, 1, Renderer.nearest_filtering(), Renderer.rgba_u8(), True, True, False)
    vert, frag = scenes.shader_code("hello")
    a_position, a_tex_coord, i_offset = r.attribute(0, 2), r.attribute(1, 2), r.instanced()
    program = r.program(vert, frag, [a_position, a_tex_coord], [i_offset], [], [(a_view, Renderer.visible_to_fragment_shader())])
    p2 = r.pipeline(program, Renderer.pre_multiplied_blend(), Renderer.triangle_strip_primitive(), 1, [Renderer.texture_target(b_tex)])
    r.set_attribute(p2, 0, scenes.UNIT_POSITION)
    r.set_attribute(p2, 1, scenes.QUAD_UV)
    r.set_instanced(p2, (0, 0), np.zeros((1, 2), dtype=np.float32))
    clear = Renderer.clear_color(0., 0., 0., 1.)
    row0, row1 = partition.strip_for_rank(S, rank, world)
    r.synchronize()
    t0 = time.perf_counter()
    r.set_strip(row0, row1)
    r.render(h1.pipeline, h1.clear, None, h1.count)          # pass 1, this rank's strip of A
    r.finish_frame()
    partition.exchange_strips(r, a_tex, rank, world, device)  # all ranks now hold all of A
    r.render(p2, clear, None, (1, 4))                         # pass 2, this rank's strip of B
    r.finish_frame()
    r.synchronize()
    partition.exchange_strips(r, b_tex, rank, world, device)  # the strips of B meet in place as well
    image = b_tex.read() if rank == 0 else None
    seconds = time.perf_counter() - t0
    r.set_strip(0, 0)
    h1.close(); r.close()
    return image, seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--quads", type=int, default=4_000_000)
    ap.add_argument("--side", type=int, default=64)
    ap.add_argument("--out", default=None, help="write B as a PNG (small sizes)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch  # noqa: F401  (before the renderer's library: one shared HIP runtime)
    device = "cpu"
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if os.environ.get("RH_REHEARSE_ON_ONE_GPU") == "1":
            # every rank on GPU 0, gloo between them: exercises the multi-process path where only one GPU is present
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        device = f"cuda:{local_rank}"
    image, seconds = render_two_pass(args.size, args.quads, args.side, rank, world, local_rank, device)
    if rank == 0:
        print(f"two-pass {args.size}x{args.size} on {world} GPU(s): {seconds * 1e3:.1f} ms (strips and the all-gathers of A and B included)")
        if args.out:
            from PIL import Image
            Image.fromarray(image, "RGBA").save(args.out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
