#!/usr/bin/env python3
"""Randomised parity sweep: HIP renderer against the CPU oracle, bit for bit, over scenes with random sizes, quad
shapes (sub-pixel, tiny, tile-spanning, partly or wholly off-screen, degenerate, non-rectangular, crossing the clip
planes), topologies, blend
modes, filters, viewports, clip switch, target formats, sample counts and LoadOp::Load.  Stops at the first mismatch
and prints the seed that reproduces it.

usage (on a GPU box, from the repo root): python3 tests/fuzz_parity.py [--iterations 300] [--seed 0] [--max-seconds 600]
(tests/test_gpu_fuzz.py runs a fixed set of seeds of the same generator under pytest)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from renderer_amd import RendererError

f32 = np.float32


def random_scene(rng):
    import scenes
    W = int(rng.integers(1, 700)); H = int(rng.integers(1, 400))
    shader = rng.choice(["hello", "hello", "sized", "tilted", "space"])
    n = int(rng.choice([0, 1, 2, 7, 64, 65, 300, 2000, 6000]))
    blend = str(rng.choice(["premultiplied", "additive"]))
    flt = str(rng.choice(["linear", "nearest"]))
    if shader == "tilted":
        scene = scenes.tilted_quads_scene(max(n, 1), W, H, seed=int(rng.integers(1 << 30)), blend=blend, filter=flt)
        scene["instances"][:, 2] *= f32(rng.choice([0.2, 1.0, 3.0]))
        scene["instances"][:, 3] *= f32(rng.choice([1.0, 1.0, 2.5]))      # |tilt| > 1: corners behind the eye (clipped)
    elif shader == "space":
        # quads in space: triangles crossing the near / far planes, the eye plane and the guard band
        scene = scenes.space_quads_scene(max(min(n, 600), 1), W, H, seed=int(rng.integers(1 << 30)), blend=blend, filter=flt,
                                         near=float(rng.choice([0.05, 0.5, 1.5])), far=float(rng.choice([3.0, 6.0, 50.0])),
                                         tilt_deg=float(rng.uniform(0.0, 89.0)), half_range=(0.02, float(rng.choice([0.3, 1.5, 6.0]))),
                                         topology=str(rng.choice(["strip", "list"])), spread=float(rng.choice([1.0, 3.0, 30.0])))
    elif shader == "sized":
        lo = int(rng.integers(1, 6)); hi = lo + int(rng.integers(0, 40))
        scene = scenes.tiny_quads_scene(max(n, 1), W, H, seed=int(rng.integers(1 << 30)), lo=lo, hi=hi)
        scene["blend"] = blend; scene["filter"] = flt
    else:
        side = float(rng.choice([0.3, 1.0, 2.5, 8.0, 16.0, 33.0, 70.0, 300.0]))
        hx, hy = f32(side / W), f32(side / H)                       # half extents in NDC
        scene = dict(shader="hello", width=W, height=H, target="texture",
                     position=np.array([[-hx, -hy], [-hx, hy], [hx, -hy], [hx, hy]], dtype=f32), uv=scenes.QUAD_UV,
                     instances=None, texture=scenes.load_letter_f(), texture_format="rgba_u8", filter=flt, blend=blend,
                     topology="strip", clear=(0., 0., 0., 1.), viewport=None, record=False)
        # positions anywhere, also outside the target; a few exactly on half-pixel lattices (ties)
        inst = rng.uniform(-1.3, 1.3, (max(n, 1), 2)).astype(f32)
        lattice = rng.random(max(n, 1)) < 0.2
        inst[lattice] = (np.rint(inst[lattice] * f32(W)) / f32(max(W, 1))).astype(f32)
        scene["instances"] = inst
        kind = rng.choice(["rect", "skew", "degenerate", "list"])
        pos = scene["position"].copy()
        if kind == "skew":
            pos = (pos + rng.uniform(-0.5, 0.5, pos.shape).astype(f32) * np.abs(pos).max()).astype(f32)   # bow-ties and folds too
        elif kind == "degenerate":
            pos[3] = pos[2]
        if kind == "list":
            order = [0, 1, 2, 1, 3, 2] if rng.random() < 0.5 else [0, 1, 2, 2, 1, 3, 0, 3, 1][: int(rng.choice([6, 9]))]
            pos = pos[order]; scene["uv"] = scenes.QUAD_UV[order]; scene["topology"] = "list"
        scene["position"] = pos
    if n == 0:
        scene["instances"] = scene["instances"][:0]
    scene["target"] = str(rng.choice(["texture", "screen"]))
    if rng.random() < 0.4:
        scene["viewport"] = (float(rng.choice([1.0, 16.0, 4.0, 3.3])), float(rng.choice([1.0, 9.0, 3.0, 7.1])))
    samples = int(rng.choice([1, 1, 1, 2, 4, 8]))
    if samples > 1:
        scene["samples"] = samples
    if scene["target"] == "texture":
        scene["target_format"] = str(rng.choice(["rgba_u8", "rgba_u8", "bgra_u8", "r_u8", "rgba_f16", "rgba_f32"]))
    if rng.random() < 0.3 and samples == 1:
        scene["record"] = True
        scene["record_clear"] = (0.2, 0.4, 0.6, 1.0) if rng.random() < 0.5 else scene["clear"]
    load = rng.random() < 0.25 and samples == 1
    clip = bool(rng.random() < 0.7)
    return scene, load, clip


def describe(scene, load, clip):
    keys = ("shader", "width", "height", "target", "target_format", "samples", "blend", "filter", "topology", "viewport", "record")
    d = {k: scene.get(k) for k in keys}
    d.update(instances=len(scene["instances"]), vertices=len(scene["position"]), load=load, clip_to_viewport=clip)
    return d


def run_one(seed):
    import scenes
    rng = np.random.default_rng(seed)
    scene, load, clip = random_scene(rng)
    initial = None
    if load:
        bpt = 4 if scene["target"] == "screen" else scenes.TARGET_FORMATS[scene.get("target_format", "rgba_u8")][1]
        shape = (scene["height"], scene["width"], bpt)
        if bpt == 8:
            initial = rng.uniform(-4, 4, (shape[0], shape[1], 4)).astype(np.float16).view(np.uint8).reshape(shape)
        elif bpt == 16:
            initial = rng.uniform(-4, 4, (shape[0], shape[1], 4)).astype(np.float32).view(np.uint8).reshape(shape)
        else:
            initial = rng.integers(0, 256, shape, dtype=np.uint8)
        scene["clear"] = None
    orc = scenes.draw_oracle(scene, clip_to_viewport=clip, initial=initial)
    # how the device cuts the last tiles of a pass into bands of rows (read when the device is made): never, one kind,
    # mixed; with the GPU taken to hold few workgroups, so that these small targets see whole tiles, halves and quarters
    os.environ["RH_TAIL_SPLIT"] = str(int(rng.integers(0, 4)))
    os.environ["RH_EARLY_BINNING"] = str(int(rng.choice([0, 2, 2])))
    os.environ["RH_WAVE_PRIORITY"] = str(int(rng.integers(0, 2)))
    os.environ["RH_TILE_ROOM"] = str(int(rng.choice([0, 16, 50, 120, 300])))
    try:
        hip = scenes.draw_hip(scene, clip_to_viewport=clip, initial=initial, twice=seed % 2 == 0)
    except RendererError as e:
        # recording through a viewport thinner than one pixel: the capture extent underflows, as in the reference
        if "subtract with overflow" in str(e) and scene.get("record") and scene["viewport"] is not None:
            return None, describe(scene, load, clip)
        raise
    for key in ("color", "recording", "msaa"):
        if key in orc and not np.array_equal(orc[key], hip[key]):
            diff = np.argwhere((orc[key] != hip[key]).reshape(orc[key].shape[0], orc[key].shape[1], -1).any(axis=2))
            return f"{key}: {len(diff)} pixels differ, first at (x={diff[0][1]}, y={diff[0][0]})", describe(scene, load, clip)
    return None, describe(scene, load, clip)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-seconds", type=float, default=600.0)
    args = ap.parse_args()
    import torch  # noqa: F401  (before the renderer's library: one HIP runtime per process)
    from oracle import loader
    loader.build()
    t0 = time.time()
    done = 0
    for i in range(args.iterations):
        seed = args.seed + i
        err, what = run_one(seed)
        done += 1
        if err:
            print(f"MISMATCH seed {seed}: {err}\n  {what}", flush=True)
            return 1
        if i % 20 == 0:
            print(f"[{i}] seed {seed} ok ({time.time() - t0:.0f} s) {what}", flush=True)
        if time.time() - t0 > args.max_seconds:
            break
    print(f"{done} random scenes identical to the oracle (seeds {args.seed}..{args.seed + done - 1}, {time.time() - t0:.0f} s)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
