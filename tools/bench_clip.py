#!/usr/bin/env python3
"""Times a pass of quads in space (many of them clipped) at 4K: binning, set-up and tile kernel by HIP events, pairs per
pass.  usage: python3 tools/bench_clip.py [--quads 2000] [--half 1.5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quads", type=int, default=2000)
ap.add_argument("--half", type=float, default=1.5)
ap.add_argument("--frames", type=int, default=10)
args = ap.parse_args()
scene = scenes.space_quads_scene(args.quads, 3840, 2160, seed=7, half_range=(0.05, args.half))
hs = scenes.HipScene(scene)
r = hs.renderer
for _ in range(3):
    hs.draw()
r.set_timing(True); r.drain_timings()
for _ in range(args.frames):
    hs.draw()
t = r.drain_timings()
st = r.last_pass_stats()
print(json.dumps({"quads": args.quads, "pairs": int(st["bin_entries"]), "reruns": int(st["reruns"]),
                  "bin_ms": t["bin_ms"] / t["passes"], "setup_ms": t["setup_ms"] / t["passes"], "tile_ms": t["tile_ms"] / t["passes"]}))
hs.close()
