#!/usr/bin/env python3
"""Binning / set-up / tile kernel times of ONE quad that covers the whole target (a background image, the second pass of
a two-pass render): the cluster touches every tile.  usage: python3 tools/bench_fullscreen.py [--size 3840x2160]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="3840x2160")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
scene = scenes.random_quads_scene(1, 16, W, H, seed=1)
scene["position"] = scenes.UNIT_POSITION
scene["instances"] = np.zeros((1, 2), dtype=np.float32)
hs = scenes.HipScene(scene)
r = hs.renderer
for _ in range(3):
    hs.draw()
r.set_timing(True); r.drain_timings()
for _ in range(10):
    hs.draw()
t = r.drain_timings()
print(json.dumps({"size": args.size, "pairs": int(r.last_pass_stats()["bin_entries"]), "bin_ms": t["bin_ms"] / t["passes"],
                  "setup_ms": t["setup_ms"] / t["passes"], "tile_ms": t["tile_ms"] / t["passes"]}))
hs.close()
