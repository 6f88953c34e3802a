"""Multisampled passes with and without the bands at the end of a pass: C1-density quads, several sample counts and target sizes.
Tile-kernel time by HIP events (RH_OPTION_TIMING), every launch on one stream.  Prints one JSON object."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (before the renderer's library: one HIP runtime per process)
import scenes
from renderer_amd import Renderer

rows = []
for (w, h), samples in (((640, 360), 4), ((1280, 720), 4), ((3840, 2160), 2), ((3840, 2160), 4)):
    quads = int(round(1_000_000 * w * h / (3840 * 2160)))
    scene = scenes.random_quads_scene(quads, 16, w, h, seed=1)
    scene["samples"] = samples
    row = {"target": [w, h], "samples": samples, "quads": quads}
    for mode in (0, 2):
        r = Renderer(w, h)
        r.set_early_binning(False)
        r.set_tail_split(mode)
        hs = scenes.HipScene(scene, renderer=r)
        for _ in range(3):
            hs.draw()
        r.synchronize(); r.set_timing(True); r.drain_timings()
        for _ in range(10):
            hs.draw()
        r.synchronize()
        t = r.drain_timings()
        row["tile_ms_whole_tiles" if mode == 0 else "tile_ms_bands"] = round(t["tile_ms"] / max(1, t["passes"]), 4)
        hs.close(); r.close()
    rows.append(row)
    print(row, file=sys.stderr, flush=True)
print(json.dumps({"workload": "C1 density, multisampled, RH_EARLY_BINNING=0", "rows": rows}))
