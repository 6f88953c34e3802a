#!/usr/bin/env python3
"""PNG encode time of one recorded 4K frame: on the GPU (rh_png_encode, only the PNG bytes are read back) against the
host path (read the frame back, zlib level 1 as renderer_amd.PngEncoder.encode_to_bytes does).  Two frames: the quads
example (mostly flat) and the 100k-quad capture benchmark scene (noise-like).  Prints one JSON line per frame kind."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import scenes
    from renderer_amd import PngEncoder, Renderer
    W, H, reps = 3840, 2160, 20
    for kind in ("quads example", "100k quads 32x32"):
        scene = scenes.example_scene(30, width=W, height=H, record=True) if kind == "quads example" else \
            dict(scenes.random_quads_scene(100_000, 32, W, H, seed=4, target="screen"), record=True)
        r = Renderer(W, H)
        frames = []
        hs = scenes.HipScene(dict(scene, record=False), renderer=r)
        r.start_recording([hs.pipeline], Renderer.clear_color(0., 0., 0., 1.), 4000., frames.append)
        r.flush()
        hs.draw(); hs.draw()
        r.synchronize(); r.recorder.process_mapped_buffers()
        frame = frames[0]
        status, png = PngEncoder.encode_on_device(r, frame)        # warm-up (allocations)
        t0 = time.perf_counter()
        for _ in range(reps):
            status, png = PngEncoder.encode_on_device(r, frame)
        gpu_ms = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        status2, png2 = PngEncoder.encode_to_bytes(frame)
        host_ms = (time.perf_counter() - t0) * 1e3
        from PIL import Image
        import io
        same = np.array_equal(np.array(Image.open(io.BytesIO(png))), np.array(Image.open(io.BytesIO(png2))))
        print(json.dumps({"frame": kind, "size": [frame.width, frame.height], "gpu_ms_per_frame": gpu_ms, "gpu_png_bytes": len(png),
                          "host_zlib1_ms_per_frame": host_ms, "host_png_bytes": len(png2), "decoded_equal": bool(same)}), flush=True)
        r.stop_recording([hs.pipeline])
        for f in frames:
            f.drop()
        hs.close(); r.close()


if __name__ == "__main__":
    main()
