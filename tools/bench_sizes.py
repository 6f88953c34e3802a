"""C1-density scenes on targets of several sizes (quads scale with the area: 1 M at 3840x2160), with and without the
bands at the end of a pass: tile-kernel time by HIP events, every launch on one stream.  Prints one JSON object."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for w, h in [(1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (5120, 2880), (7680, 4320)]:
    quads = int(round(1_000_000 * w * h / (3840 * 2160)))
    row = {"target": [w, h], "tiles": ((w + 31) // 32) * ((h + 31) // 32), "quads": quads}
    for mode in (0, 2):
        env = dict(os.environ, RH_TAIL_SPLIT=str(mode), RH_EARLY_BINNING="0")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", str(w), "--height", str(h), "--quads", str(quads),
                            "--capture-frames", "0", "--no-cpu-baseline", "--no-alone", "--steps", "20", "--warmup", "3"],
                           capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-2000:]); sys.exit(1)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        row["tile_ms_whole_tiles" if mode == 0 else "tile_ms_bands"] = round(d["roofline"]["kernel_ms"], 4)
        row["frame_ms_whole_tiles" if mode == 0 else "frame_ms_bands"] = round(d["ms_per_step"], 4)
    print(row, file=sys.stderr, flush=True)
    out.append(row)
print(json.dumps({"workload": "C1 density (16x16 px quads, 1 M per 3840x2160) on several targets, RH_EARLY_BINNING=0", "rows": out}))
