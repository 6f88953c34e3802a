#!/usr/bin/env python3
"""Recorded frames/s (BASELINE.json configs[4] shape, one GPU's share): animated quads at 4K with the recording
attachment, every frame copied into a capture buffer that stays on the device, LZ4F-compressed on the GPU into
`n_streams` stream files; only compressed bytes cross PCIe.  Prints one JSON line with the end-to-end frames/s
(wall clock including file writes) and a phase breakdown, beside liblz4 level 0 on one host core on the same frames.

usage: python3 tools/bench_capture.py [--frames 128] [--streams 64] [--quads 100000] [--side 32] [--out /tmp/rh_cap]
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_capture(frames=128, streams=64, quads=100_000, side=32, width=3840, height=2160, out="/tmp/rh_cap", hip_ordinal=0,
                seed=4, before_timing=None, after_timing=None, warmup=None, budget_mb=None, keep_files=False):
    """Renders and records `frames` frames; returns (elapsed seconds, info dict, last recorded image).  `before_timing`
    and `after_timing` are called around the timed loop (multi-process runs put their barriers there).  `warmup` untimed
    frames come first, by default two rounds of the streams: the compressor pins one host buffer per round in flight
    (about a second per 8 GB, once) and recycles them afterwards."""
    import numpy as np
    import scenes
    from renderer_amd import Compressor, Renderer

    W, H = width, height
    scene = scenes.random_quads_scene(quads, side, W, H, seed=seed, target="screen")
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-0.01, 0.01, scene["instances"].shape).astype(np.float32)
    pos = scene["instances"].copy()
    tmp, outside = np.empty_like(pos), np.empty(pos.shape, dtype=bool)

    shutil.rmtree(out, ignore_errors=True)
    r = Renderer(W, H, hip_ordinal=hip_ordinal)
    if os.environ.get("RH_EXP_TIMING") == "1":                     # (diagnostic: event marks around every launch of a pass)
        r.set_timing(True)
    hs = scenes.HipScene(scene, renderer=r)
    # budget: two rounds of the streams' frames may be held on the device at once (33 MB each at 4K): one being
    # compressed, one being captured
    budget_mb = (2 * streams + 2) * W * H * 4 / (1024.0 * 1024.0) if budget_mb is None else budget_mb
    comp = Compressor(r, out, n_streams=streams, max_buffer_size_in_megabytes=budget_mb, wait_for_rounds=os.environ.get("RH_CAPTURE_TURNS", "0") == "1")
    t_compress = [0.0]

    held = []

    def sink(frame):
        t = time.perf_counter()
        if os.environ.get("RH_EXP_SINK") == "hold":          # experiment: the compressor exists but gets no frames
            held.append(frame)
            return
        comp.compress_to_disk(frame)
        t_compress[0] += time.perf_counter() - t

    r.start_recording([hs.pipeline], Renderer.clear_color(0., 0., 0., 1.), budget_mb, sink, keep_on_device=True)
    r.flush()

    phases = {"animate": 0.0, "upload": 0.0, "render": 0.0, "finish_frame": 0.0}      # host seconds per part of a frame

    def step():
        t_a = time.perf_counter()
        np.add(pos, vel, out=pos)                                  # the example's bounce rule, vectorised in f32, in place:
        np.abs(pos, out=tmp)                                       # the velocity of what has left [-0.9, 0.9] changes sign
        np.greater(tmp, np.float32(0.9), out=outside)
        np.multiply(outside, np.float32(-2.0), out=tmp)
        np.add(tmp, np.float32(1.0), out=tmp)
        np.multiply(vel, tmp, out=vel)
        t_b = time.perf_counter()
        if os.environ.get("RH_EXP_STATIC") != "1":                 # (experiment: a scene that does not move)
            r.set_instanced(hs.pipeline, (0, 0), pos)
        t_c = time.perf_counter()
        r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        t_d = time.perf_counter()
        r.finish_frame()
        t_e = time.perf_counter()
        phases["animate"] += t_b - t_a; phases["upload"] += t_c - t_b; phases["render"] += t_d - t_c; phases["finish_frame"] += t_e - t_d

    for _ in range(2 * streams if warmup is None else warmup):
        step()
    comp.flush()
    r.synchronize()
    for f in comp.files:                 # the warm-up's bytes are not kept (nothing here reads the files back)
        f.truncate(0)
        f.seek(0)
    t_compress[0] = 0.0
    for key in phases:
        phases[key] = 0.0
    raw0, packed0 = comp.raw_video_size, comp.compressed_size
    for key in comp.host_seconds:
        comp.host_seconds[key] = 0.0
    comp.drain_kernel_stats()

    if before_timing:
        before_timing()
    t0 = time.perf_counter()
    for _ in range(frames):
        step()
    comp.flush()
    r.synchronize()
    if after_timing:
        after_timing()
    elapsed = time.perf_counter() - t0
    kernel = comp.drain_kernel_stats()
    raw = comp.raw_video_size - raw0
    packed = comp.compressed_size - packed0
    last = hs.read_recording()
    r.stop_recording([hs.pipeline])
    comp.finish()
    hs.close()
    r.close()
    if not keep_files:
        shutil.rmtree(out, ignore_errors=True)
    info = {"frames": frames, "streams": streams, "frames_dropped": comp.frames_dropped, "budget_MB": budget_mb,
            "workload": f"{quads} animated quads {side}x{side} px, {W}x{H}, recording attachment, GPU LZ4F level 0 into "
                        f"{streams} stream files per GPU",
            "compress_ms_per_frame": t_compress[0] / frames * 1e3, "kernel": kernel, "compressor_host_seconds": dict(comp.host_seconds),
            "host_ms_per_frame": {k: v / frames * 1e3 for k, v in phases.items()},
            "raw_MB_per_frame": raw / frames / 1e6, "compressed_MB_per_frame": packed / frames / 1e6}
    return elapsed, info, last


def liblz4_baseline(img, cpu_frames=4, all_cores=True):
    """The system's liblz4 (what the reference links through lzzzz) with the reference's preferences (level 0,
    auto-flush; reference src/compressor.rs:138-144) on the image bytes `img`, repeated: one stream on one core, and one
    stream per core on the box's CPU share as the reference runs them (src/compressor.rs:24-26).  Returns the
    `cpu_baseline` objects (one core, all cores)."""
    import ctypes as C

    class Prefs(C.Structure):              # LZ4F_preferences_t of lz4frame.h 1.9.x
        _fields_ = [("blockSizeID", C.c_uint32), ("blockMode", C.c_uint32), ("contentChecksumFlag", C.c_uint32),
                    ("frameType", C.c_uint32), ("contentSize", C.c_uint64), ("dictID", C.c_uint32),
                    ("blockChecksumFlag", C.c_uint32), ("compressionLevel", C.c_int32), ("autoFlush", C.c_uint32),
                    ("favorDecSpeed", C.c_uint32), ("reserved", C.c_uint32 * 3)]

    L = C.CDLL("liblz4.so.1")
    L.LZ4F_createCompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]; L.LZ4F_createCompressionContext.restype = C.c_size_t
    L.LZ4F_compressBegin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]; L.LZ4F_compressBegin.restype = C.c_size_t
    L.LZ4F_compressBound.argtypes = [C.c_size_t, C.c_void_p]; L.LZ4F_compressBound.restype = C.c_size_t
    L.LZ4F_compressUpdate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.LZ4F_compressUpdate.restype = C.c_size_t
    L.LZ4F_isError.argtypes = [C.c_size_t]; L.LZ4F_isError.restype = C.c_uint

    def compress_on_one_core(n_frames, result, slot):
        """One LZ4F stream, the way one worker thread of the reference holds one (src/compressor.rs:64-69); only the
        library call on the image is timed.  ctypes releases the GIL during the call, so threads run in parallel."""
        ctx, prefs = C.c_void_p(), Prefs()
        prefs.autoFlush = 1
        assert not L.LZ4F_isError(L.LZ4F_createCompressionContext(C.byref(ctx), 100))
        head = (C.c_uint8 * 32)()
        assert not L.LZ4F_isError(L.LZ4F_compressBegin(ctx, head, 32, C.byref(prefs)))
        src = (C.c_uint8 * len(img)).from_buffer_copy(img)
        cap = L.LZ4F_compressBound(len(img), C.byref(prefs))
        dst = (C.c_uint8 * cap)()
        busy = 0.0
        for _ in range(n_frames):
            t1 = time.perf_counter()
            n = L.LZ4F_compressUpdate(ctx, dst, cap, src, len(img), None)
            busy += time.perf_counter() - t1
            assert not L.LZ4F_isError(n)
        result[slot] = busy

    one = [0.0]
    compress_on_one_core(cpu_frames, one, 0)
    single = {"value": cpu_frames / one[0], "unit": "frames/s", "cores": 1, "kind": "reference",
              "sample": f"system liblz4 1.9.3 (the library the reference links), level 0, {cpu_frames} frames of {len(img)} B on one core "
                        f"(compress only, no rendering), {one[0]:.2f} s"}
    if not all_cores:
        return single, None
    import threading
    cores = min(16, os.cpu_count() or 1)
    busy = [0.0] * cores
    threads = [threading.Thread(target=compress_on_one_core, args=(cpu_frames, busy, i)) for i in range(cores)]
    t1 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t1
    multi = {"value": cores * cpu_frames / wall, "unit": "frames/s", "cores": cores, "kind": "reference",
             "sample": f"system liblz4 1.9.3 (the library the reference links), level 0, {cores} threads with one LZ4F stream each, "
                       f"{cpu_frames} frames of {len(img)} B per thread (compress only, no rendering), {wall:.2f} s wall including context set-up"}
    return single, multi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--quads", type=int, default=100_000)
    ap.add_argument("--side", type=int, default=32)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--out", default="/tmp/rh_cap")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed frames before the timed ones (default: two rounds of the streams)")
    ap.add_argument("--keep-files", action="store_true", help="leave the stream files in --out")
    ap.add_argument("--budget-mb", type=float, default=-1.0, help="capture budget in MB (default: two rounds of the streams' frames)")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames compressed by liblz4 on one host core for the baseline")
    args = ap.parse_args()

    import numpy as np
    elapsed, info, last = run_capture(args.frames, args.streams, args.quads, args.side, args.width, args.height, args.out,
                                      warmup=None if args.warmup < 0 else args.warmup, keep_files=args.keep_files,
                                      budget_mb=None if args.budget_mb < 0 else args.budget_mb)
    out = {"metric": "recorded frames/s", "value": args.frames / elapsed, "unit": "frames/s", "n_gpus": 1,
           "frames": args.frames, "streams": args.streams, "config": {"workload": info["workload"]},
           "ms_per_frame": elapsed / args.frames * 1e3, "compress_ms_per_frame": info["compress_ms_per_frame"],
           "raw_MB_per_frame": info["raw_MB_per_frame"], "compressed_MB_per_frame": info["compressed_MB_per_frame"],
           "raw_GBps": info["raw_MB_per_frame"] * args.frames / elapsed / 1e3, "kernel": info["kernel"],
           "host_ms_per_frame": info["host_ms_per_frame"], "compressor_host_seconds": info["compressor_host_seconds"],
           "frames_dropped": info["frames_dropped"]}

    if args.cpu_frames > 0:
        single, multi = liblz4_baseline(np.ascontiguousarray(last).tobytes(), args.cpu_frames)
        out["cpu_baseline"] = single
        out["cpu_baseline_all_cores"] = multi
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
