#!/usr/bin/env python3
"""Headline benchmark: instanced quads/s at 4K with alpha blending (BASELINE.json configs[1]).

A "step" is one frame: one render pass of 1,000,000 instanced 16x16 px textured quads (pre-multiplied alpha blend,
linear filtering) into a 3840x2160 RGBA8 texture target, cleared each frame, through the C ABI of
librenderer_hip.so.  Inputs (instance buffer, attributes, texture) are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards by frame -- every rank renders its own
frames with no data-path collective -- so the job value is the sum over ranks ("weak" scaling).

After the quads/s leg the same process measures the second half of the metric, recorded frames/s: frames of
BASELINE.json configs[4]'s scene (100,000 animated 32x32 quads at 4K, recording attachment) are rendered, captured on
the device and LZ4F-compressed there into stream files; the result is the "capture" object of the same JSON line.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak


def pmc_capture_traffic_bytes():
    """HBM bytes per rh_lz4f_compress launch (a round of 1024 streams, one 4K frame each) from the committed counter
    passes of the capture harness (tools/prof_capture.sh), corrected like pmc_traffic_bytes; None without a profile."""
    import glob
    import re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "capture_v*_pmc_avg_per_launch.json")),
                   key=lambda q: int(re.search(r"_v(\d+)_", os.path.basename(q)).group(1)))
    for path in reversed(paths):
        try:
            k = json.load(open(path))["rh_lz4f_compress"]
            return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0, int(k["streams"])
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_traffic_bytes(tag="c1"):
    """HBM bytes per rh_tile_1 launch from the committed counter passes of this same command (tools/prof.sh: one
    rocprofv3 --pmc run for FETCH_SIZE, one for WRITE_SIZE; both counters are in KiB).  FETCH_SIZE is doubled, the
    gfx950 correction for its 64-B tally of 128-B read requests; counters cannot be collected from inside this
    process, so the figure is the profiled one and is None when no profile of the default workload is present."""
    import glob
    import re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_%s_v*_pmc_avg_per_launch.json" % tag)),
                   key=lambda q: int(re.search(r"_v(\d+)_", os.path.basename(q)).group(1)))
    for path in reversed(paths):               # the newest profile of this workload
        try:
            k = json.load(open(path))["rh_tile_1"]
            return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        except (OSError, KeyError, ValueError):
            continue
    return None


def capture_plan(n_ranks, dist, local_rank):
    """(frames, streams, untimed warm-up rounds, directory) per GPU for the recorded-frames/s leg.  The frames of the leg's
    scene hardly compress (about 28 MB each) and a GPU fills its stream files at more than 10 GB/s, which no disk of the
    box takes: the files go to /dev/shm (memory) when it exists, else to /tmp.  Every rank also pins up to two host
    buffers of one round's bytes.  The largest plan is taken that leaves three tenths of the available memory (and of the
    free disk space, if the files go there) untouched with all ranks of the host running it; the ranks settle on the most
    careful choice.  (1024 streams: four wavefronts per CU, one per SIMD.)"""
    import shutil
    # (four timed rounds of 1024: a round is compressed while the next one is rendered, so the first and the last round
    # of a run are not overlapped with anything, and a run of two rounds is mostly fill and drain)
    plans = [(4096, 1024, 1), (2048, 1024, 1), (1024, 512, 1), (512, 256, 2), (256, 256, 1), (128, 128, 1)]
    frame_bytes = 30e6
    in_memory = os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)
    root = "/dev/shm" if in_memory else "/tmp"
    try:
        disk = shutil.disk_usage(root).free
    except OSError:
        disk = 0
    try:
        import psutil
        ram = psutil.virtual_memory().available
    except Exception:
        ram = 64e9
    choice = len(plans) - 1
    for i, (frames, streams, rounds) in enumerate(plans):
        files = n_ranks * max(frames, rounds * streams) * frame_bytes             # warm-up bytes are dropped before the timed frames
        pinned = n_ranks * 2 * streams * frame_bytes * 1.25
        if files <= 0.7 * disk and pinned + (files if in_memory else 0) <= 0.7 * ram:
            choice = i
            break
    if dist is not None:
        import torch
        t = torch.tensor([choice], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        choice = int(t.item())
    return plans[choice] + (root,)


def launch_ranks(n):
    """`bench.py --gpus N` started as one plain process: start the N ranks here, as fresh child processes under
    torch.distributed.run (one rank per GPU, RCCL between them), before this process has made any GPU call, relay
    rank 0's one JSON line and leave with the launcher's exit status.  The value of a run is never scaled by the
    argument: it is what the ranks that really ran measured."""
    import socket
    import subprocess
    rehearsal = os.environ.get("RH_REHEARSE_ON_ONE_GPU") == "1"
    if not rehearsal:
        import torch                      # (counting devices does not initialise the GPU)
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but this host shows {have} GPU(s); set RH_REHEARSE_ON_ONE_GPU=1 to rehearse "
                             f"the {n}-rank control flow on one GPU (gloo, not a measurement of scaling)\n")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode        # the children inherit stdout: rank 0 writes the line there itself


def run_c3(args, dist, rank, world, local_rank, backend, devices, json_fd):
    """BASELINE.json configs[3]: render-to-texture two-pass at 16384 x 16384 with the target split into strips of tile
    rows over the ranks (sort-first).  Pass 1 draws 4 M quads of 64x64 px (pre-multiplied alpha, linear filter) into
    texture A, every rank its strip; the strips of A are all-gathered in place (RCCL; pass 2 samples A anywhere); pass 2
    draws one quad covering B that samples A, every rank its strip; the strips of B are all-gathered too.  The work of a
    frame is fixed, so more ranks make the frame shorter: "strong" scaling.  One rank: the same two passes, no exchange."""
    import ctypes as C
    import numpy as np
    import scenes
    from renderer_amd import Renderer, _ffi, partition
    S, quads, side = args.width, args.quads, args.side
    device = f"cuda:{local_rank}" if dist is not None else "cpu"
    s1 = scenes.random_quads_scene(quads, side, S, S, seed=3)
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
    if world > 1:
        r.set_strip(row0, row1)
    exchange_s = [0.0]

    def step():
        r.render(h1.pipeline, h1.clear, None, h1.count)          # pass 1, this rank's strip of A
        r.finish_frame()
        if world > 1:
            r.synchronize()                                           # (so that the time below is the exchange alone)
            t = time.perf_counter()
            partition.exchange_strips(r, a_tex, rank, world, device)  # every rank now holds all of A
            exchange_s[0] += time.perf_counter() - t
        r.render(p2, clear, None, (1, 4))                             # pass 2, this rank's strip of B
        r.finish_frame()
        if world > 1:
            r.synchronize()
            t = time.perf_counter()
            partition.exchange_strips(r, b_tex, rank, world, device)
            exchange_s[0] += time.perf_counter() - t

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
        r.synchronize()

    for _ in range(args.warmup):
        step()
    r.synchronize()
    r.set_timing(True); r.drain_timings()
    exchange_s[0] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    r.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = r.drain_timings()
    if dist is not None:
        import torch
        v = torch.tensor([elapsed, exchange_s[0]], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        elapsed, exchange_s[0] = float(v[0]), float(v[1])
    r.set_timing(False)
    r.set_strip(0, 0)
    if rank == 0:
        n_frag = quads * side * side + S * S
        # algorithmic bytes of a frame: instance records, a read-modify-write of the target texel per fragment of either pass,
        # the distinct texels sampled (the quads' texture in pass 1, all of A in pass 2)
        b_alg = quads * 8 + 8 + n_frag * 2 * 4 + s1["texture"].nbytes + S * S * 4
        share = (row1 - row0) / S                      # this rank's part of every pass
        tile_ms = t["tile_ms"] / args.steps            # both passes of a frame, this rank's strips
        achieved = b_alg * share / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        strip_bytes = 2 * S * S * 4 * (world - 1) // max(1, world)     # received per rank and frame: the others' strips of A and of B
        out = {"metric": "instanced quads/s at 16K two-pass render-to-texture, sort-first strips", "value": quads * args.steps / elapsed, "unit": "quads/s",
               "n_gpus": world, "ranks": world, "backend": backend, "devices": devices, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32->unorm8", "data": "synthetic",
               "config": {"workload": f"two-pass {S}x{S} RGBA8 render-to-texture: {quads} quads {side}x{side} px (pre-multiplied alpha, linear filter) into A, "
                                      f"then one quad sampling A (linear) into B (BASELINE.json configs[3])",
                          "parallelism": f"sort-first strips of tile rows x{world}, in-place all-gather of A and of B" if world > 1 else "single GPU",
                          "strip_rows_rank0": [row0, row1]},
               "exchange": {"ms_per_step": exchange_s[0] / args.steps * 1e3, "bytes_received_per_rank_per_step": strip_bytes,
                            "note": "host wall time of the two in-place all-gathers of a frame (the slowest rank's), after the rank's own strip is done"},
               "roofline": {"bound": "hbm", "kernel": "rh_tile_1 (both passes of a frame)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_step": b_alg, "share_of_rank0": share,
                            "kernel_ms": tile_ms, "binning_ms": t["bin_ms"] / args.steps, "setup_ms": t["setup_ms"] / args.steps,
                            "fragments_per_s": n_frag * share / (tile_ms * 1e-3) if tile_ms > 0 else 0.0,
                            "limited_by": "VALU instruction issue (LDS-resident tile)"}}
        if world == 1 and not args.no_cpu_baseline:
            # CPU restatement on a bounded sample of pass 1: the same quads at the same density on a 2048 x 2048 target
            cores = min(16, os.cpu_count() or 1)
            k = max(1, (S // 2048) ** 2)
            sample = scenes.random_quads_scene(max(1, quads // k), side, min(S, 2048), min(S, 2048), seed=3)
            t1 = time.perf_counter()
            scenes.draw_oracle(sample, threads=cores)
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": len(sample["instances"]) / cpu_s, "unit": "quads/s", "cores": cores, "kind": "port",
                                   "sample": f"pass 1 at the same density on a {min(S, 2048)}x{min(S, 2048)} target ({len(sample['instances'])} quads of "
                                             f"{side}x{side} px) on {cores} threads, {cpu_s:.2f} s wall; CPU restatement, not lavapipe"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    h1.close(); r.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--quads", type=int, default=0, help="default: 1,000,000 (c1), 10,000,000 (c2)")
    ap.add_argument("--side", type=int, default=16, help="quad edge in pixels (headline: 16)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--workload", choices=["c1", "c2", "c3"], default="c1",
                    help="c1 (headline): --quads quads of --side px, alpha blend; c2: --quads quads of 2..8 px, additive; "
                         "c3: two-pass render-to-texture at 16384x16384 (4 M quads of 64 px), the target split into strips over --gpus ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone", action="store_true", help="skip the untimed leg that runs the same frames with every launch on one stream (profiling runs: the kernel statistics then hold the timed region only)")
    ap.add_argument("--texture-size", type=int, default=0,
                    help="experiment: sample an NxN subsampling of the texture instead of the 256x256 original")
    ap.add_argument("--filter", choices=["linear", "nearest"], default="linear", help="experiment: sampler filter (headline: linear)")
    ap.add_argument("--capture-frames", type=int, default=-1,
                    help="frames of the recorded-frames/s leg (BASELINE.json configs[4] shape, per GPU); 0 skips it; "
                         "default: 4096 frames into 1024 streams, or less when the ranks of this host would not find "
                         "room for their stream files and their pinned buffers in memory (see capture_plan)")
    ap.add_argument("--no-reference-budget", action="store_true",
                    help="skip the short run of the recorded-frames/s leg under the reference example's 500 MB capture budget")
    ap.add_argument("--capture-streams", type=int, default=-1,
                    help="stream files per GPU; default 1024 (one wavefront per stream: four per compute unit; a round of up "
                         "to 1024 streams takes about as long as a round of one)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} rank(s); "
                         "the two must agree\n")
        sys.exit(2)
    # Libraries underneath (RCCL prints a version banner at start-up) write to stdout: keep the real stdout for the one
    # JSON line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("RH_BENCH_WATCHDOG"):
        # debugging aid: after that many seconds every thread's Python stack goes to stderr and the process exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RH_BENCH_WATCHDOG"]), exit=True)
    if args.workload == "c3":
        # (the configuration's own shape unless the caller gives another)
        if args.quads <= 0:
            args.quads = 4_000_000
        if (args.width, args.height, args.side) == (3840, 2160, 16):
            args.width = args.height = 16384
            args.side = 64
        args.height = args.width
    if args.quads <= 0:
        args.quads = 1_000_000 if args.workload == "c1" else 10_000_000

    capture_auto = args.capture_frames < 0 and args.capture_streams < 0      # sized below, once the ranks can agree
    capture_warm_rounds = 1
    default_workload = (args.workload, args.quads, args.side, args.width, args.height, args.texture_size, args.filter) == ("c1", 1_000_000, 16, 3840, 2160, 0, "linear")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = world                     # the ranks that run, never the argument (checked against it above)
    backend, devices = "none", [local_rank]

    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        # launched by torch.distributed.run: one rank per GPU.  torch is imported before the renderer's library on
        # purpose: it bundles a HIP runtime of its own and the two must share one (see renderer_amd/partition.py).
        import torch
        import torch.distributed as dist
        if os.environ.get("RH_REHEARSE_ON_ONE_GPU") == "1":
            # every rank on GPU 0 with gloo between them: a rehearsal of the multi-rank control flow on a one-GPU box
            # (one stream per process: processes that share a GPU take turns on it, and a pass whose binning runs on
            # a second queue then waits through the other process's turns -- seconds per pass)
            os.environ.setdefault("RH_EARLY_BINNING", "0")
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} rank(s)\n")
            sys.exit(2)
        n_gpus = dist.get_world_size()
        backend = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (rehearsal: every rank on GPU 0)")
        # the device ordinal every rank renders on, as the ranks report them
        t = torch.zeros(n_gpus, dtype=torch.int64, device=f"cuda:{local_rank}")
        t[rank] = local_rank
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        devices = [int(v) for v in t.tolist()]

    import numpy as np
    import scenes
    from renderer_amd import Renderer

    if args.workload == "c3":
        run_c3(args, dist, rank, world, local_rank, backend, devices, json_fd)
        if dist is not None:
            dist.destroy_process_group()
        return

    # every rank renders a different frame of the same distribution
    if args.workload == "c2":
        scene = scenes.tiny_quads_scene(args.quads, args.width, args.height, seed=2 + rank)
    else:
        scene = scenes.random_quads_scene(args.quads, args.side, args.width, args.height, seed=1 + rank, filter=args.filter)
        if args.texture_size:
            step = max(1, 256 // args.texture_size)
            scene["texture"] = np.ascontiguousarray(scene["texture"][::step, ::step][:args.texture_size, :args.texture_size])
    renderer = Renderer(args.width, args.height, hip_ordinal=local_rank)
    hs = scenes.HipScene(scene, renderer=renderer)

    def step():
        renderer.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        renderer.finish_frame()

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
        renderer.synchronize()

    for _ in range(args.warmup):
        step()
    renderer.synchronize()
    # (marks around the tile kernel only, the kernel the roofline prices: every mark is a bubble of some 7 us in the
    # stream, and five per frame cost the frame 3 %; what binning and set-up take comes from an untimed leg below)
    renderer.set_timing(2)
    renderer.drain_timings()
    early_before = renderer.last_pass_stats()["early"]

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    renderer.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    timings = renderer.drain_timings()
    renderer.set_timing(False)
    stats = renderer.last_pass_stats()

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the frame the timed steps drew, kept for the comparison with the CPU oracle's frame of the same scene (outside
    # the timed region)
    hip_frame = hs.read_color() if (rank == 0 and n_gpus == 1 and not args.no_cpu_baseline) else None

    # Outside the timed region: the same frames with every launch on the one stream, so that the tile kernel's duration
    # is also known with nothing beside it (in the timed region the binning of the next frame shares the GPU with it).
    alone = None
    # (whether the timed passes were binned beside the tile kernel of the pass before: the library does that by itself
    # only for passes with millions of pairs, RH_EARLY_BINNING=2 asks for it always, 0 never)
    overlapped = stats["early"] - early_before > 0
    early_mode = int(os.environ.get("RH_EARLY_BINNING", "1"))
    phases = None
    if rank == 0 and n_gpus == 1 and not args.no_alone:
        if overlapped:
            renderer.set_early_binning(False)
        for _ in range(max(2, args.warmup)):
            step()
        renderer.synchronize()
        renderer.set_timing(True)
        renderer.drain_timings()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        renderer.synchronize()
        alone_s = time.perf_counter() - t1
        alone_t = renderer.drain_timings()
        renderer.set_timing(False)
        renderer.set_early_binning(early_mode)
        np_ = max(1, alone_t["passes"])
        phases = {"ms_per_step": alone_s / args.steps * 1e3, "kernel_ms": alone_t["tile_ms"] / np_,
                  "binning_ms": alone_t["bin_ms"] / np_, "setup_ms": alone_t["setup_ms"] / np_}
        if overlapped:
            alone = phases

    # release the quads scene before the recording leg
    hs.close()
    renderer.close()

    # Second half of the metric: recorded frames/s.  Every rank renders, records and GPU-compresses its own frames into
    # its own stream files (frame-parallel, no data-path collective); the job value is the sum over ranks.
    capture = None
    capture_root = "/tmp"
    if capture_auto:
        args.capture_frames, args.capture_streams, capture_warm_rounds, capture_root = capture_plan(n_gpus, dist, local_rank)
    elif args.capture_streams < 0:
        args.capture_streams = 256
    elif args.capture_frames < 0:
        args.capture_frames = 2 * args.capture_streams
    if args.capture_frames > 0 and args.workload == "c1":
        import bench_capture
        cap_s, cap_info, cap_error = float("inf"), None, None
        try:
            cap_s, cap_info, cap_last = bench_capture.run_capture(
                frames=args.capture_frames, streams=args.capture_streams, out=f"{capture_root}/rh_bench_capture_{os.getpid()}_{rank}",
                hip_ordinal=local_rank, seed=4 + rank,
                # untimed rounds first: the compressor pins its host buffers in them
                warmup=capture_warm_rounds * args.capture_streams)
        except Exception as e:                   # e.g. no room for the stream files: report it, keep the quads/s line
            cap_error = f"rank {rank}: {type(e).__name__}: {e}"
        # a frame the recorder dropped (no room in the capture budget) is not a recorded frame
        recorded_frames = (args.capture_frames - cap_info["frames_dropped"]) if cap_info is not None else 0
        if dist is not None:
            # ranks start the leg together (they all left the barrier of the first leg) and the slowest one counts
            import torch
            t = torch.tensor([cap_s], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cap_s = float(t.item())
            t = torch.tensor([recorded_frames], dtype=torch.int64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            recorded_frames = int(t.item())
        if cap_info is None or cap_s == float("inf"):
            capture = {"metric": "recorded frames/s", "value": None, "error": cap_error or "failed on another rank"}
        else:
            capture = {"metric": "recorded frames/s", "value": recorded_frames / cap_s, "unit": "frames/s",
                       "frames_per_gpu": args.capture_frames, "streams_per_gpu": args.capture_streams,
                       "ms_per_frame_per_gpu": cap_s / args.capture_frames * 1e3, "workload": cap_info["workload"],
                       "raw_MB_per_frame": cap_info["raw_MB_per_frame"], "compressed_MB_per_frame": cap_info["compressed_MB_per_frame"],
                       "frames_recorded": recorded_frames, "frames_dropped": cap_info["frames_dropped"], "budget_MB": cap_info["budget_MB"],
                       "stream_files_in": capture_root,
                       # where this rank's host time went: per frame in the render loop, and in total in the compressor's hand-offs
                       "host_ms_per_frame": cap_info["host_ms_per_frame"], "compressor_host_seconds": cap_info["compressor_host_seconds"]}
            # rh_lz4f_compress against the HBM roofline: the bytes it must read and write (raw frames in, LZ4F streams
            # out) over the HIP-event time of its launches (this rank's)
            k = cap_info["kernel"]
            if k["kernel_ms"] > 0:
                achieved = (k["bytes_in"] + k["bytes_out"]) / (k["kernel_ms"] * 1e-3) / 1e9
                capture["roofline"] = {"bound": "hbm", "kernel": "rh_lz4f_compress", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": k["rounds"],
                                       "algorithmic_bytes_per_launch": (k["bytes_in"] + k["bytes_out"]) / max(1, k["rounds"]),
                                       "kernel_ms_per_launch": k["kernel_ms"] / max(1, k["rounds"]),
                                       "note": "latency-bound: one wavefront per stream walks LZ4's sequential match search; "
                                               "the launches share the GPU with the rendering of the next round's frames"}
                traffic, profiled_streams = pmc_capture_traffic_bytes()
                if traffic is not None and profiled_streams == args.capture_streams:
                    capture["roofline"]["traffic"] = traffic
                    capture["roofline"]["traffic_source"] = "committed rocprofv3 counter passes of tools/prof_capture.sh (a round of the same size)"
            if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
                _, multi = bench_capture.liblz4_baseline(np.ascontiguousarray(cap_last).tobytes(), cpu_frames=4)
                capture["cpu_baseline"] = multi
        # The companion figure: the same leg under the capture budget of the reference's example (500 MB, reference
        # examples/quads_main_thread.rs:95).  Sixteen 4K frames fit, so rounds are eight frames: eight streams compress
        # at a time, and a round takes as long as a round of a thousand.
        if capture is not None and capture.get("value") and not args.no_reference_budget:
            small_s, small_info = float("inf"), None
            small_frames = 32
            try:
                small_s, small_info, _ = bench_capture.run_capture(
                    frames=small_frames, streams=args.capture_streams, out=f"{capture_root}/rh_bench_capture_small_{os.getpid()}_{rank}",
                    hip_ordinal=local_rank, seed=4 + rank, warmup=8, budget_mb=500.0)
            except Exception as e:
                capture["at_reference_budget"] = {"value": None, "error": f"rank {rank}: {type(e).__name__}: {e}"}
            small_recorded = (small_frames - small_info["frames_dropped"]) if small_info is not None else 0
            if dist is not None:
                import torch
                t = torch.tensor([small_s], dtype=torch.float64, device=f"cuda:{local_rank}")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                small_s = float(t.item())
                t = torch.tensor([small_recorded], dtype=torch.int64, device=f"cuda:{local_rank}")
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                small_recorded = int(t.item())
            if small_info is not None and small_s != float("inf"):
                capture["at_reference_budget"] = {
                    "budget_MB": 500.0, "value": small_recorded / small_s, "unit": "frames/s", "frames_per_gpu": small_frames,
                    "frames_recorded": small_recorded, "frames_dropped": small_info["frames_dropped"],
                    "note": "16 frames of 4K fit the budget: rounds of 8 frames, 8 streams compressing at a time; a stream is "
                            "sequential, so the rate is 8 frames per round whatever the GPU's size"}

    parity_failed = False
    if rank == 0:
        total_quads = args.quads * args.steps * n_gpus          # every one of the n_gpus ranks drew `steps` frames of `quads`
        value = total_quads / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        # Algorithmic bytes of one tile-kernel launch (one frame): every instance record read once, every blended
        # fragment a read-modify-write of its 4-byte target texel, plus the distinct texels of the texture.
        if args.workload == "c2":
            side_px = np.rint(scene["instances"][:, 2] * args.width).astype(np.int64)
            n_frag = int((side_px * side_px).sum())
        else:
            n_frag = args.quads * args.side * args.side
        tex_bytes = scene["texture"].nbytes
        b_alg = args.quads * scene["instances"].shape[1] * 4 + n_frag * 2 * 4 * 1 + tex_bytes
        b_min = args.quads * scene["instances"].shape[1] * 4 + args.width * args.height * 4 + tex_bytes
        tile_ms = timings["tile_ms"] / max(1, timings["passes"])
        # (from the untimed leg with marks around every phase; beside a tile kernel -- `overlapped` -- they are longer)
        bin_ms = phases["binning_ms"] if phases else None
        setup_ms = phases["setup_ms"] if phases else None
        achieved = b_alg / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        c2_profiled = (args.workload == "c2" and args.quads == 10_000_000 and (args.width, args.height) == (3840, 2160))
        traffic = pmc_traffic_bytes("c1") if default_workload else (pmc_traffic_bytes("c2") if c2_profiled else None)
        out = {
            "metric": "instanced quads/s at 4K alpha-blend" if args.workload == "c1" else "instanced quads/s at 4K additive tiny quads", "value": value, "unit": "quads/s",
            "n_gpus": n_gpus, "ranks": n_gpus, "backend": backend, "devices": devices, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32->unorm8",
            "data": "synthetic",
            "config": {"workload": (f"{args.quads} instanced textured quads {args.side}x{args.side} px, pre-multiplied alpha blend, "
                                    f"linear filter, {args.width}x{args.height} RGBA8 texture target, clear each frame "
                                    "(BASELINE.json configs[1])") if args.workload == "c1" else
                                   (f"{args.quads} instanced textured quads of 2..8 px, additive blend, linear filter, "
                                    f"{args.width}x{args.height} RGBA8 texture target, clear each frame (BASELINE.json configs[2])"),
                       "parallelism": f"frame-parallel x{n_gpus}" if n_gpus > 1 else "single GPU",
                       "bin_entries": int(stats["bin_entries"]), "tile_px": [int(stats["tile_w"]), int(stats["tile_h"])]},
            "roofline": {"bound": "hbm", "kernel": "rh_tile_1", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "compulsory_bytes_per_launch": b_min,
                         "kernel_ms": tile_ms, "binning_ms": bin_ms, "setup_ms": setup_ms,
                         "phases_from": "an untimed leg of this run, every launch on one stream, HIP events around every phase of a pass (the timed region has them around the tile kernel only)",
                         "fragments_per_s": n_frag / (tile_ms * 1e-3) if tile_ms > 0 else 0.0,
                         # true: binning and set-up of frame n+1 ran on a stream of their own beside the tile kernel of
                         # frame n (passes with millions of pairs, or RH_EARLY_BINNING=2): binning_ms is then their wall
                         # time spread under that kernel, not time added to the frame, and kernel_ms is the tile kernel
                         # with that work sharing the GPU
                         "overlapped": overlapped,
                         # "hbm" is how the path is priced (bytes the reference's semantics move per fragment); the tile
                         # lives in LDS, so what the kernel actually runs into is on the CU, not in memory
                         "limited_by": "instruction issue and, by now as much, the chain of LDS and texture round trips a wave makes per 64-pixel pass (LDS-resident tile: HBM traffic is a tenth of the algorithmic bytes)"},
        }
        if alone is not None and alone["kernel_ms"] > 0:
            # the same kernel over the same frames with nothing beside it (untimed leg above): the kernel's own figure
            a = b_alg / (alone["kernel_ms"] * 1e-3) / 1e9
            out["roofline"]["alone"] = dict(alone, achieved=a, frac=a / HBM_PEAK_GBS,
                                            note="early binning off: every launch on one stream; not the timed region")
        if n_gpus == 1 and not args.no_cpu_baseline:
            # CPU restatement (the parity oracle) timed on the host cores, same scene, one full frame.
            cores = min(16, os.cpu_count() or 1)
            t1 = time.perf_counter()
            orc = scenes.draw_oracle(scene, threads=cores)
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": args.quads / cpu_s, "unit": "quads/s", "cores": cores, "kind": "port",
                                   "sample": f"one full frame of the same scene ({args.quads} quads, {orc['fragments']} fragments) "
                                             f"on {cores} threads, {cpu_s:.2f} s wall; CPU restatement, not lavapipe"}
            # parity of exactly what was timed: every byte of the HIP frame against the oracle's frame
            differ = int((hip_frame != orc["color"]).any(axis=2).sum())
            out["parity"] = {"pixels": int(hip_frame.shape[0] * hip_frame.shape[1]), "differ": differ,
                             "against": "CPU oracle, same scene, bit-exact"}
            parity_failed = differ != 0
        if capture is not None:
            out["capture"] = capture
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.destroy_process_group()
    if parity_failed:
        sys.stderr.write("bench.py: the timed frame differs from the CPU oracle's\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
