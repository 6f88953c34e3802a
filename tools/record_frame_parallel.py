#!/usr/bin/env python3
"""Frame-parallel offline recording with one ordered sink (BASELINE.json configs[4] with the ffmpeg pipe as output).

Launch one process per GPU (python -m torch.distributed.run --nproc-per-node N tools/record_frame_parallel.py ...) or a
single process.  Rank r renders the frames f with (f - 1) % N == r of the quads example's animation (the scene state is a
pure function of the frame number, every rank steps it redundantly), records them, and encodes each as a PNG on its GPU.
Per round of N frames the PNG byte strings are gathered to rank 0 (torch.distributed: RCCL between GPUs), which writes
them in frame order to the sink: the ffmpeg pipe when ffmpeg is installed (reference src/ffmpeg_pipe.rs:34-55 feeds
`-f image2pipe` the same concatenated PNGs), else a file holding the identical byte stream.

usage: record_frame_parallel.py --frames 120 --width 1920 --height 1080 --out /tmp/recording.pngstream [--mp4 out.mp4]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def record(n_frames, width, height, write, rank=0, world=1, local_rank=0, dist=None, device="cpu"):
    """Renders frames 1..n_frames split over `world` ranks; rank 0 calls write(frame_number, png_bytes) in frame order.
    Returns the number of frames this rank rendered."""
    import scenes
    from renderer_amd import PngEncoder, Renderer, partition
    scene = scenes.example_scene(1, width=width, height=height)
    r = Renderer(width, height, hip_ordinal=local_rank)
    hs = scenes.HipScene(scene, renderer=r)
    delivered = []
    r.start_recording([hs.pipeline], Renderer.clear_color(0., 0., 0., 1.), 2000., delivered.append, keep_on_device=True)
    r.flush()
    mine = set(partition.frames_for_rank(n_frames, rank, world))
    rendered = 0
    rounds = (n_frames + world - 1) // world
    for k in range(rounds):
        f = k * world + rank + 1                      # this rank's frame of the round (may be past the end)
        png = b""
        if f in mine:
            r.set_instanced(hs.pipeline, (0, 0), scenes.example_positions(f))
            r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
            r.finish_frame()
            r.recorder.process_mapped_buffers()
            frame = delivered.pop(0)
            status, png = PngEncoder.encode_on_device(r, frame)
            assert status == "ok"
            frame.drop()
            rendered += 1
        if dist is not None:
            parts = partition.gather_bytes(png, dst=0, device=device)
        else:
            parts = [png]
        if rank == 0:
            for src, data in enumerate(parts):
                if data:
                    write(k * world + src + 1, data)
    r.stop_recording([hs.pipeline])
    hs.close(); r.close()
    return rendered


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--out", default="/tmp/recording.pngstream")
    ap.add_argument("--mp4", default=None, help="encode with ffmpeg into this file instead (needs ffmpeg)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch  # noqa: F401  (before the renderer's library: one shared HIP runtime)
    dist, device = None, "cpu"
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
    sink = pipe = None
    if rank == 0:
        from renderer_amd import FfmpegPipe
        if args.mp4 and FfmpegPipe.available():
            pipe = FfmpegPipe(None, None, args.mp4, ["-c:v", "libx264", "-r", "60", "-pix_fmt", "yuv420p"])
        else:
            sink = open(args.out, "wb")

    def write(frame_number, png):
        if pipe is not None:
            pipe.write(None, png)
        else:
            sink.write(png)

    t0 = time.perf_counter()
    n = record(args.frames, args.width, args.height, write, rank, world, local_rank, dist, device)
    if dist is not None:
        dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
    if rank == 0:
        dt = time.perf_counter() - t0
        if sink is not None:
            sink.close()
        if pipe is not None:
            pipe.close()
        print(f"{args.frames} frames on {world} GPU(s) in {dt:.2f} s = {args.frames / dt:.1f} frames/s -> {args.mp4 if pipe else args.out}")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
