"""CPU tests of the multi-GPU partitioning: pure-function properties, and the N > 1 exchange steps with world size 2
over gloo."""
import os
import socket

import numpy as np
import pytest

from renderer_amd import partition


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_frames_partition_is_exact_cover(world):
    n = 3600
    seen = []
    for r in range(world):
        frames = partition.frames_for_rank(n, r, world)
        assert all(partition.rank_of_frame(f, world) == r for f in frames)
        seen += frames
    assert sorted(seen) == list(range(1, n + 1))
    ids = [i for r in range(world) for i in partition.stream_ids_for_rank(r, 4)]
    assert ids == list(range(world * 4))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("height", [16384, 2160, 1080, 33, 7])
def test_strips_tile_aligned_exact_cover(world, height):
    prev = 0
    for r in range(world):
        r0, r1 = partition.strip_for_rank(height, r, world)
        assert r0 == prev and r0 <= r1 <= height
        assert (r0 % 32 == 0 or r0 == height) and (r1 % 32 == 0 or r1 == height)
        prev = r1
    assert prev == height
    if height == 16384 and world == 8:
        assert partition.strip_for_rank(height, 3, 8) == (3 * 2048, 4 * 2048)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # frame-parallel: every rank "encodes" its frames; rank 0 receives them and restores frame order
        n_frames = 9
        mine = partition.frames_for_rank(n_frames, rank, world)
        payload = b"".join(f.to_bytes(2, "big") + bytes([f]) * (f * 3) for f in mine)
        gathered = partition.gather_bytes(payload, dst=0)
        if rank == 0:
            frames = {}
            for blob in gathered:
                pos = 0
                while pos < len(blob):
                    f = int.from_bytes(blob[pos:pos + 2], "big"); pos += 2
                    frames[f] = blob[pos:pos + f * 3]; pos += f * 3
            assert sorted(frames) == list(range(1, n_frames + 1))
            assert all(frames[f] == bytes([f]) * (f * 3) for f in frames)
        # sort-first strips: the all-gathered image equals the reference image on every rank
        # (100 rows: strips of 64 and 36 rows, the point-to-point form; 128 rows: equal strips, one all-gather)
        for H, W in ((100, 5), (128, 7)):
            full = (np.arange(H * W * 4) % 251).astype(np.uint8).reshape(H, W, 4)
            r0, r1 = partition.strip_for_rank(H, rank, world)
            image = torch.zeros((H, W, 4), dtype=torch.uint8)
            image[r0:r1] = torch.from_numpy(full[r0:r1].copy())
            partition.all_gather_strips_in_place(image)
            assert np.array_equal(image.numpy(), full)
            if H == 128:                                # the form RCCL takes for equal strips, on gloo
                image[:r0] = 0; image[r1:] = 0
                partition.all_gather_strips_in_place(image, collective=True)
                assert np.array_equal(image.numpy(), full)
        # in place: at the size of a 16384^2 target's strips the exchange needs no second copy of a strip
        import resource
        for H in (4096, 4096 + 32):                     # equal strips; unequal strips
            W = 8192                                    # 128 MiB image, 64 MiB strips
            image = torch.zeros((H, W, 4), dtype=torch.uint8)
            r0, r1 = partition.strip_for_rank(H, rank, world)
            image[r0:r1] = rank + 1
            image[:] += 0                               # every page touched before the measurement
            dist.barrier()
            before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
            partition.all_gather_strips_in_place(image)
            grown = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) * 1024
            strip_bytes = min(b - a for a, b in (partition.strip_for_rank(H, r, world) for r in range(world))) * W * 4
            o0, o1 = partition.strip_for_rank(H, 1 - rank, world)
            assert int(image[o0:o1].min()) == int(image[o0:o1].max()) == 2 - rank
            assert grown < strip_bytes, f"the exchange grew the process by {grown} bytes; a strip has {strip_bytes}"
            del image
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_gathers_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_texture_as_tensor_refuses_two_hip_runtimes():
    """A process that loaded librenderer_hip.so before importing PyTorch holds two HIP runtimes: torch cannot use the
    renderer's device memory then, and the zero-copy view says so instead of faulting."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from renderer_amd import _ffi, partition\n"
        "_ffi.lib()\n"
        "assert _ffi.torch_came_first is False\n"
        "class T: size = (4, 4, 1); handle = None\n"
        "try:\n"
        "    partition.texture_as_tensor(T(), 'cpu')\n"
        "except _ffi.RendererError as e:\n"
        "    assert 'import torch first' in str(e); print('refused')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "refused" in out.stdout, out.stderr[-2000:]
