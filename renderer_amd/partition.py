"""Work partitioning over the GPUs of one node (one process per GPU).  Both partitionings are pure functions of
(rank, world), so they need no communication:

* frame-parallel recording: rank r renders and compresses frames f with (f - 1) % world == r (frame numbers start at 1,
  reference src/video_recorder.rs:95) into its own stream files; the `.sz` format is already a set of independent
  streams re-ordered by frame number at read time (reference src/decompressor.rs:96-151);
* sort-first strips for one very large target: contiguous row strips aligned to the 32-pixel tile height; the order of
  primitives inside a strip is unchanged, so the union of the strips equals the single-GPU image.

The only exchange steps are gathers of finished results to rank 0 (encoded frames for a single ordered sink such as the
ffmpeg pipe; strips of a target that a later pass samples), done with torch.distributed (RCCL on GPUs, gloo on CPU).
"""
TILE = 32


def frames_for_rank(n_frames, rank, world):
    """1-based frame numbers rendered by `rank`."""
    return list(range(1 + rank, n_frames + 1, world))


def rank_of_frame(frame_number, world):
    return (frame_number - 1) % world


def stream_ids_for_rank(rank, streams_per_rank):
    """Global stream (file) indices owned by `rank`: files are `<timestamp>--<i>.sz` with i unique across ranks."""
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def strip_for_rank(height, rank, world, tile=TILE):
    """Pixel rows [row0, row1) of a `height`-row target rendered by `rank`: tile rows are dealt as evenly as possible,
    earlier ranks taking the remainder."""
    tile_rows = (height + tile - 1) // tile
    base, extra = divmod(tile_rows, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return min(first * tile, height), min((first + count) * tile, height)


def gather_bytes(payload, dst=0, device="cpu", group=None):
    """Gathers one variable-length byte string per rank to `dst` (RCCL/gloo have no gatherv: lengths are all-gathered,
    then every peer sends its bytes point to point).  Returns the list of payloads in rank order on `dst`, else None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    length = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    lengths = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lengths, length, group=group)
    lengths = [int(x.item()) for x in lengths]
    mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device) if payload else torch.empty(0, dtype=torch.uint8, device=device)
    if rank != dst:
        if len(payload):
            dist.send(mine, dst=dst, group=group)
        return None
    out = []
    for r in range(world):
        if r == dst:
            out.append(bytes(payload))
        elif lengths[r] == 0:
            out.append(b"")
        else:
            buf = torch.empty(lengths[r], dtype=torch.uint8, device=device)
            dist.recv(buf, src=r, group=group)
            out.append(bytes(buf.cpu().numpy().tobytes()))
    return out


def all_gather_strips(strip, height, device="cpu", group=None):
    """All-gathers the row strips of a target (uint8 tensor [rows, width, 4] per rank, rows = strip_for_rank) into the
    full image on every rank, for a later pass that samples the whole target."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    width = strip.shape[1]
    rows = [strip_for_rank(height, r, world) for r in range(world)]
    max_rows = max(r1 - r0 for r0, r1 in rows)
    padded = torch.zeros((max_rows, width, 4), dtype=torch.uint8, device=device)
    padded[:strip.shape[0]] = strip
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][:rows[r][1] - rows[r][0]] for r in range(world)], dim=0)
