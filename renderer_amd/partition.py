"""Work partitioning over the GPUs of one node (one process per GPU).  Both partitionings are pure functions of
(rank, world), so they need no communication:

* frame-parallel recording: rank r renders and compresses frames f with (f - 1) % world == r (frame numbers start at 1,
  reference src/video_recorder.rs:95) into its own stream files; the `.sz` format is already a set of independent
  streams re-ordered by frame number at read time (reference src/decompressor.rs:96-151);
* sort-first strips for one very large target: contiguous row strips aligned to the 32-pixel tile height; the order of
  primitives inside a strip is unchanged, so the union of the strips equals the single-GPU image.

The only exchange steps are gathers of finished results to rank 0 (encoded frames for a single ordered sink such as the
ffmpeg pipe; strips of a target that a later pass samples), done with torch.distributed (RCCL on GPUs, gloo on CPU).

PyTorch bundles its own HIP runtime: a process that uses torch on the GPU together with the renderer must import torch
before the renderer's library is loaded (or set RH_HIP_RUNTIME=torch), so that both share one runtime.
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


def comm_device(device, group=None):
    """Where the tensors handed to the collectives live: the GPU with RCCL; host memory with gloo, which has no
    all-gather or send/recv for device tensors (rehearsals of the multi-process path on one GPU use gloo)."""
    import torch.distributed as dist
    return "cpu" if dist.get_backend(group) == "gloo" else device


def gather_bytes(payload, dst=0, device="cpu", group=None):
    """Gathers one variable-length byte string per rank to `dst` (RCCL/gloo have no gatherv: lengths are all-gathered,
    then every peer sends its bytes point to point).  Returns the list of payloads in rank order on `dst`, else None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = comm_device(device, group)
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


def all_gather_strips_in_place(image, group=None, collective=None):
    """Every rank holds its own strip (strip_for_rank) of `image`, a contiguous uint8 tensor [height, width, 4]; on return
    every rank holds the whole image.  Row strips of a contiguous image are contiguous in memory, so the collective works
    on the image itself and nothing is staged: strips of equal size are one all_gather_into_tensor whose input is the
    rank's slice of the output; otherwise every rank posts one send of its slice and one receive into each peer's slice
    (a grouped point-to-point exchange; RCCL runs the group as one launch).  `collective`: None picks the all-gather for
    equal strips unless the backend is gloo, whose all-gather stages a copy of the whole image inside the library while
    its sends and receives work on the caller's memory; True / False force one form (equal strips only for True)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return
    if not image.is_contiguous():
        raise ValueError("all_gather_strips_in_place: the image must be contiguous")
    height = image.shape[0]
    rows = [strip_for_rank(height, r, world) for r in range(world)]
    r0, r1 = rows[rank]
    equal = all(b - a == r1 - r0 for a, b in rows)
    if collective is None:
        collective = equal and dist.get_backend(group) != "gloo"
    if collective:
        if not equal:
            raise ValueError("all_gather_strips_in_place: the all-gather form needs strips of equal size")
        dist.all_gather_into_tensor(image.view(-1), image[r0:r1].view(-1), group=group)
        return
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        p0, p1 = rows[peer]
        if r1 > r0:
            ops.append(dist.P2POp(dist.isend, image[r0:r1], peer, group))
        if p1 > p0:
            ops.append(dist.P2POp(dist.irecv, image[p0:p1], peer, group))
    for work in dist.batch_isend_irecv(ops) if ops else []:
        work.wait()


class _DeviceArray:
    """Exposes a device allocation of the renderer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, shape, typestr="|u1"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def texture_as_tensor(texture, device):
    """The texels of an RGBA8/BGRA8 texture (layer 0) as a torch uint8 tensor [height, width, 4] that aliases the
    texture's device memory: what the strip all-gather sends and receives."""
    import torch
    from . import _ffi
    w, h, _ = texture.size
    lib = _ffi.lib()
    if not _ffi.torch_came_first:
        # two HIP runtimes in the process: torch's cannot see the renderer's allocations (it would fault, or find no GPU)
        raise _ffi.RendererError("texture_as_tensor: librenderer_hip.so was loaded before PyTorch was imported, so the two use "
                                 "separate HIP runtimes and cannot share device memory; import torch first, or set RH_HIP_RUNTIME=torch")
    ptr = lib.rh_texture_device_ptr(texture.handle)
    return torch.as_tensor(_DeviceArray(ptr, (h, w, 4)), device=device)


def exchange_strips(renderer, texture, rank, world, device, group=None):
    """After every rank has rendered its strip of `texture` (Renderer.set_strip(*strip_for_rank(...))), makes the whole
    image present on every rank, so that a following pass can sample the complete target: the strips are all-gathered in
    place, out of and into the texture's device memory (RCCL over xGMI between GPUs).  With gloo, which takes no device
    tensors (rehearsals of the multi-process path on one GPU), the image goes through host memory."""
    import torch
    renderer.synchronize()                       # the strip is rendered on the renderer's own HIP stream
    if world == 1:
        return
    image = texture_as_tensor(texture, device)
    if comm_device(device, group) == "cpu":
        r0, r1 = strip_for_rank(image.shape[0], rank, world)
        host = torch.empty(image.shape, dtype=torch.uint8)
        host[r0:r1] = image[r0:r1].cpu()
        all_gather_strips_in_place(host, group=group)
        image.copy_(host)
    else:
        all_gather_strips_in_place(image, group=group)
    torch.cuda.synchronize()
