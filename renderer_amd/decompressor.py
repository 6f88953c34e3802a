"""Inverse of the compressor (reference src/decompressor.rs:21-285): groups `.sz` files by timestamp, decodes each
file's LZ4 frame and packet sequence, re-orders frames globally by frame_number starting at 1 and synthesises Missing
frames for gaps.  The reference runs one reader thread per file behind a rendezvous channel; here each file is a lazy
iterator, which gives the same pull order."""
import ctypes as C
import datetime
import heapq
import os

from . import _ffi
from .video_frame import FrameStatus, ImageData, VideoFrame

U64_LEN = 8


def recover_timestamp_from_filename(filename):
    """reference src/decompressor.rs:84-94"""
    if not filename.endswith(".sz"):
        return None
    prefix = filename.split("--")[0]
    try:
        return datetime.datetime.fromisoformat(prefix.replace("_", ":").replace("Z", "+00:00"))
    except ValueError:
        return None


def scan_directory_for_timestamps(directory):
    """timestamp -> filenames, ordered by timestamp; empty files are skipped (reference src/decompressor.rs:58-82)."""
    groups = {}
    try:
        listing = os.listdir(directory)
    except OSError:
        return {}
    for filename in listing:
        path = os.path.join(directory, filename)
        try:
            if os.path.getsize(path) == 0:
                continue
        except OSError:
            continue
        timestamp = recover_timestamp_from_filename(filename)
        if timestamp is None:
            continue
        groups.setdefault(timestamp, []).append(filename)
    return dict(sorted(groups.items()))


def lz4f_decompress(data):
    """Decodes one LZ4 frame held in memory with the library's host decoder; a truncated tail is tolerated."""
    data = bytes(data)
    cap = max(1 << 16, len(data) * 4)
    L = _ffi.lib()
    src = C.create_string_buffer(data, len(data))
    while True:
        dst = (C.c_uint8 * cap)()
        produced, consumed, complete = C.c_size_t(), C.c_size_t(), C.c_int()
        rc = L.rh_lz4f_decompress(src, len(data), dst, cap, C.byref(produced), C.byref(consumed), C.byref(complete))
        if rc == 0:
            return bytes(memoryview(dst)[:produced.value]), bool(complete.value)
        msg = L.rh_last_error().decode()
        if "too small" not in msg:
            raise _ffi.RendererError(msg)
        cap *= 4


class StreamReaders:
    """The readers of one group of stream files (reference src/decompressor.rs:197-198: a BufReadDecompressor over a
    BufReader per file), decoded together: the files are read a piece at a time, and the complete LZ4F blocks of all of
    them go through one decoder run (rh_decoder_*: one wavefront per stream on the device; on the host when no device
    is given).  What a reader hands out is bounded: a stream with `high_water` decoded bytes waiting is not fed."""

    FILE_PIECE = 1 << 20
    LARGEST_BLOCK = (4 << 20) + 8       # an LZ4F block with its header and checksum at the largest block size

    def __init__(self, paths, device=None, high_water=64 << 20):
        self.files = [open(p, "rb") for p in paths]
        self.eof = [False] * len(paths)
        self.bytes = [bytearray() for _ in paths]       # decoded, not yet read
        self.high_water = high_water
        out = C.c_void_p()
        _ffi.check(_ffi.lib().rh_decoder_create(device, len(paths), C.byref(out)))
        self.handle = out

    def close(self):
        if self.handle is not None:
            _ffi.lib().rh_decoder_destroy(self.handle)
            self.handle = None
        for f in self.files:
            f.close()

    def _info(self, i):
        state, buffered, ready = C.c_int(), C.c_size_t(), C.c_size_t()
        _ffi.check(_ffi.lib().rh_decoder_stream_info(self.handle, i, C.byref(state), C.byref(buffered), C.byref(ready)))
        return state.value, buffered.value, ready.value

    def _pump(self, wanted):
        """One decoder run over every stream that can use one; returns False when stream `wanted` cannot get any further."""
        L = _ffi.lib()
        fed = False
        for i, f in enumerate(self.files):
            state, buffered, _ = self._info(i)
            if state != 0 or self.eof[i] or buffered >= max(2 * self.FILE_PIECE, self.LARGEST_BLOCK):
                continue              # (enough fed for the next run: at least one whole block, whatever the piece size)
            if i != wanted and len(self.bytes[i]) >= self.high_water:
                continue
            piece = f.read(self.FILE_PIECE)
            if not piece:
                self.eof[i] = True
                continue
            _ffi.check(L.rh_decoder_feed(self.handle, i, piece, len(piece)))
            fed = fed or i == wanted
        before = self._info(wanted)[1]
        _ffi.check(L.rh_decoder_run(self.handle))
        progressed = False
        for i in range(len(self.files)):
            if i != wanted and len(self.bytes[i]) >= self.high_water:
                continue                                    # (its decoded bytes stay with the decoder for now)
            while True:
                ptr, n = C.c_void_p(), C.c_size_t()
                _ffi.check(L.rh_decoder_peek(self.handle, i, C.byref(ptr), C.byref(n)))
                if not n.value:
                    break
                self.bytes[i] += C.string_at(ptr.value, n.value)
                _ffi.check(L.rh_decoder_consume(self.handle, i, n.value))
                if i == wanted:
                    progressed = True
        if progressed:
            return True
        state, after, _ = self._info(wanted)
        # without new decoded bytes the stream can still get further if it runs and new bytes arrived or the run took some in
        return state == 0 and (fed or after != before)

    def read_exact(self, i, n):
        """n decoded bytes of stream i, or None when the stream ends before (Read::read_exact failing)."""
        buf = self.bytes[i]
        while len(buf) < n:
            if not self._pump(i):
                if len(buf) >= n:
                    break
                return None
        out = bytes(buf[:n])
        del buf[:n]
        return out


def read_packets(read_exact):
    """Yields VideoFrames from a decoded stream, `read_exact(n)` returning n bytes or None at the end; stops at an
    unclean end (reference src/decompressor.rs:214-243)."""
    while True:
        head = read_exact(U64_LEN)
        if head is None:
            return
        packet_len = int.from_bytes(head, "big")
        head = read_exact(U64_LEN)
        if head is None:
            return
        video_frame_len = int.from_bytes(head, "big")
        if video_frame_len > (1 << 20):                     # (a bincode VideoFrame takes tens of bytes)
            return
        meta = read_exact(video_frame_len)
        if meta is None:
            return
        try:
            video_frame, _ = VideoFrame.decode(meta)
        except (ValueError, IndexError):
            return
        if video_frame.image_data is not None:
            remainder_len = packet_len - U64_LEN - U64_LEN - video_frame_len
            if remainder_len < 0:
                return
            image = read_exact(remainder_len)
            if image is None:
                return
            video_frame.image_data = ImageData.Bytes(image)
        yield video_frame


def packets_of_bytes(raw):
    """read_packets over bytes held in memory."""
    pos = [0]

    def read_exact(n):
        if len(raw) - pos[0] < n:
            return None
        out = raw[pos[0]:pos[0] + n]
        pos[0] += n
        return out

    return read_packets(read_exact)


class Decompressor:
    def __init__(self, directory, remove_files_after_decompression, renderer=None):
        """renderer: the Renderer whose device decodes the LZ4F blocks of all the files together; None: the library's host
        decoder does (the reference decodes on one CPU thread per file)."""
        self.directory = directory
        self.remove_files_after_decompression = remove_files_after_decompression
        self.device = renderer.device if renderer is not None else None

    @staticmethod
    def can_run(directory):
        return len(scan_directory_for_timestamps(directory)) > 0

    def decompress_from_disk(self, per_thread_function, in_order_function):
        """per_thread_function(video_frame, timestamp) -> T runs as each frame is decoded;
        in_order_function(video_frame, result, timestamp) is called in frame order, result = ("ok", T) or ("err", message)."""
        ordered = scan_directory_for_timestamps(self.directory)
        for timestamp, filenames in ordered.items():
            filenames.sort()
            readers = StreamReaders([os.path.join(self.directory, f) for f in filenames], self.device)
            try:
                workers = [self._worker(readers, i, per_thread_function, timestamp) for i in range(len(filenames))]
                order_frames_from_workers(workers, in_order_function, timestamp)
            finally:
                readers.close()
        # files are removed only after everything succeeded (reference src/decompressor.rs:43-50)
        if self.remove_files_after_decompression:
            for filenames in ordered.values():
                for filename in filenames:
                    try:
                        os.remove(os.path.join(self.directory, filename))
                    except OSError:
                        pass

    def _worker(self, readers, i, per_thread_function, timestamp):
        for video_frame in read_packets(lambda n: readers.read_exact(i, n)):
            yield video_frame, per_thread_function(video_frame, timestamp)


def order_frames_from_workers(workers, in_order_function, timestamp):
    """Min-heap re-ordering by frame_number (reference src/decompressor.rs:96-184)."""
    heap = []
    expected_frame = 1
    seq = 0
    while True:
        # pull from every worker until it yields a frame WITH image data (or runs out), mirroring how dropped frames
        # cluster in one file on the compression side
        remaining = []
        for worker in workers:
            exhausted = False
            while True:
                item = next(worker, None)
                if item is None:
                    exhausted = True
                    break
                video_frame, t = item
                heapq.heappush(heap, (video_frame.frame_number, seq, video_frame, t)); seq += 1
                if video_frame.image_data is not None:
                    break
            if not exhausted:
                remaining.append(worker)
        workers = remaining

        advanced = False
        while True:
            if not heap:
                if not workers:
                    return
                break
            if heap[0][0] == expected_frame:
                _, _, video_frame, t = heapq.heappop(heap)
                in_order_function(video_frame, ("ok", t), timestamp)
                expected_frame += 1
                advanced = True
            else:
                break
        if advanced:
            continue
        # no progress: frames are missing from the files (e.g. a deleted .sz); yield Missing frames up to the next one
        next_available = heap[0][0]
        while True:
            missing = VideoFrame(status=FrameStatus.Missing, image_data=None, frame_number=expected_frame)
            in_order_function(missing, ("err", "The frame was missing from the compressed files."), timestamp)
            expected_frame += 1
            if expected_frame == next_available:
                break
