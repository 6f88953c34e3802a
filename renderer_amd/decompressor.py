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
    """Decodes one LZ4 frame with the library's host decoder; a truncated tail is tolerated."""
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


def read_packets(raw):
    """Yields VideoFrames from a decompressed stream; stops at an unclean end (reference src/decompressor.rs:214-243)."""
    pos = 0
    while True:
        if len(raw) - pos < U64_LEN:
            return
        packet_len = int.from_bytes(raw[pos:pos + 8], "big"); pos += 8
        if len(raw) - pos < U64_LEN:
            return
        video_frame_len = int.from_bytes(raw[pos:pos + 8], "big"); pos += 8
        if len(raw) - pos < video_frame_len:
            return
        try:
            video_frame, _ = VideoFrame.decode(raw[pos:pos + video_frame_len])
        except (ValueError, IndexError):
            return
        pos += video_frame_len
        if video_frame.image_data is not None:
            remainder_len = packet_len - U64_LEN - U64_LEN - video_frame_len
            if len(raw) - pos < remainder_len:
                return
            video_frame.image_data = ImageData.Bytes(raw[pos:pos + remainder_len])
            pos += remainder_len
        yield video_frame


class Decompressor:
    def __init__(self, directory, remove_files_after_decompression):
        self.directory = directory
        self.remove_files_after_decompression = remove_files_after_decompression

    @staticmethod
    def can_run(directory):
        return len(scan_directory_for_timestamps(directory)) > 0

    def decompress_from_disk(self, per_thread_function, in_order_function):
        """per_thread_function(video_frame, timestamp) -> T runs as each frame is decoded;
        in_order_function(video_frame, result, timestamp) is called in frame order, result = ("ok", T) or ("err", message)."""
        ordered = scan_directory_for_timestamps(self.directory)
        for timestamp, filenames in ordered.items():
            filenames.sort()
            workers = [self._worker(f, per_thread_function, timestamp) for f in filenames]
            order_frames_from_workers(workers, in_order_function, timestamp)
        # files are removed only after everything succeeded (reference src/decompressor.rs:43-50)
        if self.remove_files_after_decompression:
            for filenames in ordered.values():
                for filename in filenames:
                    try:
                        os.remove(os.path.join(self.directory, filename))
                    except OSError:
                        pass

    def _worker(self, filename, per_thread_function, timestamp):
        with open(os.path.join(self.directory, filename), "rb") as f:
            data = f.read()
        raw, _complete = lz4f_decompress(data)
        for video_frame in read_packets(raw):
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
