"""Frame compressor (reference src/compressor.rs:15-148): N LZ4-frame stream files `<timestamp>--<i>.sz`, each a
sequence of packets [packet_len u64 BE][video_frame_len u64 BE][bincode(VideoFrame)][raw padded image].

The reference runs one CPU worker thread per stream.  Here the streams are compressed on the GPU (one workgroup per
stream, librenderer_hip.so rh_compressor_*): captured frames stay in device memory and only compressed bytes are read
back.  Frame f goes to stream (f - 1) % n_streams (the reference's assignment is a channel race, so any assignment is
valid; the decompressor re-orders by frame number).  The terminal statistics display of the reference is not
reproduced."""
import ctypes as C
import datetime
import os
import time

from . import _ffi

CLEVEL_MAX = 12          # lz4f::CLEVEL_MAX


def generate_timestamp(now=None):
    """Utc::now().to_rfc3339_opts(SecondsFormat::Millis, true) with ':' replaced by '_' (reference src/compressor.rs:71-73)."""
    now = now or datetime.datetime.now(datetime.timezone.utc)
    return now.strftime("%Y-%m-%dT%H_%M_%S.") + f"{now.microsecond // 1000:03d}Z"


class Compressor:
    def __init__(self, renderer, directory, max_frames_queued=None, lz4_compression_level=0, print_stats=False,
                 n_streams=64, timestamp=None, max_buffer_size_in_megabytes=None, wait_for_rounds=False):
        if lz4_compression_level > CLEVEL_MAX:
            raise AssertionError(f"Please choose a compression level in the range 0..={CLEVEL_MAX}")
        os.makedirs(directory, exist_ok=True)
        self.directory = directory
        # wait_for_rounds: a hand-off returns when its round has been compressed, so that rendering and compression take
        # turns on the GPU instead of sharing it (the file writes still go on beside the rendering)
        self.wait_for_rounds = wait_for_rounds
        self.timestamp = timestamp or generate_timestamp()
        self.n_streams = n_streams
        # packets are compressed when every stream has one queued, or when this many frames wait
        self.max_frames_queued = max_frames_queued if max_frames_queued else n_streams
        # The capture budget the recorder was started with (Renderer.start_recording), when the caller passes it on: a
        # frame's bytes go back to the budget once its round has been compressed, so a round plus the frames captured
        # while it is being compressed must fit; rounds are cut short accordingly (see compress_to_disk).
        self.max_buffer_size_in_bytes = None
        if max_buffer_size_in_megabytes is not None:
            import numpy as np
            self.max_buffer_size_in_bytes = int(np.float32(max_buffer_size_in_megabytes) * np.float32(1024.0) * np.float32(1024.0))
        self.print_stats = print_stats
        out = C.c_void_p()
        _ffi.check(_ffi.lib().rh_compressor_create(renderer.device, n_streams, lz4_compression_level, C.byref(out)))
        self.handle = out
        self.files = [open(os.path.join(directory, f"{self.timestamp}--{i}.sz"), "wb", buffering=0) for i in range(n_streams)]
        self.queued = []            # frames whose packets are queued on the device
        self.in_flight = []         # frames of the round being compressed: they hold their share of the budget until it is done
        self.frames_captured = 0
        self.frames_dropped = 0
        self.raw_video_size = 0
        self.compressed_size = 0
        self._pool = None
        self._background = None     # one thread that appends finished rounds to the files beside the render thread
        self._draining = None
        # host seconds: waiting for the previous hand-off's file writes, inside rh_compressor_submit/flush (waits for the
        # round in flight), waiting for a round's bytes to arrive from the device, writing them to the files
        self.host_seconds = {"join_drain": 0.0, "submit": 0.0, "await_bytes": 0.0, "write_files": 0.0}

    def compress_to_disk(self, video_frame):
        stream = (video_frame.frame_number - 1) % self.n_streams
        meta = video_frame.encode()
        image = video_frame.image_data.buffer() if (video_frame.image_data is not None and video_frame.image_data.is_buffer()) else None
        buf = C.create_string_buffer(meta, len(meta))
        _ffi.check(_ffi.lib().rh_compressor_write_packet(self.handle, stream, buf, len(meta), image))
        if video_frame.image_data is not None:
            self.frames_captured += 1
            self.raw_video_size += video_frame.frame_size_in_bytes
        else:
            self.frames_dropped += 1
        self.queued.append(video_frame)
        if len(self.queued) >= self._round_size(video_frame):
            self.submit()

    def _round_size(self, video_frame):
        """Frames per round: one per stream, unless the capture budget holds fewer.  The recorder accepts a frame while
        the bytes in flight do not exceed the budget (reference src/video_recorder.rs:80-82), i.e. budget // frame + 1
        frames; half of them may wait in a round while the other half is captured."""
        n = min(self.n_streams, self.max_frames_queued)
        if self.max_buffer_size_in_bytes is not None and video_frame.frame_size_in_bytes > 0:
            frames_in_budget = self.max_buffer_size_in_bytes // video_frame.frame_size_in_bytes + 1
            n = min(n, max(1, frames_in_budget // 2))
        return n

    def _release(self, frames):
        for frame in frames:
            frame.drop()                # returns the frame's bytes to the capture budget (reference src/video_frame.rs:70-77)

    def submit(self):
        """Hands the queued packets to the GPU (streams run concurrently there) and, while they are being compressed,
        appends the bytes of the previous hand-off to the files -- the reference's worker threads do both beside the
        render thread (src/compressor.rs:100-131)."""
        t0 = time.perf_counter()
        self._join_drain()
        t1 = time.perf_counter()
        _ffi.check(_ffi.lib().rh_compressor_submit(self.handle))    # waits for the round in flight, then starts the queued one
        self.host_seconds["join_drain"] += t1 - t0
        self.host_seconds["submit"] += time.perf_counter() - t1
        self._release(self.in_flight)   # compressed: the reference's worker drops a frame once it has written it
        self.in_flight, self.queued = self.queued, []
        if self.wait_for_rounds:
            _ffi.check(_ffi.lib().rh_compressor_wait(self.handle))
            self._release(self.in_flight)
            self.in_flight = []
        # The file writes go to a background thread: until the next hand-off this thread only queues packets
        # (rh_compressor_write_packet), which touches none of the state that peek/consume use.
        if self._background is None:
            from concurrent.futures import ThreadPoolExecutor
            self._background = ThreadPoolExecutor(max_workers=1)
        self._draining = self._background.submit(self._drain)

    def flush(self):
        """Compresses all queued packets, waits for them and appends the bytes to the files."""
        t0 = time.perf_counter()
        self._join_drain()
        t1 = time.perf_counter()
        _ffi.check(_ffi.lib().rh_compressor_flush(self.handle))
        self.host_seconds["join_drain"] += t1 - t0
        self.host_seconds["submit"] += time.perf_counter() - t1
        self._release(self.in_flight + self.queued)
        self.in_flight, self.queued = [], []
        self._drain()

    def _join_drain(self):
        """Waits for the background file writes of the previous hand-off (and re-raises what they raised)."""
        if self._draining is not None:
            pending, self._draining = self._draining, None
            pending.result()

    def _drain(self):
        """Writes every ready byte to its stream file straight from the library's pinned memory.  The files are written
        by a few threads at once (file writes release the interpreter lock), and a stream's write starts as soon as its
        bytes have arrived: the copies of a round come in pieces of some tens of streams, so the first files are being
        written while the last streams still cross PCIe."""
        L = _ffi.lib()
        pool = self._writers()
        t_start, waited = time.perf_counter(), 0.0
        streams = range(self.n_streams)
        while streams:
            writes = []
            for i in streams:
                # peek exposes one run at a time (and waits for the copies that bring it): consume after the write,
                # then look again
                ptr, n = C.c_void_p(), C.c_size_t()
                t0 = time.perf_counter()
                _ffi.check(L.rh_compressor_peek(self.handle, i, C.byref(ptr), C.byref(n)))
                waited += time.perf_counter() - t0
                if n.value > (64 << 10):
                    writes.append((i, n.value, pool.submit(self._write_run, (i, ptr.value, n.value))))
                elif n.value:
                    self._write_run((i, ptr.value, n.value))
                    writes.append((i, n.value, None))
            streams = []
            for i, n, pending in writes:
                if pending is not None:
                    pending.result()
                _ffi.check(L.rh_compressor_consume(self.handle, i, n))
                self.compressed_size += n
                streams.append(i)
        self.host_seconds["await_bytes"] += waited
        self.host_seconds["write_files"] += time.perf_counter() - t_start - waited

    def _write_run(self, run):
        i, ptr, n = run
        view = memoryview((C.c_uint8 * n).from_address(ptr)).cast("B")
        done = 0
        while done < n:                 # an unbuffered file may take less than it is given
            done += self.files[i].write(view[done:])

    def _writers(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=min(int(os.environ.get("RH_WRITER_THREADS", "16")), os.cpu_count() or 1))
        return self._pool

    def drain_kernel_stats(self):
        """What the compression kernel did since the previous call: launches, packets, bytes in and out, kernel time."""
        st = _ffi.CompressorStats()
        _ffi.check(_ffi.lib().rh_compressor_drain_stats(self.handle, C.byref(st)))
        return {"rounds": st.rounds, "jobs": st.jobs, "bytes_in": st.bytes_in, "bytes_out": st.bytes_out, "kernel_ms": st.kernel_ms}

    def finish(self):
        """Waits for everything queued, writes the end marks and closes the files (reference src/compressor.rs:46-62)."""
        if self.handle is None:
            return
        self.flush()
        for i in range(self.n_streams):
            _ffi.check(_ffi.lib().rh_compressor_finish_stream(self.handle, i))
        self._drain()
        for f in self.files:
            f.close()
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        if self._background is not None:
            self._background.shutdown()
            self._background = None
        _ffi.lib().rh_compressor_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.finish()
        except Exception:
            pass
