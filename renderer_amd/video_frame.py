"""Captured-frame metadata plus payload handle (reference src/video_frame.rs:6-104)."""
import ctypes as C
import enum
import threading

import numpy as np

from . import _ffi
from .format import Format


class FrameStatus(enum.IntEnum):
    # declaration order = bincode discriminant (reference src/video_frame.rs:25-29)
    Captured = 0    # image_data is Some
    Dropped = 1     # dropped to save memory, image_data is None
    Missing = 2     # absent from the compressed files, image_data is None

    def __str__(self):
        return self.name.lower()


class AtomicUsize:
    """Arc<AtomicUsize> stand-in: the in-flight capture byte budget shared by recorder and frames."""

    def __init__(self, value=0):
        self._v = value
        self._lock = threading.Lock()

    def fetch_add(self, n):
        with self._lock:
            old = self._v
            self._v += n
            return old

    def fetch_sub(self, n):
        with self._lock:
            old = self._v
            self._v -= n
            return old

    def load(self):
        return self._v


class ImageData:
    """Either a mapped device buffer (Buffer) or owned bytes (Bytes) (reference src/video_frame.rs:42-68)."""

    def __init__(self, buffer=None, data=None):
        self._buffer = buffer       # rh_buffer* (c_void_p), one reference owned
        self._bytes = data

    @staticmethod
    def Buffer(handle):
        return ImageData(buffer=handle)

    @staticmethod
    def Bytes(data):
        return ImageData(data=bytes(data))

    def is_buffer(self):
        return self._buffer is not None

    def buffer(self):
        if self._buffer is None:
            raise _ffi.RendererError("The buffer is no longer available.")
        return self._buffer

    def bytes(self):
        if self._buffer is not None:
            raise _ffi.RendererError("Please use ImageData::bytes_fn instead.")
        return self._bytes

    def bytes_fn(self, f):
        """Calls f with the image bytes (a zero-copy view of the mapped pinned memory for buffers)."""
        if self._buffer is not None:
            ptr = C.POINTER(C.c_uint8)()
            n = C.c_size_t()
            _ffi.check(_ffi.lib().rh_buffer_mapped_ptr(self._buffer, C.byref(ptr), C.byref(n)))
            f(np.ctypeslib.as_array(ptr, shape=(n.value,)))
        else:
            f(self._bytes)

    def release(self):
        if self._buffer is not None:
            _ffi.lib().rh_buffer_release(self._buffer)
            self._buffer = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class VideoFrame:
    def __init__(self, status=FrameStatus.Missing, image_data=None, width=0, height=0, format=Format.RgbaU8,
                 unpadded_bytes_per_row=0, padded_bytes_per_row=0, frame_number=0, frame_size_in_bytes=0,
                 buffer_size_in_bytes=None):
        self.status = status
        self.image_data = image_data
        self.width = width
        self.height = height
        self.format = Format(format)
        self.unpadded_bytes_per_row = unpadded_bytes_per_row
        self.padded_bytes_per_row = padded_bytes_per_row
        self.frame_number = frame_number
        self.frame_size_in_bytes = frame_size_in_bytes
        self.buffer_size_in_bytes = buffer_size_in_bytes if buffer_size_in_bytes is not None else AtomicUsize(0)
        self._dropped = False

    def drop(self):
        """Returns the frame's bytes to the budget (reference src/video_frame.rs:70-77); idempotent."""
        if self._dropped:
            return
        self._dropped = True
        if self.image_data is not None:
            self.buffer_size_in_bytes.fetch_sub(self.frame_size_in_bytes)
            self.image_data.release()

    def __del__(self):
        try:
            self.drop()
        except Exception:
            pass

    # ---- bincode 2 `config::standard()` (varint, little endian), field order = declaration order ------------------
    # (reference src/video_frame.rs:5-21; ImageData encodes to zero bytes: src/video_frame.rs:85-90)
    def encode(self):
        out = bytearray()
        out += _varint(int(self.status))
        out += b"\x01" if self.image_data is not None else b"\x00"      # Option tag; Some(ImageData) adds nothing
        out += _varint(self.width)
        out += _varint(self.height)
        out += _varint(int(self.format))
        out += _varint(self.unpadded_bytes_per_row)
        out += _varint(self.padded_bytes_per_row)
        out += _varint(self.frame_number)
        out += _varint(self.frame_size_in_bytes)
        out += _varint(self.buffer_size_in_bytes.load())                # Arc<AtomicUsize> encodes its current value
        return bytes(out)

    @staticmethod
    def decode(data):
        pos = 0

        def take():
            nonlocal pos
            v, pos = _read_varint(data, pos)
            return v

        status = FrameStatus(take())
        tag = data[pos]; pos += 1
        if tag not in (0, 1):
            raise ValueError("bad Option tag")
        image_data = ImageData.Bytes(b"") if tag == 1 else None          # reference src/video_frame.rs:92-104
        width, height = take(), take()
        fmt = Format(take())
        unpadded, padded, frame_number, frame_size = take(), take(), take(), take()
        budget = AtomicUsize(take())
        frame = VideoFrame(status, image_data, width, height, fmt, unpadded, padded, frame_number, frame_size, budget)
        frame._dropped = True        # a decoded frame owns no budget
        return frame, pos


def _varint(v):
    """bincode 2 varint: < 251 one byte; else 251 + u16, 252 + u32, 253 + u64 (little endian)."""
    if v < 251:
        return bytes([v])
    if v < (1 << 16):
        return b"\xfb" + v.to_bytes(2, "little")
    if v < (1 << 32):
        return b"\xfc" + v.to_bytes(4, "little")
    return b"\xfd" + v.to_bytes(8, "little")


def _read_varint(data, pos):
    b = data[pos]
    if b < 251:
        return b, pos + 1
    width = {251: 2, 252: 4, 253: 8}.get(b)
    if width is None:
        raise ValueError("unsupported varint marker")
    return int.from_bytes(data[pos + 1:pos + 1 + width], "little"), pos + 1 + width
