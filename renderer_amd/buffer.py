"""Growable f32 device buffer with a generation counter (reference src/buffer.rs:16-67)."""
import ctypes as C

import numpy as np

from . import _ffi

INITIAL_SIZE = 4 * 16      # enough for a mat4 uniform (reference src/buffer.rs:16)
HEADROOM = 4 * 256         # reference src/buffer.rs:17
U64_MAX = (1 << 64) - 1


def next_power_of_two(n):
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


class Buffer:
    def __init__(self, device, usage):
        self.device = device            # rh_device*
        self.usage = usage
        self.size = INITIAL_SIZE
        self.generation = 0
        self.previous = U64_MAX
        self.handle = self._create(INITIAL_SIZE, None)

    def _create(self, size, init):
        out = C.c_void_p()
        if init is None:
            _ffi.check(_ffi.lib().rh_buffer_create(self.device, size, self.usage, None, 0, C.byref(out)))
        else:
            _ffi.check(_ffi.lib().rh_buffer_create(self.device, size, self.usage, init.ctypes.data, init.nbytes, C.byref(out)))
        return out

    def set_data(self, data, flushes):
        # reference src/buffer.rs:30: a second upload before the next flush would silently replace the first
        if flushes == self.previous:
            raise _ffi.RendererError("Wasteful call to buffer.set_data(). The previous data would be overridden.")
        self.previous = flushes
        data = np.ascontiguousarray(data, dtype=np.float32)
        nbytes = data.nbytes
        if nbytes > self.size:
            # reference src/buffer.rs:35-40,57-67: new buffer created pre-filled; the old one stays alive for
            # command buffers already recorded against it
            new_size = next_power_of_two(nbytes + HEADROOM)
            new_handle = self._create(new_size, data)
            _ffi.lib().rh_buffer_release(self.handle)
            self.handle, self.size = new_handle, new_size
            self.generation += 1
        elif nbytes:
            _ffi.check(_ffi.lib().rh_buffer_write(self.handle, 0, data.ctypes.data, nbytes))

    def __del__(self):
        try:
            if self.handle:
                _ffi.lib().rh_buffer_release(self.handle)
                self.handle = None
        except Exception:
            pass
