"""2D / 2D-array texture with one mip level and an optional clamp-to-edge sampler (reference src/texture.rs:23-158)."""
import ctypes as C

import numpy as np

from . import _ffi
from .format import Format


class Texture:
    def __init__(self, device, size, filter_mode, format, msaa_samples, renderable, copyable, with_sampler):
        self.device = device
        self.size = tuple(int(s) for s in size)      # (width, height, layers)
        self.filter_mode = filter_mode
        self.format = Format(format)
        self.msaa_samples = msaa_samples
        self.renderable = renderable
        self.copyable = copyable
        self.generation = 0
        self.handle = self._create_texture()
        self.sampler = self._create_sampler() if with_sampler else None

    def _usage(self):
        usage = _ffi.TEXTURE_BINDING | _ffi.TEXTURE_COPY_DST            # reference src/texture.rs:111
        if self.renderable:
            usage |= _ffi.TEXTURE_RENDER_ATTACHMENT
        if self.copyable:
            usage |= _ffi.TEXTURE_COPY_SRC
        return usage

    def _create_texture(self):
        out = C.c_void_p()
        w, h, layers = self.size
        _ffi.check(_ffi.lib().rh_texture_create(self.device, w, h, layers, int(self.format), self.msaa_samples,
                                                self._usage(), C.byref(out)))
        return out

    def _create_sampler(self):
        out = C.c_void_p()
        _ffi.check(_ffi.lib().rh_sampler_create(self.device, int(self.filter_mode), C.byref(out)))
        return out

    def resize(self, new_size):
        """Reallocates (contents are lost) and bumps the generation, only when width or height change
        (reference src/texture.rs:34-43)."""
        new_size = tuple(int(s) for s in new_size)
        if self.size[0] == new_size[0] and self.size[1] == new_size[1]:
            return
        if 0 in new_size:
            return
        self.size = new_size
        old = self.handle
        self.handle = self._create_texture()
        _ffi.lib().rh_texture_release(old)
        self.generation += 1

    def set_data(self, offset, size, data):
        """Uploads tightly packed rows into (x, y, layer); size (0, 0) means the whole layer
        (reference src/texture.rs:45-56)."""
        if tuple(size) == (0, 0):
            size = (self.size[0], self.size[1])
        data = np.ascontiguousarray(data)
        bytes_per_row = size[0] * self.format.bytes_per_texel()
        _ffi.check(_ffi.lib().rh_texture_write(self.handle, offset[0], offset[1], offset[2], size[0], size[1],
                                               data.ctypes.data, data.nbytes, bytes_per_row))

    def read(self, layer=0):
        """Synchronous read-back of one layer as an array [height, width, bytes_per_texel] of uint8
        ([height, width, samples, bytes_per_texel] for a multisampled texture)."""
        w, h, _ = self.size
        shape = (h, w, self.format.bytes_per_texel()) if self.msaa_samples == 1 else (h, w, self.msaa_samples, self.format.bytes_per_texel())
        out = np.empty(shape, dtype=np.uint8)
        _ffi.check(_ffi.lib().rh_texture_read(self.handle, layer, out.ctypes.data, out.nbytes))
        return out

    def texture_binding(self, visibility):
        return _ffi.SlotDesc(_ffi.SLOT_TEXTURE, visibility.shader_stage(), self.handle)

    def sampler_binding(self, visibility):
        return _ffi.SlotDesc(_ffi.SLOT_SAMPLER, visibility.shader_stage(), self.sampler)

    def __del__(self):
        try:
            if self.handle:
                _ffi.lib().rh_texture_release(self.handle)
                self.handle = None
            if self.sampler:
                _ffi.lib().rh_sampler_release(self.sampler)
                self.sampler = None
        except Exception:
            pass
