"""Uniform buffer with per-binding stage visibility, bound whole (reference src/uniform.rs:9-38)."""
from . import _ffi
from .buffer import Buffer


class Uniform:
    def __init__(self, device):
        self.buffer = Buffer(device, _ffi.BUFFER_UNIFORM | _ffi.BUFFER_COPY_DST)

    def binding(self, visibility):
        return _ffi.SlotDesc(_ffi.SLOT_UNIFORM, visibility.shader_stage(), self.buffer.handle)
