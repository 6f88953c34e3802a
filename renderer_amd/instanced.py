"""Per-instance data: a read-only storage buffer visible to the vertex stage, indexed by the instance index in
the shader, bound whole (reference src/instanced.rs:9-38)."""
from . import _ffi
from .buffer import Buffer


class Instanced:
    def __init__(self, device):
        self.buffer = Buffer(device, _ffi.BUFFER_STORAGE | _ffi.BUFFER_COPY_DST)

    def binding(self):
        return _ffi.SlotDesc(_ffi.SLOT_STORAGE, _ffi.STAGE_VERTEX, self.buffer.handle)
