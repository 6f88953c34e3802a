"""Per-vertex attribute: one tightly packed f32 x size vertex buffer at a shader location, offset 0
(reference src/attribute.rs:10-29)."""
from . import _ffi
from .buffer import Buffer


class Attribute:
    def __init__(self, device, location, size):
        if size not in (1, 2, 3, 4):
            raise _ffi.RendererError("Unsupported attribute size")     # reference src/attribute.rs:25
        self.buffer = Buffer(device, _ffi.BUFFER_VERTEX | _ffi.BUFFER_COPY_DST)
        self.location = location
        self.size = size
