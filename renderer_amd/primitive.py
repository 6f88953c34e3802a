"""Primitive topology (reference src/primitive.rs:2-14)."""
import enum


class Primitive(enum.IntEnum):
    Triangle = 0        # wgpu::PrimitiveTopology::TriangleList
    TriangleStrip = 1   # wgpu::PrimitiveTopology::TriangleStrip
