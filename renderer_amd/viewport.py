"""Letterbox viewport (reference src/viewport.rs:10-28); all arithmetic in f32, no rounding."""
import numpy as np

f32 = np.float32


class Viewport:
    def __init__(self, aspect_x, aspect_y, max_width, max_height):
        aspect_x, aspect_y, max_width, max_height = f32(aspect_x), f32(aspect_y), f32(max_width), f32(max_height)
        current_aspect = max_width / max_height
        desired_aspect = aspect_x / aspect_y
        width, height = max_width, max_height
        margin_x, margin_y = f32(0), f32(0)
        if current_aspect > desired_aspect:
            width = height * desired_aspect
            margin_x = (max_width - width) / f32(2)
        else:
            height = width / desired_aspect
            margin_y = (max_height - height) / f32(2)
        self.width, self.height, self.margin_x, self.margin_y = float(width), float(height), float(margin_x), float(margin_y)

    def __eq__(self, other):
        return isinstance(other, Viewport) and (self.width, self.height, self.margin_x, self.margin_y) == \
            (other.width, other.height, other.margin_x, other.margin_y)

    def __repr__(self):
        return f"Viewport(width={self.width}, height={self.height}, margin_x={self.margin_x}, margin_y={self.margin_y})"
