"""Clear colour (reference src/clear_color.rs:7-16): f32 components widened to f64."""
import numpy as np


class ClearColor:
    def __init__(self, red, green, blue, alpha):
        self.inner = tuple(float(np.float32(c)) for c in (red, green, blue, alpha))

    def __repr__(self):
        return f"ClearColor{self.inner}"
