"""Shader-stage visibility of a binding (reference src/visibility.rs:2-15)."""
import enum


class Visibility(enum.IntEnum):
    VertexShader = 1
    FragmentShader = 2
    BothShaders = 3

    def shader_stage(self):
        return int(self)
