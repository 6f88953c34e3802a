"""Blend modes (reference src/blend_mode.rs:8-40): the same factors for colour and alpha, op Add, mask ALL."""
from . import _ffi


class BlendMode:
    def __init__(self, src_factor, dst_factor):
        self.src_factor, self.dst_factor = src_factor, dst_factor

    @staticmethod
    def additive():
        return BlendMode(_ffi.BLEND_ONE, _ffi.BLEND_ONE)

    @staticmethod
    def pre_multiplied_alpha():
        return BlendMode(_ffi.BLEND_ONE, _ffi.BLEND_ONE_MINUS_SRC_ALPHA)

    def state(self, target_format):
        """One colour target state: (format, src_factor, dst_factor)."""
        return _ffi.ColorTargetDesc(int(target_format), self.src_factor, self.dst_factor)
