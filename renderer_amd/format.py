"""Texel formats (reference src/format.rs:3-43)."""
import enum


class Format(enum.IntEnum):
    # declaration order of the reference enum = bincode discriminant
    RU8 = 0
    BgraU8 = 1
    RgbaU8 = 2
    RgbaF16 = 3
    RgbaF32 = 4

    def channels(self):
        return 1 if self is Format.RU8 else 4

    def bytes_per_channel(self):
        return {Format.RU8: 1, Format.BgraU8: 1, Format.RgbaU8: 1, Format.RgbaF16: 2, Format.RgbaF32: 4}[self]

    def bytes_per_texel(self):
        return self.channels() * self.bytes_per_channel()

    @staticmethod
    def default():
        return Format.RgbaU8
