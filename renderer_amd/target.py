"""Render targets (reference src/target.rs:2-27).  Screen is an offscreen BGRA8 texture of the window size."""
from .format import Format


class Target:
    def __init__(self, texture=None):
        self.texture = texture          # None = Screen

    @staticmethod
    def Screen():
        return Target(None)

    @staticmethod
    def Texture(texture):
        return Target(texture)

    def is_screen(self):
        return self.texture is None

    def format(self):
        return Format.BgraU8 if self.is_screen() else self.texture.format

    def handle(self, renderer):
        """The texture rendered into (the reference's TextureView)."""
        return renderer.screen_handle() if self.is_screen() else self.texture.handle

    def size(self, window_size):
        return (window_size[0], window_size[1], 1) if self.is_screen() else self.texture.size
