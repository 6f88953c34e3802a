"""renderer_amd -- MI355X (gfx950) implementation of the `renderer` crate's vertex/fragment pipeline.

Host-side mirror of the reference's public API (reference src/lib.rs:23-59) over the C ABI of librenderer_hip.so.
"""
from ._ffi import RendererError
from .attribute import Attribute
from .blend_mode import BlendMode
from .buffer import Buffer
from .clear_color import ClearColor
from .compiler import Compiler
from .compressor import Compressor
from .decompressor import Decompressor
from .ffmpeg_pipe import FfmpegPipe
from .filter_mode import FilterMode
from .format import Format
from .instanced import Instanced
from .png_encoder import PngEncoder
from .pipeline import BINDINGS_PER_GROUP, Pipeline, RecordingPosition
from .primitive import Primitive
from .program import Program
from .render_pass import RenderPass
from .render_thread import RenderThread
from .renderer import Renderer
from .target import Target
from .texture import Texture
from .uniform import Uniform
from .video_frame import FrameStatus, ImageData, VideoFrame
from .video_recorder import VideoRecorder
from .viewport import Viewport
from .visibility import Visibility

__all__ = [n for n in dir() if n[0].isupper()]
