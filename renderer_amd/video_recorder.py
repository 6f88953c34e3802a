"""Capture hooks (reference src/video_recorder.rs:25-188): an RGBA8 recording texture of the window size rendered
as an extra colour attachment, copied after the last recorded pipeline into a per-frame MAP_READ buffer under a byte
budget, mapped asynchronously and delivered to `process_function` in order."""
import ctypes as C
import math
from collections import deque

from . import _ffi
from .filter_mode import FilterMode
from .format import Format
from .texture import Texture
from .video_frame import AtomicUsize, FrameStatus, ImageData, VideoFrame


def capture_layout(width, height, bytes_per_texel):
    """(unpadded_bytes_per_row, padded_bytes_per_row, frame_size_in_bytes): rows padded to wgpu's 256-byte copy
    alignment (reference src/video_recorder.rs:73-79)."""
    unpadded_bytes_per_row = width * bytes_per_texel
    alignment = _ffi.COPY_BYTES_PER_ROW_ALIGNMENT
    row_padding = (alignment - unpadded_bytes_per_row % alignment) % alignment
    padded_bytes_per_row = unpadded_bytes_per_row + row_padding
    return unpadded_bytes_per_row, padded_bytes_per_row, padded_bytes_per_row * height


class VideoRecorder:
    def __init__(self, renderer, clear_color, max_buffer_size_in_bytes, process_function, keep_on_device=False):
        self.device = renderer.device
        # keep_on_device: frames are delivered without mapping them into host memory (for the GPU compressor, which
        # reads the capture buffer on the device; only compressed bytes are read back)
        self.keep_on_device = keep_on_device
        self.max_buffer_size_in_bytes = max_buffer_size_in_bytes
        self.process_function = process_function
        size = (renderer.window_size[0], renderer.window_size[1], 1)
        # reference src/video_recorder.rs:179-188: RGBA8, renderable, copyable, no sampler
        self.recording_texture = Texture(self.device, size, FilterMode.Nearest, Format.RgbaU8, 1, True, True, False)
        self.clear_color = clear_color
        self.cleared_this_frame = False
        self.buffer_size_in_bytes = AtomicUsize(0)
        self.video_frames = deque()
        self.frame_states = deque()      # 0 = dropped, 1 = mapping, 2 = mapped, 3 = failed to map
        self.frame_number = 0
        self._callbacks = {}             # keeps ctypes callbacks alive until they fire

    def color_attachment(self):
        """Cleared on first use per frame only (reference src/video_recorder.rs:43-60)."""
        a = _ffi.AttachmentDesc()
        a.texture = self.recording_texture.handle
        if self.cleared_this_frame or self.clear_color is None:
            a.load_op = _ffi.LOAD_OP_LOAD
        else:
            self.cleared_this_frame = True
            a.load_op = _ffi.LOAD_OP_CLEAR
            a.clear[:] = self.clear_color.inner
        return a

    def finish_frame(self):
        self.cleared_this_frame = False

    def create_buffer_if_within_memory_limit(self, viewport):
        """reference src/video_recorder.rs:66-105"""
        tex = self.recording_texture
        width = math.floor(viewport.width) if viewport is not None else tex.size[0]
        height = math.floor(viewport.height) if viewport is not None else tex.size[1]
        fmt = tex.format
        unpadded_bytes_per_row, padded_bytes_per_row, frame_size_in_bytes = capture_layout(width, height, fmt.bytes_per_texel())

        prev_size = self.buffer_size_in_bytes.fetch_add(frame_size_in_bytes)
        drop_frame = prev_size > self.max_buffer_size_in_bytes
        buffer = None
        if drop_frame:
            self.buffer_size_in_bytes.fetch_sub(frame_size_in_bytes)
        else:
            out = C.c_void_p()
            _ffi.check(_ffi.lib().rh_buffer_create(self.device, frame_size_in_bytes,
                                                   _ffi.BUFFER_COPY_DST | _ffi.BUFFER_MAP_READ, None, 0, C.byref(out)))
            buffer = out
        # the frame number is incremented regardless of whether the frame is dropped
        self.frame_number += 1
        status = FrameStatus.Dropped if drop_frame else FrameStatus.Captured
        image_data = ImageData.Buffer(buffer) if buffer is not None else None
        self.video_frames.append(VideoFrame(status, image_data, width, height, fmt, unpadded_bytes_per_row,
                                            padded_bytes_per_row, self.frame_number, frame_size_in_bytes,
                                            self.buffer_size_in_bytes))

    def copy_texture_to_buffer_if_present(self, encoder, viewport):
        """reference src/video_recorder.rs:107-128"""
        video_frame = self.video_frames[-1]
        if video_frame.image_data is None:
            return
        margin_x = math.ceil(viewport.margin_x) if viewport is not None else 0
        margin_y = math.ceil(viewport.margin_y) if viewport is not None else 0
        tex = self.recording_texture
        extent_w = tex.size[0] - 2 * margin_x
        extent_h = tex.size[1] - 2 * margin_y
        if extent_w < 0 or extent_h < 0:
            # u32 arithmetic in the reference (src/video_recorder.rs:124-125): a viewport thinner than one pixel panics
            raise _ffi.RendererError("attempt to subtract with overflow")
        _ffi.check(_ffi.lib().rh_cmd_copy_texture_to_buffer(
            encoder, tex.handle, margin_x, margin_y, video_frame.image_data.buffer(),
            video_frame.padded_bytes_per_row, video_frame.height, extent_w, extent_h))

    def initiate_buffer_mapping(self):
        """reference src/video_recorder.rs:130-150"""
        for i in range(len(self.video_frames)):
            if i < len(self.frame_states):
                continue
            video_frame = self.video_frames[i]
            if video_frame.image_data is not None and self.keep_on_device:
                self.frame_states.append([2])                 # usable as is: the consumer reads device memory
            elif video_frame.image_data is not None:
                state = [1]                                   # 1 = mapping
                key = id(state)

                def on_mapped(_user, ok, state=state):
                    state[0] = 2 if ok else 3                 # 2 = mapped, 3 = failed to map

                cb = _ffi.MAP_CALLBACK(on_mapped)
                self._callbacks[key] = (cb, state)
                _ffi.check(_ffi.lib().rh_buffer_map_async(video_frame.image_data.buffer(), cb, None))
                self.frame_states.append(state)
            else:
                self.frame_states.append([0])                 # 0 = dropped

    def process_mapped_buffers(self):
        """In-order delivery; stops at the first frame that is still mapping (reference src/video_recorder.rs:152-176)."""
        _ffi.check(_ffi.lib().rh_poll(self.device, 0))
        for key in [k for k, (_cb, st) in self._callbacks.items() if st[0] != 1]:
            del self._callbacks[key]                          # fired callbacks are no longer needed
        while self.video_frames:
            state = self.frame_states[0][0]
            if state in (0, 2):
                video_frame = self.video_frames.popleft()
                self.frame_states.popleft()
                self.process_function(video_frame)
            elif state == 1:
                break
            else:
                raise _ffi.RendererError("Failed to memory map buffer data for a video frame.")
