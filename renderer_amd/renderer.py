"""Facade: device bring-up, per-frame lifecycle, upload routing by (set, binding), recording on/off and the
enum/const factories (reference src/renderer.rs:26-378).  There is no window: `Renderer(width, height)` replaces
`Renderer::new(window)` and Target::Screen is an offscreen BGRA8 image of that size."""
import ctypes as C

import numpy as np

from . import _ffi
from .attribute import Attribute
from .blend_mode import BlendMode
from .clear_color import ClearColor
from .filter_mode import FilterMode
from .format import Format
from .instanced import Instanced
from .pipeline import BINDINGS_PER_GROUP, Pipeline, RecordingPosition
from .primitive import Primitive
from .program import Program
from .render_pass import RenderPass
from .target import Target
from .texture import Texture
from .uniform import Uniform
from .video_recorder import VideoRecorder
from .viewport import Viewport
from .visibility import Visibility


class Renderer:
    def __init__(self, width, height, hip_ordinal=0):
        self.window_size = (int(width), int(height))
        out = C.c_void_p()
        _ffi.check(_ffi.lib().rh_device_create(hip_ordinal, self.window_size[0], self.window_size[1], C.byref(out)))
        self.device = out
        self.vsync = True
        self.frame = True                # a "current surface texture" is held (reference src/renderer.rs:46)
        self.commands = []
        self.recorder = None
        self.flushes = 0

    # ---- surface ------------------------------------------------------------------------------------------------------
    def screen_handle(self):
        out = C.c_void_p()
        _ffi.check(_ffi.lib().rh_screen_texture(self.device, C.byref(out)))
        return out

    def resize_swap_chain(self, new_size):
        """reference src/renderer.rs:56-66"""
        if new_size[0] == 0 or new_size[1] == 0:
            return
        self.window_size = (int(new_size[0]), int(new_size[1]))
        self.frame = None
        _ffi.check(_ffi.lib().rh_screen_resize(self.device, self.window_size[0], self.window_size[1]))

    def resize_texture(self, texture, new_size):
        texture.resize(new_size)

    def read_screen(self):
        """The offscreen screen image as uint8 [height, width, 4] in BGRA byte order."""
        w, h = self.window_size
        out = np.empty((h, w, 4), dtype=np.uint8)
        _ffi.check(_ffi.lib().rh_texture_read(self.screen_handle(), 0, out.ctypes.data, out.nbytes))
        return out

    # ---- drawing ------------------------------------------------------------------------------------------------------
    def render(self, pipeline, clear_color, viewport, count):
        """count = (instance_count, vertices_per_instance) (reference src/renderer.rs:72-74)"""
        self.render_to(pipeline.targets, pipeline, clear_color, viewport, count)

    def render_to(self, targets, pipeline, clear_color, viewport, count):
        """Targets may differ from the pipeline's but their formats must match (reference src/renderer.rs:76-90).
        Nothing executes until flush()."""
        for target in targets:
            if target.is_screen():
                self._start_frame()
        cbuffer = RenderPass(self).render(targets, pipeline, clear_color, viewport, count)
        self.commands.append(cbuffer)

    def _start_frame(self):
        if self.frame:
            return
        self.frame = True

    def finish_frame(self):
        """reference src/renderer.rs:102-117"""
        self.flush()
        if self.recorder is not None:
            self.recorder.initiate_buffer_mapping()
            self.recorder.process_mapped_buffers()
            self.recorder.finish_frame()
        if not self.frame:
            return
        self.frame = None                # present() is a no-op without a display

    def flush(self):
        """queue.submit of every recorded command buffer, in order (reference src/renderer.rs:119-122)."""
        n = len(self.commands)
        arr = (C.c_void_p * max(1, n))(*[c.value for c in self.commands])
        self.commands = []
        self.flushes += 1
        _ffi.check(_ffi.lib().rh_submit(self.device, arr, n))

    def flush_static(self):
        """flush() through rh_submit_static, the entry point that allocates nothing: the command buffers are not consumed
        by the call and are released here once the work has finished."""
        n = len(self.commands)
        arr = (C.c_void_p * max(1, n))(*[c.value for c in self.commands])
        commands, self.commands = self.commands, []
        self.flushes += 1
        try:
            _ffi.check(_ffi.lib().rh_submit_static(self.device, arr, n))
            self.synchronize()
        finally:
            for c in commands:
                _ffi.lib().rh_cmdbuf_release(c)

    def synchronize(self):
        _ffi.check(_ffi.lib().rh_device_synchronize(self.device))

    # ---- uploads ------------------------------------------------------------------------------------------------------
    def set_attribute(self, pipeline, location, data):
        attribute = next(a for a in pipeline.program.attributes if a.location == location)
        attribute.buffer.set_data(data, self.flushes)

    def set_instanced(self, pipeline, index_tuple, data):
        index = index_tuple[0] * BINDINGS_PER_GROUP + index_tuple[1]
        pipeline.program.instances[index].buffer.set_data(data, self.flushes)

    def set_uniform(self, pipeline, index_tuple, data):
        index = index_tuple[0] * BINDINGS_PER_GROUP + index_tuple[1]
        uniform, _ = pipeline.program.uniforms[uniform_index(index, pipeline.program)]
        uniform.buffer.set_data(data, self.flushes)

    def set_texture(self, pipeline, index_tuple, layers_data):
        for layer, data in enumerate(layers_data):
            self.set_part_of_texture(pipeline, index_tuple, (0, 0, layer), (0, 0), data)

    def set_part_of_texture(self, pipeline, index_tuple, offset, size, data):
        index = index_tuple[0] * BINDINGS_PER_GROUP + index_tuple[1]
        texture, _ = pipeline.program.textures[texture_index(index, pipeline.program)]
        texture.set_data(offset, size, data)

    def set_vsync(self, boolean):
        self.vsync = bool(boolean)
        self.frame = None

    def set_msaa_samples(self, pipeline, msaa_samples):
        pipeline.set_msaa_samples(msaa_samples)

    # ---- recording ----------------------------------------------------------------------------------------------------
    def start_recording(self, pipelines, clear_color, max_buffer_size_in_megabytes, process_function, keep_on_device=False):
        """reference src/renderer.rs:178-188.  keep_on_device=True hands frames to process_function without mapping
        them into host memory (for the GPU Compressor)."""
        max_size_in_bytes = int(np.float32(max_buffer_size_in_megabytes) * np.float32(1024.0) * np.float32(1024.0))
        self.recorder = VideoRecorder(self, clear_color, max_size_in_bytes, process_function, keep_on_device)
        for i, pipeline in enumerate(pipelines):
            is_last = i == len(pipelines) - 1
            pipeline.set_stream_position(RecordingPosition.Last if is_last else RecordingPosition.NotLast)

    def stop_recording(self, pipelines):
        """reference src/renderer.rs:190-197"""
        self.recorder = None
        for pipeline in pipelines:
            pipeline.set_stream_position(RecordingPosition.NONE)

    def adapter_info(self):
        buf = C.create_string_buffer(256)
        _ffi.check(_ffi.lib().rh_device_info(self.device, buf, 256))
        return buf.value.decode()

    def set_clip_to_viewport(self, enabled):
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_CLIP_TO_VIEWPORT, 1 if enabled else 0))

    def set_strip(self, row0, row1):
        """Sort-first split: this device renders (and clears) only pixel rows [row0, row1); (0, 0) disables."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_STRIP_ROW0, int(row0)))
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_STRIP_ROW1, int(row1)))

    def set_general_kernels(self, enabled):
        """Forces the kernel body that reads all pipeline state at run time (identical results; for testing)."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_GENERAL_KERNELS, 1 if enabled else 0))

    def set_binning_tickets(self, enabled):
        """Forces the form of the binning chain that launches too large for the device use (identical results; for testing)."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_BINNING_TICKETS, 1 if enabled else 0))

    def set_early_binning(self, mode):
        """Binning and set-up of a pass on a stream of their own, beside the tile kernel of the pass before: False / 0
        never, True / 1 (default) for passes with a long binning chain (room for four million pairs or more), 2 for
        every pass it is valid for (identical results whichever)."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_EARLY_BINNING, int(mode)))

    def set_tail_split(self, mode):
        """How the tiles of a pass's last, partly filled round of workgroups are cut into bands of rows: 0 never, 1 one
        kind of band per pass (halves, quarters or eighths), 2 (default) quarters beside the halves where there is room
        left, 3 thirds (11, 11 and 10 rows) where the tail fits the round in thirds but not in quarters (identical
        results)."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_TAIL_SPLIT, int(mode)))

    def set_wave_priority(self, enabled):
        """True (default): the tile kernel's waves set their issue priority by the work they have left; False: priorities
        are left alone (identical results)."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_WAVE_PRIORITY, 1 if enabled else 0))

    def set_setup_launch(self, enabled):
        """True: every set-up record comes from the set-up launch; False (default): the tile kernel sets lean quads up
        itself (identical results)."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_SETUP_LAUNCH, 1 if enabled else 0))

    def set_timing(self, mode):
        """HIP-event marks around the phases of every pass (True / 1), around the tile kernel only (2), or none."""
        _ffi.check(_ffi.lib().rh_device_set_option(self.device, _ffi.OPTION_TIMING, int(mode)))

    def drain_timings(self):
        t = _ffi.Timings()
        _ffi.check(_ffi.lib().rh_device_drain_timings(self.device, C.byref(t)))
        return {"passes": t.passes, "bin_ms": t.bin_ms, "setup_ms": t.setup_ms, "tile_ms": t.tile_ms}

    def stream(self):
        return _ffi.lib().rh_device_stream(self.device)

    def last_pass_stats(self):
        s = _ffi.PassStats()
        _ffi.check(_ffi.lib().rh_device_last_pass_stats(self.device, C.byref(s)))
        return {"clusters": s.clusters, "bin_entries": s.bin_entries, "tiles_x": s.tiles_x, "tiles_y": s.tiles_y,
                "tile_w": s.tile_w, "tile_h": s.tile_h, "reruns": s.reruns, "early": s.early}

    # ---- factories (reference src/renderer.rs:203-298) ----------------------------------------------------------------
    def pipeline(self, program, blend_mode, primitive, msaa_samples, targets):
        return Pipeline(self.device, self.window_size, program, blend_mode, primitive, msaa_samples, targets)

    def attribute(self, location, size):
        return Attribute(self.device, location, size)

    def instanced(self):
        return Instanced(self.device)

    def uniform(self):
        return Uniform(self.device)

    def texture(self, width, height, layers, filter_mode, format, renderable, copyable, with_sampler):
        return Texture(self.device, (width, height, layers), filter_mode, format, 1, renderable, copyable, with_sampler)

    def program(self, vert, frag, attributes, instances, uniforms, textures):
        return Program(self.device, vert, frag, attributes, instances, uniforms, textures)

    def viewport(self, aspect_x, aspect_y):
        return Viewport(aspect_x, aspect_y, float(self.window_size[0]), float(self.window_size[1]))

    screen_target = staticmethod(lambda: Target.Screen())
    texture_target = staticmethod(lambda texture: Target.Texture(texture))
    bgra_u8 = staticmethod(lambda: Format.BgraU8)
    red_u8 = staticmethod(lambda: Format.RU8)
    rgba_u8 = staticmethod(lambda: Format.RgbaU8)
    rgba_f16 = staticmethod(lambda: Format.RgbaF16)
    rgba_f32 = staticmethod(lambda: Format.RgbaF32)
    linear_filtering = staticmethod(lambda: FilterMode.Linear)
    nearest_filtering = staticmethod(lambda: FilterMode.Nearest)
    visible_to_vertex_shader = staticmethod(lambda: Visibility.VertexShader)
    visible_to_fragment_shader = staticmethod(lambda: Visibility.FragmentShader)
    visible_to_both_shaders = staticmethod(lambda: Visibility.BothShaders)
    additive_blend = staticmethod(lambda: BlendMode.additive())
    pre_multiplied_blend = staticmethod(lambda: BlendMode.pre_multiplied_alpha())
    triangle_primitive = staticmethod(lambda: Primitive.Triangle)
    triangle_strip_primitive = staticmethod(lambda: Primitive.TriangleStrip)
    clear_color = staticmethod(lambda red, green, blue, alpha: ClearColor(red, green, blue, alpha))

    def close(self):
        if self.device:
            for c in self.commands:
                _ffi.lib().rh_cmdbuf_release(c)
            self.commands = []
            self.recorder = None
            _ffi.lib().rh_device_destroy(self.device)
            self.device = None


def uniform_index(index, program):
    """reference src/renderer.rs:356-358"""
    return index - len(program.instances)


def texture_index(index, program):
    """Walks the textures skipping sampler slots (reference src/renderer.rs:360-378)."""
    remaining = index - len(program.instances) - len(program.uniforms)
    for i, (texture, _) in enumerate(program.textures):
        if remaining == 0:
            return i
        remaining -= 1
        if texture.sampler is not None:
            remaining -= 1
        if remaining < 0:
            raise _ffi.RendererError("Tried to get a texture but a sampler is in that slot.")
    raise _ffi.RendererError("Tried to a get a texture but nothing is in that slot.")
