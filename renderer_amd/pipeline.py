"""Immutable draw state (reference src/pipeline.rs:30-237): binding slots, colour target states (+1 RGBA8 target
while recording), vertex buffer layouts, topology, sample count.  Rebuilt when any bound resource's generation
changes."""
import ctypes as C
import enum

from . import _ffi
from .format import Format


class RecordingPosition(enum.Enum):
    # Only the LAST pipeline of a recorded frame copies the recording texture to a buffer
    # (reference src/pipeline.rs:21-23).
    NONE = 0
    NotLast = 1
    Last = 2


# wgpu limited bind groups to 8 sets of 4 bindings (reference src/pipeline.rs:25-27)
BINDINGS_PER_GROUP = 4


class Pipeline:
    def __init__(self, device, window_size, program, blend_mode, primitive, msaa_samples, targets):
        self.device = device
        self.program = program
        self.blend_mode = blend_mode
        self.primitive = primitive
        self.msaa_samples = msaa_samples
        self.msaa_texture = None
        self.position_in_recording = RecordingPosition.NONE
        self.targets = list(targets)
        self.window_size = tuple(window_size)
        self.handle = None
        if msaa_samples > 1:
            self.msaa_texture = self._create_msaa_texture(msaa_samples)
        self._rebuild()
        self.seen_generations = program.latest_generations()

    # -- reference src/pipeline.rs:164-187 ---------------------------------------------------------------------------
    def _create_msaa_texture(self, msaa_samples):
        """One multisampled texture per pipeline, shaped after the FIRST target; every pass of the pipeline renders
        into it (clear and load act on it) and resolves into the target."""
        from .texture import Texture
        from .filter_mode import FilterMode
        target = self.targets[0]
        return Texture(self.device, target.size(self.window_size), FilterMode.Nearest, target.format(), msaa_samples,
                       True, False, False)

    def _resize_msaa_texture(self, window_size, targets):
        if self.msaa_texture is not None:
            self.msaa_texture.resize(targets[0].size(window_size))

    # -- reference src/pipeline.rs:90-131 ---------------------------------------------------------------------------
    def create_bind_groups(self):
        """Flat slot list: instanced..., uniforms..., then per texture: texture [, sampler].
        Slot i is (set i // 4, binding i % 4)."""
        slots = []
        for instanced in self.program.instances:
            slots.append(instanced.binding())
        for uniform, visibility in self.program.uniforms:
            slots.append(uniform.binding(visibility))
        for texture, visibility in self.program.textures:
            slots.append(texture.texture_binding(visibility))
            if texture.sampler is not None:
                slots.append(texture.sampler_binding(visibility))
        return slots

    # -- reference src/pipeline.rs:133-142 ---------------------------------------------------------------------------
    def create_color_target_states(self):
        states = [self.blend_mode.state(t.format()) for t in self.targets]
        if self.position_in_recording is not RecordingPosition.NONE:
            states.append(self.blend_mode.state(Format.RgbaU8))
        return states

    # -- reference src/pipeline.rs:144-162,197-237 --------------------------------------------------------------------
    def _rebuild(self):
        slots = self.create_bind_groups()
        states = self.create_color_target_states()
        attrs = [_ffi.AttributeDesc(a.location, a.size) for a in self.program.attributes]
        desc = _ffi.PipelineDesc()
        desc.vertex_module = self.program.vertex_shader
        desc.fragment_module = self.program.fragment_shader
        attr_arr = (_ffi.AttributeDesc * max(1, len(attrs)))(*attrs)
        slot_arr = (_ffi.SlotDesc * max(1, len(slots)))(*slots)
        state_arr = (_ffi.ColorTargetDesc * max(1, len(states)))(*states)
        desc.n_attributes, desc.attributes = len(attrs), attr_arr
        desc.n_slots, desc.slots = len(slots), slot_arr
        desc.topology = int(self.primitive)
        desc.samples = self.msaa_samples
        desc.n_targets, desc.targets = len(states), state_arr
        out = C.c_void_p()
        _ffi.check(_ffi.lib().rh_pipeline_create(self.device, C.byref(desc), C.byref(out)))
        if self.handle:
            _ffi.lib().rh_pipeline_release(self.handle)
        self.handle = out
        # the vertex buffers bound at draw time, slot order = attribute list order (reference src/render_pass.rs:35-37)

    def recreate_on_buffer_or_texture_resize(self, window_size, targets):
        """reference src/pipeline.rs:44-62"""
        if targets:
            self._resize_msaa_texture(window_size, targets)
        actual = self.program.latest_generations()
        if actual == self.seen_generations:
            return
        self._rebuild()
        self.window_size = tuple(window_size)
        self.seen_generations = actual

    def set_msaa_samples(self, msaa_samples):
        """reference src/pipeline.rs:64-76"""
        self.msaa_texture = self._create_msaa_texture(msaa_samples) if msaa_samples > 1 else None
        self.msaa_samples = msaa_samples
        self._rebuild()

    def set_stream_position(self, position_in_recording):
        """reference src/pipeline.rs:78-88"""
        self.position_in_recording = position_in_recording
        self._rebuild()

    def __del__(self):
        try:
            if self.handle:
                _ffi.lib().rh_pipeline_release(self.handle)
                self.handle = None
        except Exception:
            pass
