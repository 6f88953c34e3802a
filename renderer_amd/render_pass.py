"""Encodes exactly one render pass with one instanced draw; after the last recorded pipeline appends the capture
copy (reference src/render_pass.rs:13-82)."""
import ctypes as C

from . import _ffi
from .pipeline import RecordingPosition


class RenderPass:
    def __init__(self, renderer):
        self.renderer = renderer

    def render(self, targets, pipeline, clear, viewport, count):
        r = self.renderer
        window_size = r.window_size
        pipeline.recreate_on_buffer_or_texture_resize(window_size, targets)
        if r.recorder is not None:
            r.recorder.recording_texture.resize((window_size[0], window_size[1], 1))

        instance_count, vertices_per_instance = count
        encoder = C.c_void_p()
        _ffi.check(_ffi.lib().rh_cmd_begin(r.device, C.byref(encoder)))
        if targets:                                           # reference src/render_pass.rs:26
            attachments = self.color_attachments(targets, pipeline, clear)
            att_arr = (_ffi.AttachmentDesc * len(attachments))(*attachments)
            vbs = [a.buffer.handle for a in pipeline.program.attributes]
            vb_arr = (C.c_void_p * max(1, len(vbs)))(*[h.value for h in vbs])
            desc = _ffi.PassDesc()
            desc.pipeline = pipeline.handle
            desc.n_attachments, desc.attachments = len(attachments), att_arr
            desc.n_vertex_buffers, desc.vertex_buffers = len(vbs), vb_arr
            if viewport is not None:                          # reference src/render_pass.rs:39-41
                desc.has_viewport = 1
                desc.viewport[:] = [viewport.margin_x, viewport.margin_y, viewport.width, viewport.height, 0.0, 1.0]
            desc.vertex_count, desc.instance_count = vertices_per_instance, instance_count
            _ffi.check(_ffi.lib().rh_cmd_render_pass(encoder, C.byref(desc)))

            if pipeline.position_in_recording is RecordingPosition.Last:
                recorder = r.recorder
                recorder.create_buffer_if_within_memory_limit(viewport)
                recorder.copy_texture_to_buffer_if_present(encoder, viewport)
        cbuffer = C.c_void_p()
        _ffi.check(_ffi.lib().rh_cmd_finish(encoder, C.byref(cbuffer)))
        return cbuffer

    def color_attachments(self, targets, pipeline, clear):
        attachments = [self.color_attachment(t.handle(self.renderer), clear, pipeline) for t in targets]
        if pipeline.position_in_recording is not RecordingPosition.NONE:
            attachments.append(self.renderer.recorder.color_attachment())
        return attachments

    @staticmethod
    def color_attachment(texture_handle, clear, pipeline=None):
        a = _ffi.AttachmentDesc()
        if pipeline is not None and pipeline.msaa_samples != 1:
            # reference src/render_pass.rs:76-79: render into the pipeline's multisampled texture, resolve into the target
            a.texture = pipeline.msaa_texture.handle
            a.resolve_target = texture_handle
        else:
            a.texture = texture_handle
        if clear is not None:
            a.load_op = _ffi.LOAD_OP_CLEAR
            a.clear[:] = clear.inner
        else:
            a.load_op = _ffi.LOAD_OP_LOAD
        return a
