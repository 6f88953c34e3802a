"""ctypes binding of librenderer_hip.so (the C ABI in include/renderer_hip.h).

The library is built in-tree by renderer_amd.build (hipcc, gfx950).  There is no CPU fallback: if the library is
missing or a call fails, an exception is raised (the reference panics on the corresponding wgpu failures).
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librenderer_hip.so")

FORMAT_R_U8, FORMAT_BGRA_U8, FORMAT_RGBA_U8, FORMAT_RGBA_F16, FORMAT_RGBA_F32 = range(5)
FILTER_LINEAR, FILTER_NEAREST = 0, 1
TOPOLOGY_TRIANGLE_LIST, TOPOLOGY_TRIANGLE_STRIP = 0, 1
BLEND_ZERO, BLEND_ONE, BLEND_SRC_ALPHA, BLEND_ONE_MINUS_SRC_ALPHA = range(4)
LOAD_OP_LOAD, LOAD_OP_CLEAR = 0, 1
SLOT_STORAGE, SLOT_UNIFORM, SLOT_TEXTURE, SLOT_SAMPLER = range(4)
STAGE_VERTEX, STAGE_FRAGMENT = 1, 2
BUFFER_VERTEX, BUFFER_STORAGE, BUFFER_UNIFORM, BUFFER_COPY_DST, BUFFER_MAP_READ = 1, 2, 4, 8, 16
TEXTURE_BINDING, TEXTURE_COPY_DST, TEXTURE_RENDER_ATTACHMENT, TEXTURE_COPY_SRC = 1, 2, 4, 8
OPTION_CLIP_TO_VIEWPORT = 0
OPTION_TIMING = 1
OPTION_STRIP_ROW0, OPTION_STRIP_ROW1 = 2, 3
OPTION_GENERAL_KERNELS = 4
OPTION_BINNING_TICKETS = 5
OPTION_EARLY_BINNING = 6
OPTION_TAIL_SPLIT = 7
OPTION_EAGER_COUNT = 8
OPTION_SETUP_LAUNCH = 9
OPTION_WAVE_PRIORITY = 10
COPY_BYTES_PER_ROW_ALIGNMENT = 256
MAX_COLOR_TARGETS = 4


class RendererError(RuntimeError):
    """Raised where the reference would panic."""


class AttributeDesc(C.Structure):
    _fields_ = [("location", C.c_uint32), ("size", C.c_uint32)]


class SlotDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("visibility", C.c_uint32), ("resource", C.c_void_p)]


class ColorTargetDesc(C.Structure):
    _fields_ = [("format", C.c_uint32), ("src_factor", C.c_uint32), ("dst_factor", C.c_uint32)]


class PipelineDesc(C.Structure):
    _fields_ = [("vertex_module", C.c_void_p), ("fragment_module", C.c_void_p),
                ("n_attributes", C.c_uint32), ("attributes", C.POINTER(AttributeDesc)),
                ("n_slots", C.c_uint32), ("slots", C.POINTER(SlotDesc)),
                ("topology", C.c_uint32), ("samples", C.c_uint32),
                ("n_targets", C.c_uint32), ("targets", C.POINTER(ColorTargetDesc))]


class AttachmentDesc(C.Structure):
    _fields_ = [("texture", C.c_void_p), ("load_op", C.c_uint32), ("clear", C.c_double * 4), ("resolve_target", C.c_void_p)]


class PassDesc(C.Structure):
    _fields_ = [("pipeline", C.c_void_p), ("n_attachments", C.c_uint32), ("attachments", C.POINTER(AttachmentDesc)),
                ("n_vertex_buffers", C.c_uint32), ("vertex_buffers", C.POINTER(C.c_void_p)),
                ("has_viewport", C.c_int32), ("viewport", C.c_float * 6),
                ("vertex_count", C.c_uint32), ("instance_count", C.c_uint32)]


class PassStats(C.Structure):
    _fields_ = [("clusters", C.c_uint64), ("bin_entries", C.c_uint64), ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("reruns", C.c_uint64), ("early", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [("passes", C.c_uint32), ("bin_ms", C.c_double), ("tile_ms", C.c_double), ("setup_ms", C.c_double)]


class CompressorStats(C.Structure):
    _fields_ = [("rounds", C.c_uint64), ("jobs", C.c_uint64), ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64),
                ("kernel_ms", C.c_double)]


MAP_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_int)

# name -> (restype, argtypes); every symbol declared in include/renderer_hip.h
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "rh_last_error": (C.c_char_p, []),
    "rh_abi_version": (C.c_int, []),
    "rh_device_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, _PP]),
    "rh_device_destroy": (None, [_P]),
    "rh_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "rh_device_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "rh_device_synchronize": (C.c_int, [_P]),
    "rh_device_stream": (C.c_void_p, [_P]),
    "rh_screen_texture": (C.c_int, [_P, _PP]),
    "rh_screen_resize": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "rh_buffer_create": (C.c_int, [_P, C.c_size_t, C.c_uint32, _P, C.c_size_t, _PP]),
    "rh_buffer_write": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "rh_buffer_size": (C.c_size_t, [_P]),
    "rh_buffer_device_ptr": (C.c_void_p, [_P]),
    "rh_buffer_retain": (None, [_P]),
    "rh_buffer_release": (None, [_P]),
    "rh_buffer_map_async": (C.c_int, [_P, MAP_CALLBACK, _P]),
    "rh_buffer_mapped_ptr": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    "rh_buffer_unmap": (C.c_int, [_P]),
    "rh_poll": (C.c_int, [_P, C.c_int]),
    "rh_texture_create": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _PP]),
    "rh_texture_write": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_size_t, C.c_uint32]),
    "rh_texture_read": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t]),
    "rh_texture_size": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rh_texture_device_ptr": (C.c_void_p, [_P]),
    "rh_texture_retain": (None, [_P]),
    "rh_texture_release": (None, [_P]),
    "rh_sampler_create": (C.c_int, [_P, C.c_uint32, _PP]),
    "rh_sampler_release": (None, [_P]),
    "rh_module_load": (C.c_int, [_P, _P, C.c_size_t, _PP]),
    "rh_module_retain": (None, [_P]),
    "rh_module_release": (None, [_P]),
    "rh_pipeline_create": (C.c_int, [_P, C.POINTER(PipelineDesc), _PP]),
    "rh_pipeline_retain": (None, [_P]),
    "rh_pipeline_release": (None, [_P]),
    "rh_cmd_begin": (C.c_int, [_P, _PP]),
    "rh_cmd_render_pass": (C.c_int, [_P, C.POINTER(PassDesc)]),
    "rh_cmd_copy_texture_to_buffer": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rh_cmd_finish": (C.c_int, [_P, _PP]),
    "rh_cmdbuf_release": (None, [_P]),
    "rh_submit": (C.c_int, [_P, _PP, C.c_size_t]),
    "rh_submit_static": (C.c_int, [_P, _PP, C.c_size_t]),
    "rh_device_last_pass_stats": (C.c_int, [_P, C.POINTER(PassStats)]),
    "rh_device_drain_timings": (C.c_int, [_P, C.POINTER(Timings)]),
    "rh_compressor_create": (C.c_int, [_P, C.c_uint32, C.c_int, _PP]),
    "rh_compressor_destroy": (None, [_P]),
    "rh_compressor_write_packet": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t, _P]),
    "rh_compressor_submit": (C.c_int, [_P]),
    "rh_compressor_flush": (C.c_int, [_P]),
    "rh_compressor_wait": (C.c_int, [_P]),
    "rh_compressor_peek": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rh_compressor_consume": (C.c_int, [_P, C.c_uint32, C.c_size_t]),
    "rh_compressor_pending": (C.c_size_t, [_P, C.c_uint32]),
    "rh_compressor_read": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rh_compressor_finish_stream": (C.c_int, [_P, C.c_uint32]),
    "rh_compressor_drain_stats": (C.c_int, [_P, _P]),
    "rh_png_bound": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "rh_png_encode": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rh_lz4f_decompress": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "rh_decoder_create": (C.c_int, [_P, C.c_uint32, _PP]),
    "rh_decoder_destroy": (None, [_P]),
    "rh_decoder_feed": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t]),
    "rh_decoder_run": (C.c_int, [_P]),
    "rh_decoder_peek": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rh_decoder_consume": (C.c_int, [_P, C.c_uint32, C.c_size_t]),
    "rh_decoder_stream_info": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}

_lib = None
# Whether PyTorch was already imported when the library was loaded: only then do the two share one HIP runtime, and
# only then can torch use memory the renderer allocated (partition.texture_as_tensor checks this).
torch_came_first = None


def lib():
    """Loads librenderer_hip.so, failing loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RendererError(f"{LIB_PATH} is missing: run `python -m renderer_amd.build` (needs hipcc) first; "
                                "there is no CPU fallback")
        # PyTorch bundles a HIP runtime with the system runtime's soname.  When torch is going to touch the GPU in this
        # process (torch.distributed over RCCL, tensors aliasing renderer memory) it has to be loaded before this library,
        # so that both share the one runtime; RH_HIP_RUNTIME=torch asks for that explicitly.
        if os.environ.get("RH_HIP_RUNTIME") == "torch":
            import torch  # noqa: F401
        global torch_came_first
        torch_came_first = "torch" in sys.modules
        L = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RendererError(lib().rh_last_error().decode("utf-8", "replace"))
