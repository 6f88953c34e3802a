"""ctypes bindings for the CPU oracle libraries (TEST INFRASTRUCTURE ONLY).

raster: oracle/raster_oracle.cpp  -- software restatement of the draw path
lz4:    oracle/lz4_oracle.c       -- restatement of the LZ4 frame codec
liblz4: the system liblz4.so.1 (1.9.3 in this image), driven with the reference's preferences
        (reference src/compressor.rs:138-144) -- used to pin the lz4 restatement
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

FMT_R_U8, FMT_BGRA_U8, FMT_RGBA_U8, FMT_RGBA_F16, FMT_RGBA_F32 = range(5)
FILTER_LINEAR, FILTER_NEAREST = 0, 1
TOPO_TRIANGLE_LIST, TOPO_TRIANGLE_STRIP = 0, 1
BF_ZERO, BF_ONE, BF_SRC_ALPHA, BF_ONE_MINUS_SRC_ALPHA = range(4)
SHADER_HELLO, SHADER_SIZED, SHADER_HELLO_LAYER, SHADER_TILTED, SHADER_SPACE, SHADER_CROWD = 0, 1, 2, 3, 4, 5


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


class OrcTexture(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("layers", C.c_uint32),
                ("format", C.c_uint32), ("filter", C.c_uint32)]


class OrcTarget(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32),
                ("clear", C.c_int32), ("clear_rgba", C.c_double * 4)]


class OrcDraw(C.Structure):
    _fields_ = [("shader", C.c_int32), ("topology", C.c_int32), ("vertex_count", C.c_uint32),
                ("instance_count", C.c_uint32), ("attr", C.c_void_p * 8), ("attr_size", C.c_uint32 * 8),
                ("storage", C.c_void_p * 4), ("uniform", C.c_void_p * 8), ("tex", OrcTexture * 8),
                ("has_viewport", C.c_int32), ("viewport", C.c_float * 4), ("clip_to_viewport", C.c_int32),
                ("src_factor", C.c_uint32), ("dst_factor", C.c_uint32), ("n_targets", C.c_uint32),
                ("targets", OrcTarget * 4), ("threads", C.c_int32), ("fragments_out", C.c_uint64),
                ("samples", C.c_uint32), ("msaa", C.c_void_p)]


_raster = None
_lz4 = None
_liblz4 = None


def raster_lib():
    global _raster
    if _raster is None:
        name = "libraster_oracle.so" if _cpu_has_fma() else "libraster_oracle_generic.so"
        path = os.path.join(HERE, name)
        if not os.path.exists(path):
            build()
        _raster = C.CDLL(path)
        _raster.orc_render.argtypes = [C.POINTER(OrcDraw)]
        _raster.orc_render.restype = C.c_int
    return _raster


def render(shader, topology, vertex_count, instance_count, attrs, storage, textures, targets, blend,
           viewport=None, clip_to_viewport=True, uniforms=(), threads=0, samples=1, msaa=None):
    """Run the oracle.

    attrs:    list of (np.float32 array, size)
    storage:  list of np.float32 arrays
    textures: list of (np.uint8 array [layers*H*W*C], width, height, layers, format, filter)
    targets:  list of dicts {data: np.uint8 [H, W, bytes per texel] (modified in place), format, clear: None | (r, g, b, a)}
    blend:    (src_factor, dst_factor)
    samples:  1, or 2/4/8 with msaa = np.uint8 [H, W, samples, bytes per texel of the target] (the pipeline's multisampled texture, modified in
              place and resolved into the single target)
    Returns the number of fragments blended per attachment.
    """
    d = OrcDraw()
    keep = []
    d.shader, d.topology, d.vertex_count, d.instance_count = shader, topology, vertex_count, instance_count
    for i, (a, size) in enumerate(attrs):
        a = np.ascontiguousarray(a, dtype=np.float32); keep.append(a)
        d.attr[i] = a.ctypes.data; d.attr_size[i] = size
    for i, s in enumerate(storage):
        s = np.ascontiguousarray(s, dtype=np.float32); keep.append(s)
        d.storage[i] = s.ctypes.data
    for i, u in enumerate(uniforms):
        u = np.ascontiguousarray(u, dtype=np.float32); keep.append(u)
        d.uniform[i] = u.ctypes.data
    for i, (data, w, h, layers, fmt, flt) in enumerate(textures):
        data = np.ascontiguousarray(data)
        data = data.view(np.uint8) if data.dtype != np.uint8 else data      # float texels travel as raw bytes
        keep.append(data)
        d.tex[i] = OrcTexture(data.ctypes.data, w, h, layers, fmt, flt)
    if viewport is not None:
        d.has_viewport = 1
        d.viewport[:] = [float(v) for v in viewport]
    d.clip_to_viewport = 1 if clip_to_viewport else 0
    d.src_factor, d.dst_factor = blend
    d.n_targets = len(targets)
    for i, t in enumerate(targets):
        buf = t["data"]
        bpt = {FMT_R_U8: 1, FMT_RGBA_F16: 8, FMT_RGBA_F32: 16}.get(t.get("format", FMT_RGBA_U8), 4)
        assert buf.dtype == np.uint8 and buf.flags["C_CONTIGUOUS"] and buf.ndim == 3 and buf.shape[2] == bpt
        d.targets[i].data = buf.ctypes.data
        d.targets[i].height, d.targets[i].width = buf.shape[0], buf.shape[1]
        d.targets[i].format = t.get("format", FMT_RGBA_U8)
        clear = t.get("clear")
        d.targets[i].clear = 0 if clear is None else 1
        if clear is not None:
            # reference src/clear_color.rs:10: f32 components widened to f64
            d.targets[i].clear_rgba[:] = [float(np.float32(c)) for c in clear]
    d.threads = threads
    if samples > 1:
        assert msaa is not None and msaa.dtype == np.uint8 and msaa.flags["C_CONTIGUOUS"]
        assert msaa.shape == (targets[0]["data"].shape[0], targets[0]["data"].shape[1], samples, targets[0]["data"].shape[2])
        d.samples, d.msaa = samples, msaa.ctypes.data
    rc = raster_lib().orc_render(C.byref(d))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return int(d.fragments_out)


# ---- LZ4 restatement ------------------------------------------------------------------------------------------------
def lz4_lib():
    global _lz4
    if _lz4 is None:
        path = os.path.join(HERE, "liblz4_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.lz4o_create.restype = C.c_void_p
        L.lz4o_destroy.argtypes = [C.c_void_p]
        L.lz4o_bound.argtypes = [C.c_size_t]; L.lz4o_bound.restype = C.c_size_t
        L.lz4o_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.lz4o_begin.restype = C.c_size_t
        L.lz4o_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.lz4o_update.restype = C.c_size_t
        L.lz4o_end.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.lz4o_end.restype = C.c_size_t
        L.lz4o_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.lz4o_decompress_frame.restype = C.c_int64
        _lz4 = L
    return _lz4


class Lz4OracleStream:
    """One LZ4F stream; write(b) = one Write::write_all in the reference (src/compressor.rs:126-129)."""

    def __init__(self):
        self.L = lz4_lib()
        self.h = self.L.lz4o_create()
        self.out = bytearray()
        buf = (C.c_uint8 * 16)()
        n = self.L.lz4o_begin(self.h, buf, 16)
        self.out += bytes(buf[:n])

    def write(self, data):
        data = bytes(data)
        if not data:
            return
        src = (C.c_uint8 * len(data)).from_buffer_copy(data)
        cap = self.L.lz4o_bound(len(data))
        dst = (C.c_uint8 * cap)()
        n = self.L.lz4o_update(self.h, dst, cap, src, len(data))
        assert n > 0
        self.out += bytes(memoryview(dst)[:n])

    def finish(self):
        buf = (C.c_uint8 * 8)()
        n = self.L.lz4o_end(self.h, buf, 8)
        self.out += bytes(buf[:n])
        self.L.lz4o_destroy(self.h); self.h = None
        return bytes(self.out)


def lz4_decompress_frame(stream, cap):
    L = lz4_lib()
    src = (C.c_uint8 * max(1, len(stream))).from_buffer_copy(bytes(stream) or b"\0")
    dst = (C.c_uint8 * max(1, cap))()
    consumed = C.c_size_t(0); complete = C.c_int(0)
    n = L.lz4o_decompress_frame(src, len(stream), dst, cap, C.byref(consumed), C.byref(complete))
    if n < 0:
        raise ValueError("malformed LZ4 frame")
    return bytes(memoryview(dst)[:n]), consumed.value, bool(complete.value)


# ---- system liblz4 driven with the reference's preferences ----------------------------------------------------------------
class _Lz4fPreferences(C.Structure):
    # LZ4F_preferences_t (lz4frame.h, 1.9.x): frameInfo {blockSizeID, blockMode, contentChecksumFlag, frameType,
    # contentSize, dictID, blockChecksumFlag}, compressionLevel, autoFlush, favorDecSpeed, reserved[3]
    _fields_ = [("blockSizeID", C.c_uint32), ("blockMode", C.c_uint32), ("contentChecksumFlag", C.c_uint32),
                ("frameType", C.c_uint32), ("contentSize", C.c_uint64), ("dictID", C.c_uint32),
                ("blockChecksumFlag", C.c_uint32), ("compressionLevel", C.c_int32), ("autoFlush", C.c_uint32),
                ("favorDecSpeed", C.c_uint32), ("reserved", C.c_uint32 * 3)]


def liblz4():
    global _liblz4
    if _liblz4 is None:
        L = C.CDLL("liblz4.so.1")
        L.LZ4_versionNumber.restype = C.c_int
        L.LZ4F_createCompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        L.LZ4F_createCompressionContext.restype = C.c_size_t
        L.LZ4F_freeCompressionContext.argtypes = [C.c_void_p]
        L.LZ4F_compressBegin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.LZ4F_compressBegin.restype = C.c_size_t
        L.LZ4F_compressBound.argtypes = [C.c_size_t, C.c_void_p]; L.LZ4F_compressBound.restype = C.c_size_t
        L.LZ4F_compressUpdate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.LZ4F_compressUpdate.restype = C.c_size_t
        L.LZ4F_compressEnd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.LZ4F_compressEnd.restype = C.c_size_t
        L.LZ4F_isError.argtypes = [C.c_size_t]; L.LZ4F_isError.restype = C.c_uint
        _liblz4 = L
    return _liblz4


class LibLz4Stream:
    """liblz4's LZ4F streaming compressor with the reference's preferences: level = arg, favor_dec_speed off,
    auto_flush on, everything else default (reference src/compressor.rs:138-144)."""

    def __init__(self, level=0):
        self.L = liblz4()
        self.ctx = C.c_void_p()
        err = self.L.LZ4F_createCompressionContext(C.byref(self.ctx), 100)
        assert not self.L.LZ4F_isError(err)
        self.prefs = _Lz4fPreferences()
        self.prefs.compressionLevel = level
        self.prefs.autoFlush = 1
        self.out = bytearray()
        buf = (C.c_uint8 * 32)()
        n = self.L.LZ4F_compressBegin(self.ctx, buf, 32, C.byref(self.prefs))
        assert not self.L.LZ4F_isError(n)
        self.out += bytes(buf[:n])

    def write(self, data):
        data = bytes(data)
        if not data:
            return
        src = (C.c_uint8 * len(data)).from_buffer_copy(data)
        cap = self.L.LZ4F_compressBound(len(data), C.byref(self.prefs))
        dst = (C.c_uint8 * cap)()
        n = self.L.LZ4F_compressUpdate(self.ctx, dst, cap, src, len(data), None)
        assert not self.L.LZ4F_isError(n)
        self.out += bytes(memoryview(dst)[:n])

    def finish(self):
        buf = (C.c_uint8 * 32)()
        n = self.L.LZ4F_compressEnd(self.ctx, buf, 32, None)
        assert not self.L.LZ4F_isError(n)
        self.out += bytes(buf[:n])
        self.L.LZ4F_freeCompressionContext(self.ctx); self.ctx = None
        return bytes(self.out)
