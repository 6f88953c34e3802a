"""RGBA8 PNG from a VideoFrame, stripping the row padding (reference src/png_encoder.rs:6-38).  The byte stream is not
pinned by the reference (it depends on the `png` crate's deflate); what is pinned is the decoded pixels."""
import struct
import zlib


class PngEncoder:
    @staticmethod
    def encode_to_bytes(video_frame):
        """Returns ("ok", bytes) or ("err", message) like the reference's Result."""
        if video_frame.image_data is None:
            return ("err", "VideoFrame could not be written because image_data is None.")
        rows = []

        def collect(data):
            data = memoryview(data).cast("B") if not isinstance(data, (bytes, bytearray)) else data
            padded, unpadded = video_frame.padded_bytes_per_row, video_frame.unpadded_bytes_per_row
            for y in range(video_frame.height):
                rows.append(b"\x00" + bytes(data[y * padded:y * padded + unpadded]))       # filter type 0

        video_frame.image_data.bytes_fn(collect)

        def chunk(tag, payload):
            return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xffffffff)

        ihdr = struct.pack(">IIBBBBB", video_frame.width, video_frame.height, 8, 6, 0, 0, 0)     # 8-bit RGBA
        idat = zlib.compress(b"".join(rows), 1)
        return ("ok", b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b""))

    @staticmethod
    def encode_on_device(renderer, video_frame):
        """The same result type as encode_to_bytes for a frame whose image is a device buffer, encoded on the GPU
        (librenderer_hip.so rh_png_encode): only the finished PNG bytes are read back.  The deflate stream differs
        from zlib's; the decoded pixels are the same."""
        import ctypes as C
        from . import _ffi
        if video_frame.image_data is None:
            return ("err", "VideoFrame could not be written because image_data is None.")
        L = _ffi.lib()
        cap = L.rh_png_bound(video_frame.width, video_frame.height)
        out = (C.c_uint8 * cap)()
        n = C.c_size_t()
        _ffi.check(L.rh_png_encode(renderer.device, video_frame.image_data.buffer(), video_frame.width, video_frame.height,
                                   video_frame.padded_bytes_per_row, out, cap, C.byref(n)))
        return ("ok", bytes(memoryview(out)[:n.value]))

    @staticmethod
    def encode(video_frame, writer):
        status, value = PngEncoder.encode_to_bytes(video_frame)
        if status == "ok":
            writer.write(value)
            return ("ok", None)
        return (status, value)
