"""GPU parity of the capture codec: LZ4F streams compressed on the device must be byte-identical to the oracle
(restatement of liblz4) and to the system liblz4 1.9.3 for the same write sequences, and the recorded-frame pipeline
(record -> GPU compress -> files -> decompress -> re-order) must return the captured pixels."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def gpu_streams(packets_per_stream, n_streams):
    """packets_per_stream: list (per stream) of lists of (metadata bytes, image bytes | None).  Returns stream bytes."""
    from renderer_amd import Renderer, _ffi
    L = _ffi.lib()
    r = Renderer(64, 64)
    comp = C.c_void_p()
    _ffi.check(L.rh_compressor_create(r.device, n_streams, 0, C.byref(comp)))
    buffers = []
    rounds = max(len(p) for p in packets_per_stream)
    for k in range(rounds):
        for s, packets in enumerate(packets_per_stream):
            if k >= len(packets):
                continue
            meta, image = packets[k]
            handle = None
            if image is not None:
                arr = np.frombuffer(image, dtype=np.uint8)
                out = C.c_void_p()
                _ffi.check(L.rh_buffer_create(r.device, len(image), _ffi.BUFFER_COPY_DST | _ffi.BUFFER_MAP_READ,
                                              arr.ctypes.data, len(image), C.byref(out)))
                handle = out; buffers.append(out)
            _ffi.check(L.rh_compressor_write_packet(comp, s, meta, len(meta), handle))
        _ffi.check(L.rh_compressor_flush(comp))
    streams = []
    for s in range(n_streams):
        _ffi.check(L.rh_compressor_finish_stream(comp, s))
        n = L.rh_compressor_pending(comp, s)
        buf = (C.c_uint8 * n)(); got = C.c_size_t()
        _ffi.check(L.rh_compressor_read(comp, s, buf, n, C.byref(got)))
        streams.append(bytes(buf[:got.value]))
    for b in buffers:
        L.rh_buffer_release(b)
    L.rh_compressor_destroy(comp)
    r.close()
    return streams


def reference_stream(cls, packets):
    s = cls()
    for meta, image in packets:
        plen = 16 + len(meta) + (len(image) if image is not None else 0)
        s.write(plen.to_bytes(8, "big")); s.write(len(meta).to_bytes(8, "big")); s.write(meta)
        if image is not None:
            s.write(image)
    return s.finish()


def image_bytes(rng, n, kind):
    if kind == "flat":
        return bytes([0, 0, 0, 255]) * (n // 4)
    if kind == "noise":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += rng.integers(0, 256, 4, dtype=np.uint8).tobytes() * int(rng.integers(1, 300))
        return bytes(out[:n])
    if kind == "text":
        base = rng.integers(0, 256, 2000, dtype=np.uint8).tobytes()
        out = bytearray()
        while len(out) < n:
            s = int(rng.integers(0, 1900)); out += base[s:s + int(rng.integers(4, 100))]
        return bytes(out[:n])
    raise ValueError(kind)


def test_streams_byte_identical_to_oracle_and_liblz4(oracle):
    rng = np.random.default_rng(42)
    meta = bytes(range(23))
    kinds = ["flat", "noise", "runs", "text"]
    sizes = [256, 4352 * 3, 65536, 65536 + 4, 3 * 65536, 200_000, 1_000_000]
    packets_per_stream = []
    for s in range(12):
        packets = []
        for k in range(3):
            if (s + k) % 5 == 4:
                packets.append((meta[:11], None))                    # dropped frame: metadata only
            else:
                n = sizes[(s * 3 + k) % len(sizes)]
                packets.append((meta, image_bytes(rng, n - n % 4, kinds[(s + k) % 4])))
        packets_per_stream.append(packets)
    got = gpu_streams(packets_per_stream, len(packets_per_stream))
    for s, packets in enumerate(packets_per_stream):
        want = reference_stream(oracle.Lz4OracleStream, packets)
        assert got[s] == want, f"stream {s}: GPU stream differs from the oracle (len {len(got[s])} vs {len(want)})"
        assert got[s] == reference_stream(lambda: oracle.LibLz4Stream(0), packets), f"stream {s}: differs from liblz4"
        # and it decodes back to the packet bytes
        plain = b"".join((16 + len(m) + (len(i) if i else 0)).to_bytes(8, "big") + len(m).to_bytes(8, "big") + m + (i or b"")
                         for m, i in packets)
        dec, _, complete = oracle.lz4_decompress_frame(got[s], len(plain) + 16)
        assert complete and dec == plain


def test_golden_streams_on_gpu(oracle):
    import hashlib
    import json
    import test_oracle_cpu
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lz4_kat.json")))
    cases = test_oracle_cpu.kat_cases()
    # the golden write sequences are packets: 8 B, 8 B, metadata [, image] (+ a second packet in one case)
    names = sorted(cases)
    packets_per_stream = []
    for name in names:
        w = cases[name]
        packets = []
        i = 0
        while i < len(w):
            meta = w[i + 2]
            plen = int.from_bytes(w[i], "big")
            has_image = plen > 16 + len(meta)
            packets.append((meta, w[i + 3] if has_image else None))
            i += 4 if has_image else 3
        packets_per_stream.append(packets)
    got = gpu_streams(packets_per_stream, len(names))
    for name, stream in zip(names, got):
        assert len(stream) == gold["cases"][name]["stream_len"], name
        assert hashlib.sha256(stream).hexdigest() == gold["cases"][name]["stream_sha256"], name


def test_record_compress_decompress_roundtrip(oracle, tmp_path):
    """Quads example with recording on: frames stay on the device, are LZ4-compressed there, written as .sz files and
    come back through the decompressor in frame order with the recorded pixels."""
    from renderer_amd import Compressor, Decompressor, PngEncoder, Renderer
    W, H = 320, 180
    scene = scenes.example_scene(1, width=W, height=H)
    r = Renderer(W, H)
    hs = scenes.HipScene(scene, renderer=r)
    comp = Compressor(r, str(tmp_path), n_streams=3)
    r.start_recording([hs.pipeline], Renderer.clear_color(0., 0., 0., 1.), 500., comp.compress_to_disk, keep_on_device=True)
    r.flush()                # the scene set-up uploaded the instance buffer once; a new upload needs a new flush interval
    expected = []
    n_frames = 7
    for k in range(1, n_frames + 1):
        s = scenes.example_scene(k, width=W, height=H, record=True)
        r.set_instanced(hs.pipeline, (0, 0), s["instances"])
        r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        r.finish_frame()
        expected.append(scenes.draw_oracle(s)["recording"])
    r.stop_recording([hs.pipeline])
    comp.finish()
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 3 and all(f.endswith(".sz") and "--" in f for f in files)
    assert Decompressor.can_run(str(tmp_path))

    seen = []

    def in_order(video_frame, result, timestamp):
        seen.append((video_frame, result))

    Decompressor(str(tmp_path), True, renderer=r).decompress_from_disk(lambda f, ts: PngEncoder.encode_to_bytes(f), in_order)
    assert [f.frame_number for f, _ in seen] == list(range(1, n_frames + 1))
    vw = hs.viewport
    x0, y0 = int(np.ceil(vw.margin_x)), int(np.ceil(vw.margin_y))
    for k, (frame, (status, png)) in enumerate(seen):
        assert status == "ok" and png[0] == "ok"
        assert frame.width == int(vw.width) and frame.height == int(vw.height)
        rows = np.frombuffer(frame.image_data.bytes(), dtype=np.uint8).reshape(frame.height, frame.padded_bytes_per_row)
        pixels = rows[:, :frame.unpadded_bytes_per_row].reshape(frame.height, frame.width, 4)
        want = expected[k][y0:y0 + frame.height, x0:x0 + frame.width]
        assert np.array_equal(pixels, want), f"frame {k + 1}"
        # the PNG decodes to the same pixels
        from PIL import Image
        import io
        assert np.array_equal(np.array(Image.open(io.BytesIO(png[1])).convert("RGBA")), want)
    assert os.listdir(tmp_path) == []          # removed after successful decompression
    hs.close(); r.close()


def test_4k_recording_under_the_500_mb_budget_drops_nothing(oracle, tmp_path):
    """The budget of the reference's example (500 MB, examples/quads_main_thread.rs:95) holds 16 frames of 4K; a frame's
    bytes stay in the budget until its round has been compressed (reference src/video_frame.rs:70-77), so the compressor
    cuts its rounds to what the budget holds instead of waiting for one frame per stream: nothing is dropped, and the
    bytes in flight never pass what the recorder admits."""
    from renderer_amd import Compressor, Renderer
    W, H = 3840, 2160
    scene = scenes.random_quads_scene(200, 40, W, H, seed=21, target="screen")
    r = Renderer(W, H)
    hs = scenes.HipScene(scene, renderer=r)
    comp = Compressor(r, str(tmp_path), n_streams=64, max_buffer_size_in_megabytes=500.)
    frame_bytes = W * H * 4
    peak = [0]
    statuses = []

    def sink(frame):
        peak[0] = max(peak[0], frame.buffer_size_in_bytes.load())
        statuses.append(frame.status.name)
        comp.compress_to_disk(frame)

    r.start_recording([hs.pipeline], Renderer.clear_color(0., 0., 0., 1.), 500., sink, keep_on_device=True)
    r.flush()
    n_frames = 40
    for _ in range(n_frames):
        r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        r.finish_frame()
    budget = r.recorder.buffer_size_in_bytes
    r.stop_recording([hs.pipeline])
    comp.finish()
    assert statuses == ["Captured"] * n_frames and comp.frames_dropped == 0 and comp.frames_captured == n_frames
    assert comp._round_size(type("F", (), {"frame_size_in_bytes": frame_bytes})) == 8
    assert peak[0] <= 500 * 1024 * 1024 + frame_bytes
    assert budget.load() == 0                   # every frame gave its bytes back
    assert sum(os.path.getsize(os.path.join(tmp_path, f)) for f in os.listdir(tmp_path)) > 0
    hs.close(); r.close()


def test_device_decoder_against_liblz4_streams(oracle, tmp_path):
    """Stream files written by liblz4 (the library the reference links), decoded on the device, all streams in the same
    launches: frames of many blocks (history across blocks and across launches), stored blocks, short overlapping
    matches, dropped frames; the same bytes as the host decoder gives.  Then with file pieces of 777 bytes."""
    import test_capture_cpu as cc
    from renderer_amd import Renderer
    r = Renderer(64, 64)
    try:
        streams = [[cc.noisy_frame(1, 500, 300, 1), cc.make_frame(4, dropped=True), cc.noisy_frame(7, 1920, 1080, 2)],
                   [cc.noisy_frame(2, 64, 64, 3), cc.noisy_frame(5, 1000, 900, 4)],
                   [cc.make_frame(3), cc.make_frame(6, w=100, h=3)],
                   [cc.noisy_frame(8 + k, 320, 200, 10 + k) for k in range(5)]]
        cc.check_streams(cc.decode_with(r.device, streams, tmp_path, oracle), streams)
        cc.check_streams(cc.decode_with(r.device, streams[:3], tmp_path, oracle, piece=777), streams[:3])
    finally:
        r.close()


def test_device_decoder_stops_at_a_truncated_tail_and_at_malformed_blocks(oracle, tmp_path):
    """A file cut inside a block ends at its last complete block (reference src/decompressor.rs:216-235); a block whose
    bytes are damaged ends its stream without touching the others, whatever the damage claims (lengths and offsets are
    checked against the block and the decoded data before anything is copied)."""
    import test_capture_cpu as cc
    from renderer_amd import Renderer
    from renderer_amd.decompressor import StreamReaders, read_packets
    r = Renderer(64, 64)
    try:
        streams = [[cc.noisy_frame(1, 300, 200, 5), cc.noisy_frame(3, 300, 200, 6)], [cc.noisy_frame(2, 300, 200, 7)]]
        got = cc.decode_with(r.device, streams, tmp_path, oracle, cut=5000)
        assert [f.frame_number for f in got[0]] == [1]
        cc.check_streams(got, [streams[0][:1], streams[1]])
        # damage: every position of a small stream in turn overwritten with 0xff (long lengths, far offsets, bad headers)
        good = tmp_path / "good.sz"
        cc.write_sz(good, [cc.noisy_frame(9, 40, 30, 8)], oracle)
        data = bytearray(open(good, "rb").read())
        rng = np.random.default_rng(9)
        for at in sorted(set(rng.integers(7, len(data) - 4, 60).tolist())):
            bad = bytearray(data)
            bad[at] = 0xff
            p = tmp_path / "bad.sz"
            open(p, "wb").write(bad)
            readers = StreamReaders([str(p), str(good)], r.device)
            try:
                damaged = list(read_packets(lambda n: readers.read_exact(0, n)))
                intact = list(read_packets(lambda n: readers.read_exact(1, n)))
            finally:
                readers.close()
            assert len(damaged) <= 1 and [f.frame_number for f in intact] == [9]
    finally:
        r.close()


def png_idat_stream(png):
    """Concatenated IDAT payloads, checking every chunk's CRC on the way."""
    import struct
    import zlib
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, kinds = 8, b"", []
    while pos < len(png):
        n, tag = struct.unpack(">I4s", png[pos:pos + 8])
        body = png[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(tag + body) & 0xffffffff == crc, f"bad CRC in {tag} chunk at {pos}"
        kinds.append(tag)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND" and pos == len(png)
    return idat


@pytest.mark.parametrize("kind", ["scene", "noise", "flat", "stripes"])
@pytest.mark.parametrize("size", [(640, 360), (67, 5), (1, 1)])
def test_png_encoded_on_the_device_decodes_to_the_pixels(oracle, kind, size):
    """GPU PNG encoder: a valid PNG (chunk CRCs, zlib stream with Adler-32) whose decoded pixels are the input's."""
    import io
    import zlib
    from PIL import Image
    from renderer_amd import PngEncoder, Renderer, _ffi
    from renderer_amd.video_frame import FrameStatus, ImageData, VideoFrame
    from renderer_amd.video_recorder import capture_layout
    W, H = size
    rng = np.random.default_rng(W * 1000 + H)
    if kind == "scene":
        pix = scenes.draw_oracle(scenes.random_quads_scene(60, 40, W, H, seed=5, clear=(0., 0., 0., 1.)) if W > 100
                                 else scenes.random_quads_scene(2, 2, max(W, 2), max(H, 2), seed=5))["color"][:H, :W]
        pix = np.ascontiguousarray(pix)
    elif kind == "noise":
        pix = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    elif kind == "flat":
        pix = np.full((H, W, 4), 200, dtype=np.uint8)
    else:
        pix = np.zeros((H, W, 4), dtype=np.uint8)
        pix[:, ::3] = [255, 0, 0, 255]; pix[::2] //= 2
    unpadded, padded, frame_size = capture_layout(W, H, 4)
    rows = np.zeros((H, padded), dtype=np.uint8)
    rows[:, :unpadded] = pix.reshape(H, unpadded)
    rows[:, unpadded:] = 0xee                         # the padding must not leak into the image
    r = Renderer(64, 64)
    L = _ffi.lib()
    buf = C.c_void_p()
    _ffi.check(L.rh_buffer_create(r.device, frame_size, _ffi.BUFFER_COPY_DST | _ffi.BUFFER_MAP_READ, rows.ctypes.data, rows.nbytes, C.byref(buf)))
    frame = VideoFrame(FrameStatus.Captured, ImageData.Buffer(buf), W, H, 2, unpadded, padded, 1, frame_size)
    status, png = PngEncoder.encode_on_device(r, frame)
    assert status == "ok"
    raw = zlib.decompress(png_idat_stream(png))
    assert len(raw) == H * (1 + W * 4)
    got = np.array(Image.open(io.BytesIO(png)).convert("RGBA"))
    assert np.array_equal(got, pix)
    if kind == "flat" and W * H > 100000:
        assert len(png) < pix.nbytes // 20            # runs really are encoded as matches
    frame.drop(); r.close()


def test_frame_parallel_recording_to_one_ordered_png_stream(oracle):
    """tools/record_frame_parallel.py at world size 1: frames rendered, recorded, PNG-encoded on the GPU and written in
    frame order as the concatenated PNG stream ffmpeg's image2pipe reads; every PNG decodes to the oracle's frame."""
    import io
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import record_frame_parallel
    W, H, n = 320, 180, 5
    got = []
    rendered = record_frame_parallel.record(n, W, H, lambda f, png: got.append((f, png)))
    assert rendered == n and [f for f, _ in got] == list(range(1, n + 1))
    from renderer_amd.viewport import Viewport
    vp = Viewport(1.0, 1.0, float(W), float(H))
    x0, y0, w, h = int(np.ceil(vp.margin_x)), int(np.ceil(vp.margin_y)), int(vp.width), int(vp.height)
    for f, png in got:
        want = scenes.draw_oracle(scenes.example_scene(f, width=W, height=H, record=True))["recording"][y0:y0 + h, x0:x0 + w]
        assert np.array_equal(np.array(Image.open(io.BytesIO(png)).convert("RGBA")), want), f"frame {f}"
