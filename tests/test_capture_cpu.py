"""CPU tests of the capture-path host logic: packet parsing, frame re-ordering with Missing synthesis, file name
grammar, PNG encoding, the ffmpeg command line and the host LZ4 frame decoder.  No GPU."""
import io
import os
import stat
import subprocess

import numpy as np
import pytest


def make_frame(number, w=4, h=2, dropped=False):
    from renderer_amd import Format, FrameStatus, ImageData, VideoFrame
    from renderer_amd.video_frame import AtomicUsize
    from renderer_amd.video_recorder import capture_layout
    unpadded, padded, size = capture_layout(w, h, 4)
    data = None
    if not dropped:
        img = np.zeros((h, padded), dtype=np.uint8)
        img[:, :unpadded] = (np.arange(h * unpadded).reshape(h, unpadded) + number) % 251
        data = ImageData.Bytes(img.tobytes())
    f = VideoFrame(FrameStatus.Dropped if dropped else FrameStatus.Captured, data, w, h, Format.RgbaU8, unpadded, padded,
                   number, size, AtomicUsize(0))
    f._dropped = True
    return f


def packet(frame):
    meta = frame.encode()
    image = frame.image_data.bytes() if frame.image_data is not None else b""
    return (16 + len(meta) + len(image)).to_bytes(8, "big") + len(meta).to_bytes(8, "big") + meta + image


def write_sz(path, frames, oracle):
    s = oracle.LibLz4Stream(0)
    for f in frames:
        p = packet(f)
        meta_len = int.from_bytes(p[8:16], "big")
        s.write(p[:8]); s.write(p[8:16]); s.write(p[16:16 + meta_len])
        if len(p) > 16 + meta_len:
            s.write(p[16 + meta_len:])
    open(path, "wb").write(s.finish())


def test_filename_grammar():
    import datetime
    from renderer_amd.compressor import generate_timestamp
    from renderer_amd.decompressor import recover_timestamp_from_filename
    now = datetime.datetime(2024, 5, 1, 12, 34, 56, 789000, tzinfo=datetime.timezone.utc)
    ts = generate_timestamp(now)
    assert ts == "2024-05-01T12_34_56.789Z"
    assert recover_timestamp_from_filename(f"{ts}--3.sz") == now
    assert recover_timestamp_from_filename(f"{ts}--3.txt") is None
    assert recover_timestamp_from_filename("garbage--0.sz") is None


def test_host_decoder_and_packet_reader(oracle, tmp_path):
    from renderer_amd.decompressor import lz4f_decompress, packets_of_bytes as read_packets
    frames = [make_frame(1), make_frame(2, dropped=True), make_frame(3, w=100, h=3)]
    path = tmp_path / "s.sz"
    write_sz(path, frames, oracle)
    data = open(path, "rb").read()
    raw, complete = lz4f_decompress(data)
    assert complete and raw == b"".join(packet(f) for f in frames)
    got = list(read_packets(raw))
    assert [f.frame_number for f in got] == [1, 2, 3]
    assert got[1].image_data is None and got[0].image_data.bytes() == frames[0].image_data.bytes()
    # truncated tail: whole packets before the cut survive
    raw2, complete2 = lz4f_decompress(data[:-30])
    assert not complete2
    assert [f.frame_number for f in read_packets(raw2)] in ([1, 2], [1, 2, 3])


def test_reordering_and_missing_frames(oracle, tmp_path):
    from renderer_amd import Decompressor, FrameStatus
    ts = "2024-05-01T12_34_56.789Z"
    # three stream files; frame 5 is absent everywhere (a deleted file's worth), frame 4 was dropped while recording
    write_sz(tmp_path / f"{ts}--0.sz", [make_frame(1), make_frame(6), make_frame(9)], oracle)
    write_sz(tmp_path / f"{ts}--1.sz", [make_frame(2), make_frame(4, dropped=True), make_frame(7)], oracle)
    write_sz(tmp_path / f"{ts}--2.sz", [make_frame(3), make_frame(8)], oracle)
    open(tmp_path / f"{ts}--3.sz", "wb").close()          # empty files are ignored
    seen = []
    Decompressor(str(tmp_path), False).decompress_from_disk(
        lambda f, t: f.frame_number * 10, lambda f, res, t: seen.append((f.frame_number, f.status, res)))
    assert [n for n, _, _ in seen] == list(range(1, 10))
    assert seen[3][1] is FrameStatus.Dropped and seen[3][2] == ("ok", 40)
    assert seen[4][1] is FrameStatus.Missing and seen[4][2][0] == "err"
    assert all(s is FrameStatus.Captured for n, s, _ in seen if n not in (4, 5))
    assert len(os.listdir(tmp_path)) == 4                  # not removed


def test_png_encoder_strips_row_padding():
    from PIL import Image
    from renderer_amd import PngEncoder
    f = make_frame(5, w=7, h=3)
    status, png = PngEncoder.encode_to_bytes(f)
    assert status == "ok"
    img = np.array(Image.open(io.BytesIO(png)).convert("RGBA"))
    rows = np.frombuffer(f.image_data.bytes(), dtype=np.uint8).reshape(3, f.padded_bytes_per_row)
    assert np.array_equal(img, rows[:, :28].reshape(3, 7, 4))
    assert PngEncoder.encode_to_bytes(make_frame(6, dropped=True)) == ("err", "VideoFrame could not be written because image_data is None.")


def test_ffmpeg_pipe_command_line_and_duplication(tmp_path, monkeypatch, capfd):
    from renderer_amd import FfmpegPipe
    # a stand-in `ffmpeg` on PATH that stores its argv and stdin
    fake = tmp_path / "ffmpeg"
    fake.write_text("#!/bin/sh\nprintf '%s\\n' \"$@\" > \"$FAKE_OUT.argv\"\ncat > \"$FAKE_OUT.stdin\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    monkeypatch.setenv("FAKE_OUT", str(tmp_path / "out"))
    assert FfmpegPipe.available()
    pipe = FfmpegPipe(None, str(tmp_path), "recorded.mp4", ["-c:v", "libx264", "-r", "60"])
    pipe.write(make_frame(1, dropped=True), b"", None)         # nothing to duplicate yet: ignored
    assert pipe.child is None
    pipe.write(make_frame(1), b"PNG1", None)
    pipe.write(make_frame(2, dropped=True), b"", None)         # duplicates the previous frame
    pipe.write(make_frame(3), b"PNG3", None)
    pipe.close()
    argv = open(tmp_path / "out.argv").read().split("\n")[:-1]
    assert argv == ["-hide_banner", "-loglevel", "error", "-stats", "-f", "image2pipe", "-framerate", "60", "-y", "-i", "-",
                    "-c:v", "libx264", "-r", "60", str(tmp_path / "recorded.mp4")]
    assert open(tmp_path / "out.stdin", "rb").read() == b"PNG1PNG1PNG3"
    assert "Duplicating previous frame" in capfd.readouterr().err


def noisy_frame(number, w, h, seed):
    """A frame whose image compresses into many kinds of sequences: flat runs, a repeating row, noise."""
    f = make_frame(number, w=w, h=h)
    rng = np.random.default_rng(seed)
    rows = np.frombuffer(f.image_data.bytes(), dtype=np.uint8).reshape(h, f.padded_bytes_per_row).copy()
    rows[h // 4: h // 2] = rng.integers(0, 255, rows[h // 4: h // 2].shape, dtype=np.uint8)       # incompressible: stored blocks
    rows[h // 2:] = rows[h // 2]                                                                # far matches, one row apart
    rows[: h // 4, ::3] = 7                                                                     # short overlapping matches
    from renderer_amd import ImageData
    f.image_data = ImageData.Bytes(rows.tobytes())
    return f


def decode_with(device, path_frames, tmp_path, oracle, cut=None, piece=None):
    """Writes stream files with liblz4, reads them back through StreamReaders; returns the frames per stream."""
    from renderer_amd.decompressor import StreamReaders, read_packets
    paths = []
    for i, frames in enumerate(path_frames):
        path = tmp_path / f"s{i}.sz"
        write_sz(path, frames, oracle)
        if cut is not None and i == 0:
            data = open(path, "rb").read()
            open(path, "wb").write(data[:-cut])
        paths.append(str(path))
    readers = StreamReaders(paths, device)
    if piece:
        readers.FILE_PIECE = piece
    try:
        # interleaved, one frame at a time, the way the re-ordering loop pulls
        workers = [read_packets(lambda n, i=i: readers.read_exact(i, n)) for i in range(len(paths))]
        got = [[] for _ in paths]
        live = list(range(len(paths)))
        while live:
            for i in list(live):
                f = next(workers[i], None)
                if f is None:
                    live.remove(i)
                else:
                    got[i].append(f)
        return got
    finally:
        readers.close()


def check_streams(got, want):
    for g, w in zip(got, want):
        assert [f.frame_number for f in g] == [f.frame_number for f in w]
        for a, b in zip(g, w):
            assert (a.image_data is None) == (b.image_data is None)
            if b.image_data is not None:
                assert a.image_data.bytes() == b.image_data.bytes(), f"frame {b.frame_number}"


def test_stream_readers_on_the_host(oracle, tmp_path):
    """Several stream files decoded together, a piece of each file at a time: frames of many blocks (history across
    blocks and across decoder runs), stored blocks, dropped frames, tiny pieces."""
    streams = [[noisy_frame(1, 500, 300, 1), make_frame(4, dropped=True), noisy_frame(7, 640, 700, 2)],
               [noisy_frame(2, 64, 64, 3), noisy_frame(5, 1000, 900, 4)],
               [make_frame(3), make_frame(6, w=100, h=3)]]
    check_streams(decode_with(None, streams, tmp_path, oracle), streams)
    check_streams(decode_with(None, streams, tmp_path, oracle, piece=777), streams)


def test_stream_readers_stop_at_a_truncated_tail(oracle, tmp_path):
    """A file cut in the middle of a block gives the packets that are complete in the blocks before the cut (reference
    src/decompressor.rs:216-235: read_exact fails, the worker ends); the other files are not affected."""
    streams = [[noisy_frame(1, 300, 200, 5), noisy_frame(3, 300, 200, 6)], [noisy_frame(2, 300, 200, 7)]]
    got = decode_with(None, streams, tmp_path, oracle, cut=5000)
    assert [f.frame_number for f in got[0]] == [1]
    check_streams([got[0], got[1]], [streams[0][:1], streams[1]])


def test_stream_readers_reject_garbage(tmp_path):
    from renderer_amd.decompressor import StreamReaders
    path = tmp_path / "g.sz"
    open(path, "wb").write(b"not an lz4 frame at all, just bytes" * 10)
    readers = StreamReaders([str(path)])
    try:
        assert readers.read_exact(0, 8) is None
    finally:
        readers.close()
