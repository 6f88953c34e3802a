"""A fixed set of seeds of the randomised parity sweep (tests/fuzz_parity.py): random scene shapes, topologies, blends,
filters, viewports, formats, sample counts and load ops, HIP against the oracle bit for bit."""
import pytest

import fuzz_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first_seed", [0, 1000, 2000])
def test_random_scenes_match_the_oracle(oracle, first_seed):
    for seed in range(first_seed, first_seed + 25):
        err, what = fuzz_parity.run_one(seed)
        assert err is None, f"seed {seed}: {err} {what}"


def test_random_lz4_streams_match_liblz4(oracle):
    import fuzz_codec
    # (91016: a match found far along the walk whose backward catch-up leaves a short literal run)
    for seed in list(range(5)) + [91016]:
        err = fuzz_codec.run_round(seed)
        assert err is None, f"seed {seed}: {err}"


def test_random_images_survive_the_gpu_png_encoder(oracle):
    """Random sizes and contents (runs to the left, copies of the row above, noise, long flat stretches crossing the
    64-pixel run limit, single rows and columns): every PNG is valid and decodes to the input."""
    import ctypes as C
    import io
    import zlib

    import numpy as np
    from PIL import Image

    import test_gpu_codec as T
    from renderer_amd import PngEncoder, Renderer, _ffi
    from renderer_amd.video_frame import FrameStatus, ImageData, VideoFrame
    from renderer_amd.video_recorder import capture_layout
    r = Renderer(64, 64)
    L = _ffi.lib()
    rng = np.random.default_rng(77)
    try:
        for it in range(120):
            W = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 129, int(rng.integers(1, 500))]))
            H = int(rng.choice([1, 2, 3, int(rng.integers(1, 120))]))
            kind = rng.choice(["noise", "flat", "rowcopy", "runs", "sparse"])
            if kind == "noise":
                pix = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            elif kind == "flat":
                pix = np.tile(rng.integers(0, 256, 4, dtype=np.uint8), (H, W, 1))
            elif kind == "rowcopy":
                pix = np.tile(rng.integers(0, 256, (1, W, 4), dtype=np.uint8), (H, 1, 1))
                pix[rng.integers(0, H), rng.integers(0, W)] ^= 0xff
            elif kind == "runs":
                pix = np.repeat(rng.integers(0, 256, (H, -(-W // 7), 4), dtype=np.uint8), 7, axis=1)[:, :W]
            else:
                pix = np.zeros((H, W, 4), dtype=np.uint8)
                pix[rng.random((H, W)) < 0.02] = rng.integers(0, 256, 4, dtype=np.uint8)
            pix = np.ascontiguousarray(pix)
            unpadded, padded, frame_size = capture_layout(W, H, 4)
            rows = np.full((H, padded), 0xee, dtype=np.uint8)
            rows[:, :unpadded] = pix.reshape(H, unpadded)
            buf = C.c_void_p()
            _ffi.check(L.rh_buffer_create(r.device, frame_size, _ffi.BUFFER_COPY_DST | _ffi.BUFFER_MAP_READ, rows.ctypes.data, rows.nbytes, C.byref(buf)))
            frame = VideoFrame(FrameStatus.Captured, ImageData.Buffer(buf), W, H, 2, unpadded, padded, 1, frame_size)
            status, png = PngEncoder.encode_on_device(r, frame)
            assert status == "ok"
            assert len(zlib.decompress(T.png_idat_stream(png))) == H * (1 + W * 4), (it, W, H, kind)
            assert np.array_equal(np.array(Image.open(io.BytesIO(png)).convert("RGBA")), pix), (it, W, H, kind)
            frame.drop()
    finally:
        r.close()
