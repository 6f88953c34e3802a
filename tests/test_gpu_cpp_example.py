"""The C++ host mirror (renderer_amd/cpp/renderer.hpp) driving the C ABI: the quads example, written call for call
like the reference's examples/quads_main_thread.rs, must reproduce the oracle's pixels for the same frame and leave a
decodable recording."""
import os
import subprocess

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "quads_main_thread")


def test_cpp_quads_example_matches_oracle(oracle, tmp_path):
    from PIL import Image
    from renderer_amd import Decompressor
    assert os.path.exists(EXE), "examples/quads_main_thread is built by `python -m renderer_amd.build`"
    W, H, frames = 480, 270, 60
    out = subprocess.run([EXE, os.path.join(ROOT, "renderer_amd", "shaders"), os.path.join(ROOT, "tests", "golden", "letter_f.png"),
                          str(tmp_path), str(frames), str(W), str(H)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    scene = scenes.example_scene(frames, width=W, height=H, record=True)
    want = scenes.draw_oracle(scene)["recording"]
    from renderer_amd.viewport import Viewport
    vp = Viewport(1.0, 1.0, float(W), float(H))
    x0, y0 = int(np.ceil(vp.margin_x)), int(np.ceil(vp.margin_y))
    w, h = int(np.floor(vp.width)), int(np.floor(vp.height))
    got = np.array(Image.open(os.path.join(tmp_path, f"frame_{frames}.png")).convert("RGBA"))
    assert got.shape == (h, w, 4)
    assert np.array_equal(got, want[y0:y0 + h, x0:x0 + w])
    os.remove(os.path.join(tmp_path, f"frame_{frames}.png"))

    # the recording: 4 stream files that decode to frames 1..60 in order, the last one equal to the PNG
    seen = []
    Decompressor(str(tmp_path), False).decompress_from_disk(lambda f, ts: None, lambda f, result, ts: seen.append(f))
    assert [f.frame_number for f in seen] == list(range(1, frames + 1))
    last = seen[-1]
    rows = np.frombuffer(last.image_data.bytes(), dtype=np.uint8).reshape(last.height, last.padded_bytes_per_row)
    assert np.array_equal(rows[:, :last.unpadded_bytes_per_row].reshape(h, w, 4), got)


def test_cpp_quads_example_multisampled(oracle, tmp_path):
    """The same example with 4 samples (no recording: the recording attachment is single-sampled): the screen equals the
    oracle's resolved image."""
    from PIL import Image
    W, H, frames = 480, 270, 30
    out = subprocess.run([EXE, os.path.join(ROOT, "renderer_amd", "shaders"), os.path.join(ROOT, "tests", "golden", "letter_f.png"),
                          str(tmp_path), str(frames), str(W), str(H), "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    scene = scenes.example_scene(frames, width=W, height=H)
    scene["samples"] = 4
    want = scenes.draw_oracle(scene)["color"][..., [2, 1, 0, 3]]          # the screen is BGRA
    got = np.array(Image.open(os.path.join(tmp_path, f"frame_{frames}.png")).convert("RGBA"))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("samples", [1, 4])
def test_cpp_render_thread_example_matches_oracle(oracle, tmp_path, samples):
    """renderer_amd/cpp/render_thread.hpp: the quads example with every call sent as a message to the worker thread that
    owns the Renderer (index handles for all objects); the final screen equals the oracle's frame."""
    from PIL import Image
    exe = os.path.join(ROOT, "examples", "quads_render_thread")
    assert os.path.exists(exe), "examples/quads_render_thread is built by `python -m renderer_amd.build`"
    W, H, frames = 480, 270, 45
    out = subprocess.run([exe, os.path.join(ROOT, "renderer_amd", "shaders"), os.path.join(ROOT, "tests", "golden", "letter_f.png"),
                          str(tmp_path), str(frames), str(W), str(H), str(samples)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    scene = scenes.example_scene(frames, width=W, height=H)
    scene["samples"] = samples
    want = scenes.draw_oracle(scene)["color"][..., [2, 1, 0, 3]]          # the screen is BGRA
    got = np.array(Image.open(os.path.join(tmp_path, f"frame_{frames}.png")).convert("RGBA"))
    assert np.array_equal(got, want)
