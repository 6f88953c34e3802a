"""GPU parity: the HIP rasteriser (through the C ABI) against the CPU oracle on the same seeded scenes.
Bar: bit-exact coverage and bit-exact colour bytes (both sides perform the same sequence of IEEE binary32 operations)."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def assert_same(hip, orc, what=""):
    assert hip.shape == orc.shape
    if not np.array_equal(hip, orc):
        diff = np.argwhere((hip != orc).any(axis=2))
        y, x = diff[0]
        raise AssertionError(f"{what}: {len(diff)} pixels differ; first at (x={x}, y={y}): hip={hip[y, x]} oracle={orc[y, x]}")


@pytest.mark.parametrize("frame", [1, 2, 60, 600])
def test_quads_example_frames(oracle, frame):
    scene = scenes.example_scene(frame)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"frame {frame}")


def test_quads_example_recording(oracle):
    scene = scenes.example_scene(37, record=True)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "screen")
    assert_same(hip["recording"], orc["recording"], "recording")
    # screen is BGRA, recording is RGBA, same content
    assert np.array_equal(hip["color"][..., [2, 1, 0, 3]], hip["recording"])


@pytest.mark.parametrize("blend", ["premultiplied", "additive"])
@pytest.mark.parametrize("filter", ["linear", "nearest"])
def test_random_quads(oracle, blend, filter):
    scene = scenes.random_quads_scene(20000, 16, 640, 360, seed=7, blend=blend, filter=filter)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"{blend}/{filter}")


def test_tiny_quads_additive(oracle):
    scene = scenes.tiny_quads_scene(100000, 512, 512, seed=2)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "tiny")


def test_large_quads_and_odd_size(oracle):
    # target size not a multiple of the tile size; quads larger than a tile; quads poking outside the target
    scene = scenes.random_quads_scene(300, 150, 333, 217, seed=3)
    scene["instances"] = (scene["instances"] * np.float32(3.0)).astype(np.float32)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "large")
