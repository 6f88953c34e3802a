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


def test_sort_first_strips_equal_single_pass(oracle):
    """Rendering a target as G row strips (what G ranks do, one strip each) gives the single-GPU image."""
    from renderer_amd import partition
    scene = scenes.random_quads_scene(4000, 40, 400, 300, seed=9)
    orc = scenes.draw_oracle(scene)
    for world in (2, 3):
        hs = scenes.HipScene(scene)
        for rank in range(world):
            r0, r1 = partition.strip_for_rank(scene["height"], rank, world)
            hs.renderer.set_strip(r0, r1)
            hs.renderer.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
            hs.renderer.finish_frame()
        hs.renderer.set_strip(0, 0)
        img = hs.read_color()
        hs.close()
        assert_same(img, orc["color"], f"{world} strips")


def test_load_op_and_second_pass_sampling_target(oracle):
    """Two-pass shape of the 16K configuration at small size: pass 1 renders quads into texture A, pass 2 draws one
    quad covering target B that samples A (linear), blended over B's previous contents (LoadOp::Load)."""
    from renderer_amd import Renderer
    W = H = 96
    s1 = scenes.random_quads_scene(200, 24, W, H, seed=21)
    a_img = scenes.draw_oracle(s1)["color"]
    s2 = dict(shader="hello", width=W, height=H, target="texture", position=scenes.UNIT_POSITION, uv=scenes.QUAD_UV,
              instances=np.zeros((1, 2), dtype=np.float32), texture=a_img, texture_format="rgba_u8", filter="linear",
              blend="premultiplied", topology="strip", clear=None, viewport=None, record=False)
    init = np.full((H, W, 4), 40, dtype=np.uint8)
    want = scenes.draw_oracle(s2, initial=init)["color"]
    # HIP: same renderer, pass 2 samples the texture pass 1 rendered into
    r = Renderer(W, H)
    h1 = scenes.HipScene(s1, renderer=r)
    h1.draw()
    assert_same(h1.read_color(), a_img, "pass 1")
    s2_hip = dict(s2); s2_hip["texture"] = h1.read_color()
    h2 = scenes.HipScene(s2_hip, renderer=r)
    h2.set_initial(init)
    h2.draw()
    assert_same(h2.read_color(), want, "pass 2")
    r.close()
