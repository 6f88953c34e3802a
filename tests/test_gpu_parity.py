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


@pytest.mark.parametrize("lo,hi", [(1, 3), (2, 8), (5, 12)])
def test_tiny_quads_order_sensitive_blend(oracle, lo, hi):
    """Small axis-aligned quads share shading passes when their boxes are disjoint; with pre-multiplied alpha the
    result depends on the submission order of every overlapping pair, ~10 layers deep here."""
    scene = scenes.tiny_quads_scene(60000, 320, 200, seed=11 + lo, lo=lo, hi=hi)
    scene["blend"] = "premultiplied"
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"tiny premultiplied {lo}..{hi}")


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


@pytest.mark.parametrize("topology", ["strip", "list"])
def test_axis_aligned_edges_on_pixel_centres(oracle, topology):
    """Quads whose edges fall exactly on pixel centres and on pixel boundaries: the top-left rule decides every border
    pixel.  Strip quads take the rectangle path of the tile kernel (exact box + diagonal), list quads the general edge
    functions; both must agree with the oracle, which only has the general rule."""
    W = H = 256
    f32 = np.float32
    rng = np.random.default_rng(11)
    n = 600
    # centres on a half-pixel lattice, half extents in half pixels: corners land on k/2 pixel coordinates
    cx = (rng.integers(8, 2 * W - 8, n) * f32(0.5) / f32(W / 2) - f32(1.0)).astype(f32)
    cy = (rng.integers(8, 2 * H - 8, n) * f32(0.5) / f32(H / 2) - f32(1.0)).astype(f32)
    hx, hy = f32(7.5 / (W / 2)), f32(5.0 / (H / 2))
    scene = scenes.random_quads_scene(n, 16, W, H, seed=11, blend="additive", filter="nearest")
    scene["instances"] = np.stack([cx, cy], axis=1).astype(f32)
    corners = np.array([[-hx, -hy], [-hx, hy], [hx, -hy], [hx, hy]], dtype=f32)
    if topology == "strip":
        scene["position"] = corners
    else:
        order = [0, 1, 2, 1, 3, 2]
        scene["position"] = corners[order]
        scene["uv"] = scenes.QUAD_UV[order]
        scene["topology"] = "list"
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"ties/{topology}")
    assert orc["fragments"] > 0


@pytest.mark.parametrize("size,room", [((512, 288), 0), ((1024, 512), 150), ((1024, 512), 155), ((1024, 512), 460), ((1024, 512), 1500), ((1024, 512), 2500), ((1024, 512), 5000)])
def test_last_tiles_cut_into_bands_of_rows(oracle, size, room, monkeypatch):
    """The tiles beyond the last full round of workgroups run as 2, 4 or 8 bands of rows (RH_OPTION_TAIL_SPLIT): whole
    tiles only, one kind of band, the mix and thirds give the oracle's bytes.  144 tiles on the real GPU: eighths.  512
    tiles with the GPU taken to hold 150 workgroups: 450 whole tiles, then halves (the last 13 quartered in the mixed
    mode); 155: 465 whole, 47 halved (14 of them quartered in the mixed mode) or, mode 3, cut into thirds of 11, 11 and
    10 rows; 460: 460 whole, 52 in eighths; 1500: all halved (238 quartered in the mixed mode); 2500: all quartered; 5000: all in
    eighths."""
    if room:
        monkeypatch.setenv("RH_TILE_ROOM", str(room))
    W, H = size
    scene = scenes.random_quads_scene(20000 if W > 1024 else 6000, 24, W, H, seed=17)
    scene["record"] = True                       # two attachments and the captured frame
    want = scenes.draw_oracle(scene)
    for mode in (0, 1, 2, 3):
        hs = scenes.HipScene(scene)
        hs.renderer.set_tail_split(mode)
        hs.draw()
        got = {"color": hs.read_color(), "recording": hs.read_recording()}
        hs.renderer.synchronize()
        hs.renderer.recorder.process_mapped_buffers()
        frame, rows = hs.frames[0], []
        frame.image_data.bytes_fn(lambda b: rows.append(np.array(b).reshape(frame.height, frame.padded_bytes_per_row)))
        hs.close()
        for key in got:
            assert_same(got[key], want[key], f"mode={mode} {key}")
        # (the tile kernel writes the captured frame as its bands go back to memory)
        assert_same(rows[0][:, :W * 4].reshape(H, W, 4), want["recording"], f"mode={mode} captured frame")
        assert not rows[0][:, W * 4:].any()


@pytest.mark.parametrize("samples,fmt,room", [(4, "rgba_u8", 0), (2, "rgba_u8", 150), (4, "rgba_u8", 155), (8, "rgba_f16", 1500), (4, "rgba_f32", 2500)])
def test_last_tiles_cut_into_bands_of_rows_multisampled(oracle, samples, fmt, room, monkeypatch):
    """The multisampled tile kernels take bands of rows the same way: samples and resolved image identical to the
    oracle's with whole tiles only, one kind of band and the mix."""
    if room:
        monkeypatch.setenv("RH_TILE_ROOM", str(room))
    W, H = (512, 288) if not room else (1024, 512)
    scene = scenes.tilted_quads_scene(1500, W, H, seed=23)
    scene.update(samples=samples, target_format=fmt)
    want = scenes.draw_oracle(scene)
    for mode in (0, 1, 2, 3):
        hs = scenes.HipScene(scene)
        hs.renderer.set_tail_split(mode)
        hs.draw()
        got = {"color": hs.read_color(), "msaa": hs.pipeline.msaa_texture.read()}
        hs.close()
        for key in got:
            assert_same(got[key], want[key], f"mode={mode} {key}")


@pytest.mark.parametrize("name,scene_fn", [
    ("12 px (21:1, not a power of two: 16-byte items)", lambda: scenes.random_quads_scene(20000, 12, 1280, 720, seed=61)),
    ("24 px (10.7:1: 16-byte items)", lambda: scenes.random_quads_scene(6000, 24, 1280, 720, seed=62)),
    ("16 px (16:1: 32-byte items)", lambda: scenes.random_quads_scene(12000, 16, 1280, 720, seed=63)),
    ("sides 3..70 px mixed (the form changes from hit to hit inside a round)", lambda: scenes.tiny_quads_scene(9000, 1280, 720, seed=64, lo=3, hi=70)),
    ("sides 2..8 px, packs (64:1 and beyond)", lambda: scenes.tiny_quads_scene(60000, 1280, 720, seed=65)),
    ("a texture 96 texels wide (not a power of two: the hint's power-of-two bit says nothing about it)",
     lambda: dict(scenes.random_quads_scene(8000, 16, 1280, 720, seed=66), texture=np.ascontiguousarray(scenes.load_letter_f()[40:104, 80:176]))),
    ("a texture 2048 texels wide, magnified and minified", lambda: dict(scenes.tiny_quads_scene(4000, 1280, 720, seed=67, lo=3, hi=90),
                                                                       texture=np.ascontiguousarray(np.tile(scenes.load_letter_f(), (1, 8, 1))[:128]))),
])
def test_both_forms_of_footprint_items_match_the_oracle(oracle, name, scene_fn):
    """The bilinear body reads a texture through 32-byte footprint items (f16 pairs) where neighbouring fragments read
    neighbouring items -- magnified, or minified by a power of two -- and through 16-byte ones (the texels as bytes)
    elsewhere, chosen per primitive from the gradient hint in its record; the two give the same bits.  Both blends."""
    for blend in ("premultiplied", "additive"):
        scene = scene_fn(); scene["blend"] = blend
        want = scenes.draw_oracle(scene)
        got = scenes.draw_hip(scene, twice=True)
        assert_same(got["color"], want["color"], f"{name} / {blend}")
        assert want["fragments"] > 0


def test_wave_priority_changes_no_byte(oracle):
    """The tile waves' issue priority (RH_OPTION_WAVE_PRIORITY) only changes which wave a SIMD serves first: the oracle's
    bytes with it and without, whole tiles and bands."""
    scene = scenes.random_quads_scene(8000, 16, 1024, 512, seed=29)
    scene["record"] = True
    want = scenes.draw_oracle(scene)
    for enabled in (True, False):
        hs = scenes.HipScene(scene)
        hs.renderer.set_wave_priority(enabled)
        hs.draw()
        got = {"color": hs.read_color(), "recording": hs.read_recording()}
        hs.close()
        for key in got:
            assert_same(got[key], want[key], f"priority={enabled} {key}")


def test_general_and_specialised_kernel_bodies_agree(oracle):
    """The state-specialised kernel bodies and the general body give identical bytes."""
    scene = scenes.random_quads_scene(5000, 24, 512, 288, seed=5)
    want = scenes.draw_oracle(scene)["color"]
    for general in (False, True):
        hs = scenes.HipScene(scene)
        hs.renderer.set_general_kernels(general)
        hs.draw()
        got = hs.read_color()
        hs.close()
        assert_same(got, want, f"general={general}")


def test_quads_set_up_by_the_tile_kernel_and_by_the_set_up_launch_agree(oracle):
    """Axis-aligned strip quads with affine varyings are set up by the tile kernel itself, in LDS, pair by pair; with
    RH_OPTION_SETUP_LAUNCH the set-up launch writes their records instead, with its general code.  Both give the oracle's
    bytes: pre-multiplied 16 px quads, tiny additive quads (packs), the letterboxed example with its recording attachment,
    and a scene that mixes lean quads with projective ones in one pass (records of both kinds in one round)."""
    cases = [("16 px", scenes.random_quads_scene(20000, 16, 1280, 720, seed=41)),
             ("tiny", scenes.tiny_quads_scene(60000, 1280, 720, seed=42)),
             ("example", scenes.example_scene(37, 640, 360, record=True)),
             ("mixed", scenes.tilted_quads_scene(3000, 640, 360, seed=43))]
    for name, scene in cases:
        want = scenes.draw_oracle(scene)
        for launch in (False, True):
            hs = scenes.HipScene(scene)
            try:
                hs.renderer.set_setup_launch(launch)
                hs.draw(); hs.draw()                  # (the second: queued before its pair count is known)
                assert_same(hs.read_color(), want["color"], f"{name}, set-up launch {launch}")
                if scene.get("record"):
                    assert_same(hs.read_recording(), want["recording"], f"{name} recording, set-up launch {launch}")
            finally:
                hs.close()


def test_pass_queued_before_its_pair_count_is_known(oracle):
    """After its first pass a device queues a pass whole, with room for the pairs of the largest pass seen so far
    plus a quarter, and looks at the count afterwards.  Scenes that fit (same size, smaller) and a scene that does not
    (ten times the pairs: its kernels write nothing and the pass runs again) all match the oracle."""
    from renderer_amd import Renderer
    W, H = 640, 360
    r = Renderer(W, H)
    for n, seed in ((3000, 1), (3000, 2), (500, 3), (40000, 4), (40000, 5), (100, 6)):
        scene = scenes.random_quads_scene(n, 12, W, H, seed=seed)
        hs = scenes.HipScene(scene, renderer=r)
        hs.draw()
        got = hs.read_color()
        hs.close()
        assert_same(got, scenes.draw_oracle(scene)["color"], f"{n} quads after the previous scene")
    r.close()


def test_short_and_general_prefix_sums_agree(oracle):
    """Enough (tile, quad) pairs that the sort's histogram scan spans several chunks: the two-launch form (chunk bases
    summed inside the apply phase) and the three-launch form that serves any length order the quads identically."""
    scene = scenes.tiny_quads_scene(300000, 1024, 512, seed=31, lo=2, hi=6)
    scene["blend"] = "premultiplied"          # the result depends on the order of every overlapping pair
    want = scenes.draw_oracle(scene)["color"]
    for general in (False, True):
        hs = scenes.HipScene(scene)
        hs.renderer.set_general_kernels(general)
        hs.draw()
        got = hs.read_color()
        hs.close()
        assert_same(got, want, f"general={general}")


@pytest.mark.parametrize("fmt", ["r_u8", "bgra_u8"])
@pytest.mark.parametrize("filter", ["linear", "nearest"])
def test_other_texture_formats(oracle, fmt, filter):
    """R8 and BGRA8 textures take the general kernel body (format read at run time)."""
    rng = np.random.default_rng(21)
    tex = rng.integers(0, 256, (37, 53) if fmt == "r_u8" else (37, 53, 4), dtype=np.uint8)
    scene = scenes.random_quads_scene(3000, 20, 400, 300, seed=21, filter=filter, texture=tex)
    scene["texture_format"] = fmt
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"{fmt}/{filter}")


def test_one_texel_texture(oracle):
    tex = np.array([[[200, 100, 50, 255]]], dtype=np.uint8)
    for filter in ("linear", "nearest"):
        scene = scenes.random_quads_scene(500, 12, 256, 128, seed=22, filter=filter, texture=tex, blend="additive")
        assert_same(scenes.draw_hip(scene)["color"], scenes.draw_oracle(scene)["color"], f"1x1/{filter}")


def test_viewport_clip_switch(oracle):
    """Quads poking out of a letterboxed viewport: clipped to it by default, to the attachment with the option off."""
    scene = scenes.example_scene(1, width=640, height=360)
    scene["instances"] = np.array([[0.95, 0.0], [-1.0, 0.9]], dtype=np.float32)
    for clip in (True, False):
        orc = scenes.draw_oracle(scene, clip_to_viewport=clip)
        hip = scenes.draw_hip(scene, clip_to_viewport=clip)
        assert_same(hip["color"], orc["color"], f"clip={clip}")
    assert not np.array_equal(scenes.draw_oracle(scene, clip_to_viewport=True)["color"],
                              scenes.draw_oracle(scene, clip_to_viewport=False)["color"])


def test_huge_and_far_away_quads(oracle):
    """Edges longer than the f32-exact limit (integer edge path), vertices far outside the guard band, quads fully off
    screen, zero-area quads and a NaN position."""
    f32 = np.float32
    scene = scenes.tiny_quads_scene(12, 700, 500, seed=23)
    inst = scene["instances"]
    inst[0] = [0.0, 0.0, 3.0, 3.0]            # covers everything, edges of 2100 px
    inst[1] = [0.5, -0.5, 40.0, 0.02]         # 28000 px wide sliver: clamped by the guard band
    inst[2] = [5.0, 5.0, 0.5, 0.5]            # off screen
    inst[3] = [0.1, 0.1, 0.0, 0.3]            # zero width
    inst[4] = [np.nan, 0.0, 0.1, 0.1]         # dropped
    inst[5] = [-0.3, 0.4, 1.7, 0.9]           # large, partly outside
    inst[6] = [0.2, 0.2, -0.6, 0.4]           # mirrored (negative extent): winding flips, still drawn
    scene["instances"] = inst.astype(f32)
    scene["blend"] = "premultiplied"
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "huge")
    assert orc["fragments"] > 700 * 500


@pytest.mark.parametrize("topology,blend", [("strip", "premultiplied"), ("list", "additive")])
def test_quads_in_space_are_clipped_like_the_oracle(oracle, topology, blend):
    """Triangles that cross the near and far planes, reach behind the eye (w <= 0) or far beyond the guard band are
    clipped (reference src/pipeline.rs:197-207: depth clipping on), not dropped: bit-exact against the oracle's clipper,
    on the first pass of a device and on one queued before its pair count is known."""
    scene = scenes.space_quads_scene(400, 640, 400, seed=41, blend=blend, topology=topology)
    orc = scenes.draw_oracle(scene)
    for twice in (False, True):
        hip = scenes.draw_hip(scene, twice=twice)
        assert_same(hip["color"], orc["color"], f"space/{topology}/{blend}/twice={twice}")
    # the scene does exercise the clipper: with every crossing triangle dropped it looks different
    inside = scenes.space_quads_scene(400, 640, 400, seed=41, blend=blend, topology=topology, half_range=(0.01, 0.02))
    assert orc["fragments"] > 50_000 and scenes.draw_oracle(inside)["fragments"] < orc["fragments"] // 10


def test_clipped_quads_with_infinite_and_enormous_coordinates(oracle):
    """Coordinates at the edge of f32 (products overflow to infinity, differences of infinities are NaN) go through the
    clipper without harm and give what the oracle gives."""
    scene = scenes.space_quads_scene(64, 320, 200, seed=46)
    inst = scene["instances"]
    inst[0] = [1e30, 0.0, 1.0, -2.0]
    inst[1] = [np.inf, 0.0, 0.5, -2.0]
    inst[2] = [0.0, -np.inf, 1.0, -1.0]
    inst[3] = [3e38, 3e38, 3e38, -1.0]
    inst[4] = [0.0, 0.0, 3e38, -2.0]
    inst[5] = [0.0, 0.0, 1e20, 1e20]
    inst[6] = [-1e38, 1e38, 1e19, -3e38]
    scene["instances"] = inst.astype(np.float32)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene, twice=True)
    assert_same(hip["color"], orc["color"], "space with infinities")


def test_clipped_quads_many_tiles_and_recording(oracle):
    """A few very large clipped quads over a 1920x1080 target (two sorting passes, hundreds of pairs per cluster, eight
    sub-records per tile) with the recording attachment."""
    scene = scenes.space_quads_scene(30, 1920, 1080, seed=42, half_range=(1.0, 4.0))
    scene["record"] = True
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene, twice=True)
    assert_same(hip["color"], orc["color"], "space large")
    assert_same(hip["recording"], orc["recording"], "space large recording")


def test_clipped_quads_multisampled(oracle):
    scene = scenes.space_quads_scene(120, 320, 200, seed=43)
    scene["samples"] = 4
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "space 4x")
    assert_same(hip["msaa"].reshape(200, 320 * 4, 4), orc["msaa"].reshape(200, 320 * 4, 4), "space 4x samples")


def test_no_instances_only_clears(oracle):
    scene = scenes.random_quads_scene(4, 16, 200, 100, seed=24, clear=(0.2, 0.4, 0.6, 1.0))
    scene["instances"] = np.zeros((0, 2), dtype=np.float32)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "empty")
    assert orc["fragments"] == 0 and (hip["color"] == hip["color"][0, 0]).all()


@pytest.mark.parametrize("filter", ["linear", "nearest"])
def test_projective_quads_uniforms_three_varyings(oracle, filter):
    """Per-vertex w: perspective-correct interpolation (division by the interpolated 1/w), mixed with affine quads."""
    scene = scenes.tilted_quads_scene(1500, 600, 400, seed=31, filter=filter)
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"tilted/{filter}")


@pytest.mark.parametrize("fmt", ["rgba_f16", "rgba_f32"])
@pytest.mark.parametrize("filter", ["linear", "nearest"])
def test_float_textures(oracle, fmt, filter):
    """RGBA16F / RGBA32F textures: texels are used as they are (no 1/255), values above 1 saturate in the ROP."""
    rng = np.random.default_rng(41)
    tex = rng.uniform(-0.2, 1.3, (19, 23, 4)).astype(np.float16 if fmt == "rgba_f16" else np.float32)
    tex[0, 0] = [np.inf, -np.inf, 0.0, 1.0]
    if fmt == "rgba_f16":
        tex[1, 1] = np.array([6e-8, 6e-5, 65504.0, 0.5], dtype=np.float16)        # subnormal, smallest normal, largest
    scene = scenes.random_quads_scene(2500, 20, 400, 300, seed=41, filter=filter, texture=tex)
    scene["texture_format"] = fmt
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], f"{fmt}/{filter}")


def test_texture_array_layers_and_sub_rectangle_upload(oracle):
    """A 3-layer texture, the sampled layer chosen by a uniform; then part of that layer replaced with
    set_part_of_texture (reference src/renderer.rs:156-162, src/texture.rs:45-56)."""
    rng = np.random.default_rng(42)
    layers, th, tw = 3, 16, 24
    tex = rng.integers(0, 256, (layers * th, tw, 4), dtype=np.uint8)
    tex[..., 3] = 255
    for layer in range(layers):
        scene = scenes.random_quads_scene(800, 30, 320, 200, seed=42, texture=tex)
        scene.update(shader="layered", texture_layers=layers, uniform=np.array([float(layer)], dtype=np.float32))
        assert_same(scenes.draw_hip(scene)["color"], scenes.draw_oracle(scene)["color"], f"layer {layer}")
    # sub-rectangle upload into layer 1
    scene = scenes.random_quads_scene(800, 30, 320, 200, seed=43, texture=tex)
    scene.update(shader="layered", texture_layers=layers, uniform=np.array([1.0], dtype=np.float32))
    patch = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    hs = scenes.HipScene(scene)
    hs.renderer.set_part_of_texture(hs.pipeline, (0, 2), (3, 4, 1), (7, 5), patch)
    hs.draw()
    got = hs.read_color(); hs.close()
    tex2 = tex.copy()
    tex2[th + 4:th + 9, 3:10] = patch
    want = scenes.draw_oracle(dict(scene, texture=tex2))["color"]
    assert_same(got, want, "sub-rectangle")


def test_two_pass_with_target_bound_as_texture(oracle):
    """tools/render_strips.py at world size 1: pass 2 samples the texture pass 1 rendered into, bound directly (no copy);
    the torch view of the texture's device memory used by the strip exchange aliases the real texels."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import render_strips
    S = 160
    image, _ = render_strips.render_two_pass(S, 300, 24, seed=3)
    s1 = scenes.random_quads_scene(300, 24, S, S, seed=3)
    a_img = scenes.draw_oracle(s1)["color"]
    s2 = dict(shader="hello", width=S, height=S, target="texture", position=scenes.UNIT_POSITION, uv=scenes.QUAD_UV,
              instances=np.zeros((1, 2), dtype=np.float32), texture=a_img, texture_format="rgba_u8", filter="linear",
              blend="premultiplied", topology="strip", clear=(0., 0., 0., 1.), viewport=None, record=False)
    assert_same(image, scenes.draw_oracle(s2)["color"], "two-pass")


def test_texture_memory_as_torch_tensor(oracle):
    import torch
    from renderer_amd import Renderer, partition
    r = Renderer(64, 64)
    tex = r.texture(40, 24, 1, Renderer.nearest_filtering(), Renderer.rgba_u8(), True, True, False)
    pixels = np.random.default_rng(1).integers(0, 256, (24, 40, 4), dtype=np.uint8)
    tex.set_data((0, 0, 0), (0, 0), pixels)
    r.synchronize()
    t = partition.texture_as_tensor(tex, "cuda:0")
    assert t.shape == (24, 40, 4) and np.array_equal(t.cpu().numpy(), pixels)
    t[3:5] = 7                                   # writes land in the texture
    torch.cuda.synchronize()
    assert (tex.read()[3:5] == 7).all()
    r.close()


def test_render_thread_proxy_runs_the_quads_example(oracle):
    """reference examples/quads_render_thread.rs shape: every call goes through the RenderThread proxy (index handles,
    messages executed in order on a worker thread that owns the Renderer); recorded frames equal the oracle's."""
    from renderer_amd import RenderThread, Renderer
    W, H = 320, 180
    rt = RenderThread(W, H)
    vert, frag = scenes.shader_code("hello")
    a_position, a_tex_coord, i_offset = rt.attribute(0, 2), rt.attribute(1, 2), rt.instanced()
    t_texture = rt.texture(256, 256, 1, Renderer.linear_filtering(), Renderer.rgba_u8(), False, False, True)
    program = rt.program(vert, frag, [a_position, a_tex_coord], [i_offset], [], [(t_texture, Renderer.visible_to_fragment_shader())])
    pipeline = rt.pipeline(program, Renderer.pre_multiplied_blend(), Renderer.triangle_strip_primitive(), 1, [RenderThread.screen_target()])
    clear = Renderer.clear_color(0., 0., 0., 1.)
    rt.set_attribute(pipeline, 0, scenes.QUAD_POSITION)
    rt.set_attribute(pipeline, 1, scenes.QUAD_UV)
    rt.set_texture(pipeline, (0, 1), [scenes.load_letter_f().reshape(-1)])
    frames = []

    def on_frame(video_frame):                   # runs on the render thread
        got = []
        video_frame.image_data.bytes_fn(lambda b: got.append(np.array(b, copy=True)))
        frames.append((video_frame.frame_number, video_frame.width, video_frame.height, video_frame.padded_bytes_per_row, got[0]))
        video_frame.drop()

    rt.start_recording([pipeline], clear, 100., on_frame)
    rt.flush()
    n = 4
    for k in range(1, n + 1):
        rt.set_instanced(pipeline, (0, 0), scenes.example_positions(k))
        rt.render(pipeline, clear, rt.viewport(1., 1.), (2, 4))
        rt.finish_frame()
    rt.render(pipeline, clear, rt.viewport(1., 1.), (2, 4)); rt.finish_frame()       # frames are delivered a frame later at the earliest
    rt.synchronize()
    assert "gfx" in rt.adapter_info()
    rt.stop_recording([pipeline])
    rt.join()
    assert [f[0] for f in frames[:n]] == list(range(1, n + 1))
    vp = rt.viewport(1., 1.)
    x0, y0 = int(np.ceil(vp.margin_x)), int(np.ceil(vp.margin_y))
    for number, w, h, pitch, data in frames[:n]:
        want = scenes.draw_oracle(scenes.example_scene(number, width=W, height=H, record=True))["recording"][y0:y0 + h, x0:x0 + w]
        got = data.reshape(h, pitch)[:, :w * 4].reshape(h, w, 4)
        assert np.array_equal(got, want), f"frame {number}"


# ---- multisampling (reference src/pipeline.rs:164-187, 209-211; src/render_pass.rs:76-81) --------------------------
@pytest.mark.parametrize("samples", [2, 4, 8])
@pytest.mark.parametrize("blend", ["premultiplied", "additive"])
def test_multisampled_random_quads(oracle, samples, blend):
    """Resolved target AND every stored sample equal the oracle's; quads at random sub-pixel offsets, deep overlap."""
    scene = scenes.random_quads_scene(6000, 12, 333, 217, seed=40 + samples, blend=blend)
    scene["samples"] = samples
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["msaa"].reshape(217, 333 * samples, 4), orc["msaa"].reshape(217, 333 * samples, 4), f"samples {samples}x {blend}")
    assert_same(hip["color"], orc["color"], f"resolve {samples}x {blend}")
    # edges are antialiased: the resolved image has values the single-sampled one cannot have on a black/opaque scene
    assert len(np.unique(hip["color"])) > 8


def test_multisampled_projective_and_screen_target(oracle):
    scene = scenes.tilted_quads_scene(800, 300, 200, seed=9)
    scene["samples"] = 4
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "tilted 4x")
    scene = scenes.example_scene(60, 640, 360)           # BGRA8 screen target with a letterbox viewport
    scene["samples"] = 4
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert_same(hip["color"], orc["color"], "screen 4x")


@pytest.mark.parametrize("fmt,samples", [("rgba_f32", 8), ("rgba_f32", 2), ("rgba_f16", 4), ("r_u8", 4)])
def test_multisampled_targets_of_every_format(oracle, fmt, samples):
    """The pipeline's multisampled texture is shaped after its target, whatever the format (reference
    src/pipeline.rs:164-176): float samples blend unclamped, their mean (f32 sum in sample order) is the resolved texel."""
    scene = scenes.tilted_quads_scene(300, 200, 136, seed=19, blend="additive")
    scene["target_format"] = fmt
    scene["samples"] = samples
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    bpt = scenes.TARGET_FORMATS[fmt][1]
    assert_same(hip["color"], orc["color"], f"{fmt} {samples}x")
    assert_same(hip["msaa"].reshape(136, 200 * samples, bpt), orc["msaa"].reshape(136, 200 * samples, bpt), f"{fmt} {samples}x samples")
    if fmt == "rgba_f32":
        assert orc["color"].view(np.float32).max() > 1.5        # additive float samples are not clamped


def test_multisampled_load_reads_the_pipelines_msaa_texture(oracle):
    """LoadOp::Load on a multisampled pipeline continues from ITS multisampled texture, not from the target: the
    target's previous contents are overwritten by the resolve."""
    scene = scenes.random_quads_scene(500, 20, 256, 128, seed=5)
    scene["samples"] = 4
    first = scenes.draw_oracle(scene)
    second_scene = dict(scene, clear=None, instances=scene["instances"][::-1].copy())
    junk = np.full((128, 256, 4), 77, dtype=np.uint8)
    second = scenes.draw_oracle(second_scene, msaa=first["msaa"], initial=junk)
    hs = scenes.HipScene(scene)
    try:
        hs.draw()
        assert_same(hs.read_color(), first["color"], "first pass")
        hs.set_initial(junk)
        hs.clear = None
        hs.renderer.set_instanced(hs.pipeline, (0, 0), second_scene["instances"])
        hs.draw()
        assert_same(hs.read_color(), second["color"], "second pass (Load)")
        assert_same(hs.pipeline.msaa_texture.read().reshape(128, 1024, 4), second["msaa"].reshape(128, 1024, 4), "samples after Load")
        # dropping back to one sample renders straight into the target again
        hs.renderer.set_msaa_samples(hs.pipeline, 1)
        hs.clear = hs.renderer.clear_color(0., 0., 0., 1.)
        hs.draw()
        plain = scenes.draw_oracle(dict(second_scene, samples=1, clear=(0., 0., 0., 1.)))
        assert_same(hs.read_color(), plain["color"], "back to single-sampled")
    finally:
        hs.close()


def test_multisampled_pass_rejects_a_recording_attachment(oracle):
    from renderer_amd import RendererError
    scene = scenes.example_scene(1, 320, 180, record=True)
    scene["samples"] = 4
    with pytest.raises(RendererError, match="samples|exactly one colour attachment"):
        scenes.draw_hip(scene)


# ---- render targets of every format (reference src/target.rs:8-13, src/format.rs:3-30) --------------------------------
@pytest.mark.parametrize("target_format", ["r_u8", "bgra_u8", "rgba_f16", "rgba_f32"])
@pytest.mark.parametrize("blend", ["premultiplied", "additive"])
def test_render_target_formats(oracle, target_format, blend):
    """Bit-exact target bytes: R8 keeps the red channel, float targets blend unclamped in f32 (RGBA16F rounded to f16
    after every fragment) - additive layers go far above 1.0 here."""
    scene = scenes.random_quads_scene(9000, 14, 333, 217, seed=21, blend=blend)
    scene["target_format"] = target_format
    orc = scenes.draw_oracle(scene)
    hip = scenes.draw_hip(scene)
    assert hip["color"].shape == orc["color"].shape
    assert np.array_equal(hip["color"], orc["color"]), f"{target_format}/{blend}: {(hip['color'] != orc['color']).any(axis=2).sum()} texels differ"
    if target_format == "rgba_f32" and blend == "additive":
        assert orc["color"].view(np.float32).max() > 2.0


def test_float_target_load_and_recording_attachment_together(oracle):
    """A float target and the 8-bit recording attachment in one pass (mixed pixel sizes), continuing from existing
    float contents (LoadOp::Load) with a non-trivial clear colour on the recording attachment."""
    scene = scenes.example_scene(60, 480, 270, record=True)
    scene["target"] = "texture"
    scene["target_format"] = "rgba_f16"
    scene["clear"] = None
    scene["record_clear"] = (0.25, 0.5, 0.125, 1.0)
    rng = np.random.default_rng(1)
    initial = rng.uniform(-2.0, 2.0, (270, 480, 4)).astype(np.float16).view(np.uint8).reshape(270, 480, 8)
    orc = scenes.draw_oracle(scene, initial=initial)
    hip = scenes.draw_hip(scene, initial=initial)
    assert np.array_equal(hip["color"], orc["color"])
    assert_same(hip["recording"], orc["recording"], "recording beside a float target")


def test_float_target_sampled_by_a_second_pass(oracle):
    """Pass 1 renders additive layers into an RGBA32F texture; pass 2 samples it (values above 1.0 survive) into RGBA8."""
    from renderer_amd import Renderer
    first = scenes.random_quads_scene(3000, 24, 256, 256, seed=8, blend="additive")
    first["target_format"] = "rgba_f32"
    o1 = scenes.draw_oracle(first)
    hdr = o1["color"].view(np.float32).reshape(256, 256, 4)
    second = scenes.random_quads_scene(50, 64, 320, 200, seed=9, texture=(hdr * np.float32(0.25)).astype(np.float32))
    second["texture_format"] = "rgba_f32"
    o2 = scenes.draw_oracle(second)
    h1 = scenes.draw_hip(first)
    assert np.array_equal(h1["color"], o1["color"])
    second_hip = dict(second, texture=(h1["color"].view(np.float32).reshape(256, 256, 4) * np.float32(0.25)).astype(np.float32))
    assert_same(scenes.draw_hip(second_hip)["color"], o2["color"], "sampling the float target's contents")


def test_two_recorded_pipelines_share_the_recording_texture(oracle):
    """reference src/renderer.rs:178-188: every recorded pipeline gets the recording attachment, the first use of a
    frame clears it, later pipelines load it, and only the Last one copies the frame out."""
    from renderer_amd import Renderer
    W, H = 480, 270
    first = scenes.random_quads_scene(300, 40, W, H, seed=1, target="screen")
    second = scenes.tiny_quads_scene(4000, W, H, seed=2)
    second["target"] = "screen"
    o1 = scenes.draw_oracle(dict(first, record=True, record_clear=(0.1, 0.2, 0.3, 1.0)))
    o2 = scenes.draw_oracle(dict(second, record=True, record_clear=None, clear=None), initial=o1["color"], initial_recording=o1["recording"])
    r = Renderer(W, H)
    try:
        a, b = scenes.HipScene(first, renderer=r), scenes.HipScene(second, renderer=r)
        frames = []
        r.start_recording([a.pipeline, b.pipeline], Renderer.clear_color(0.1, 0.2, 0.3, 1.0), 500., frames.append)
        for _ in range(2):
            r.render(a.pipeline, a.clear, None, a.count)
            r.render(b.pipeline, None, None, b.count)
            r.finish_frame()
        r.synchronize()
        r.recorder.process_mapped_buffers()
        assert_same(r.read_screen(), o2["color"], "screen after both pipelines")
        assert_same(r.recorder.recording_texture.read(), o2["recording"], "recording after both pipelines")
        assert [f.frame_number for f in frames] == [1, 2]            # one capture per frame, after the Last pipeline
        got = []
        frames[0].image_data.bytes_fn(lambda b: got.append(np.array(b).reshape(H, frames[0].padded_bytes_per_row)))
        assert_same(got[0][:, :W * 4].reshape(H, W, 4), o2["recording"], "captured frame")
    finally:
        r.close()


def test_captured_frame_of_a_loaded_sparse_pass_with_margins(oracle):
    """The capture copy behind the Last recorded pipeline (reference src/video_recorder.rs:107-128) is performed by the
    tile kernel as it writes its tiles back.  A pass that loads the recording texture and touches few tiles still has
    to deliver every texel of the viewport's region, rows padded to 256 bytes with zeros, in the form that waits for the
    pair count and in the queued one."""
    from renderer_amd import Renderer, _ffi
    W, H = 333, 217
    scene = scenes.random_quads_scene(12, 20, W, H, seed=31, target="screen")
    scene.update(record=True, record_clear=None, viewport=(1.6, 1.0))
    before = np.random.default_rng(32).integers(0, 255, (H, W, 4), dtype=np.uint8)
    r = Renderer(W, H)
    try:
        hs = scenes.HipScene(scene, renderer=r)
        tex = r.recorder.recording_texture
        _ffi.check(_ffi.lib().rh_texture_write(tex.handle, 0, 0, 0, W, H, before.ctypes.data, before.nbytes, W * 4))
        recording = before
        for k in range(3):
            want = scenes.draw_oracle(scene, initial_recording=recording)
            recording = want["recording"]
            hs.draw()
            r.synchronize()
            r.recorder.process_mapped_buffers()
            frame = hs.frames[k]
            got = []
            frame.image_data.bytes_fn(lambda b: got.append(np.array(b).reshape(frame.height, frame.padded_bytes_per_row)))
            x0, y0 = int(np.ceil(hs.viewport.margin_x)), int(np.ceil(hs.viewport.margin_y))
            assert (x0 > 0 or y0 > 0) and frame.padded_bytes_per_row > frame.width * 4
            # the copied extent is the texture minus the rounded-up margins on both sides (src/video_recorder.rs:124-125),
            # which can be a row or column short of the frame's size; what is not copied stays zero
            ew, eh = W - 2 * x0, H - 2 * y0
            assert_same(got[0][:eh, :ew * 4].reshape(eh, ew, 4), recording[y0:y0 + eh, x0:x0 + ew], f"captured frame {k + 1}")
            assert not got[0][:, ew * 4:].any() and not got[0][eh:].any(), "row padding and rows not copied"
            assert_same(tex.read(), recording, f"recording texture after frame {k + 1}")
    finally:
        r.close()


def test_twenty_binding_slots_over_five_bind_groups(oracle):
    """The reference numbers binding slots flat, four to a bind group, eight groups (src/pipeline.rs:25-27, 90-131;
    Renderer::uniform_index / texture_index, src/renderer.rs:356-378).  The `crowd` program binds one instanced buffer,
    seven uniforms and six textures with their samplers: slots 0..19, bind groups 0..4, uniforms and textures beyond the
    sixteenth slot.  Bit-exact against the oracle's restatement of the same program."""
    from oracle import loader as O
    from renderer_amd import Renderer
    W, H, N = 320, 200, 400
    rng = np.random.default_rng(77)
    scene = scenes.random_quads_scene(N, 24, W, H, seed=78)
    sizes = [(32, 32), (64, 32), (32, 64), (96, 32), (16, 16), (256, 256)]
    filters = ["linear", "nearest", "linear", "linear", "nearest", "linear"]
    texels = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for (w, h) in sizes]
    texels[5] = scenes.load_letter_f()
    gain = np.array([0.35], dtype=np.float32)
    weights = [rng.uniform(0.05, 0.6, 4).astype(np.float32) for _ in range(6)]
    r = Renderer(W, H)
    try:
        vert, frag = scenes.shader_code("crowd")
        a_position, a_tex_coord, i_offset = r.attribute(0, 2), r.attribute(1, 2), r.instanced()
        uniforms = [(r.uniform(), Renderer.visible_to_fragment_shader()) for _ in range(7)]
        textures = []
        for (w, h), f in zip(sizes, filters):
            flt = Renderer.linear_filtering() if f == "linear" else Renderer.nearest_filtering()
            textures.append((r.texture(w, h, 1, flt, Renderer.rgba_u8(), False, False, True), Renderer.visible_to_fragment_shader()))
        program = r.program(vert, frag, [a_position, a_tex_coord], [i_offset], uniforms, textures)
        target = r.texture(W, H, 1, Renderer.nearest_filtering(), Renderer.rgba_u8(), True, True, False)
        pipeline = r.pipeline(program, Renderer.pre_multiplied_blend(), Renderer.triangle_strip_primitive(), 1, [Renderer.texture_target(target)])
        r.set_attribute(pipeline, 0, scene["position"]); r.set_attribute(pipeline, 1, scene["uv"])
        r.set_instanced(pipeline, (0, 0), scene["instances"])
        r.set_uniform(pipeline, (0, 1), gain)
        for i in range(6):
            flat = 2 + i
            r.set_uniform(pipeline, (flat // 4, flat % 4), weights[i])
            flat = 8 + 2 * i
            r.set_texture(pipeline, (flat // 4, flat % 4), [texels[i].reshape(-1)])
        assert (8 + 2 * 5 + 1) // 4 == 4                      # the last sampler sits in the fifth bind group
        clear = Renderer.clear_color(0.1, 0.2, 0.3, 1.0)
        for _ in range(2):                                    # (the second pass: queued before its pair count is known)
            r.render(pipeline, clear, None, (N, 4))
            r.finish_frame()
        got = target.read()
    finally:
        r.close()
    want = np.zeros((H, W, 4), dtype=np.uint8)
    O.render(shader=O.SHADER_CROWD, topology=O.TOPO_TRIANGLE_STRIP, vertex_count=4, instance_count=N,
             attrs=[(scene["position"], 2), (scene["uv"], 2)], storage=[scene["instances"]], uniforms=[gain] + weights,
             textures=[(t, t.shape[1], t.shape[0], 1, O.FMT_RGBA_U8, O.FILTER_LINEAR if f == "linear" else O.FILTER_NEAREST) for t, f in zip(texels, filters)],
             targets=[dict(data=want, format=O.FMT_RGBA_U8, clear=(0.1, 0.2, 0.3, 1.0))], blend=(O.BF_ONE, O.BF_ONE_MINUS_SRC_ALPHA))
    assert_same(got, want, "twenty slots")
    assert len(np.unique(got)) > 32
