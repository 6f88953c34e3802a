"""CPU tests of the oracle: hand-derived known answers for the rasterisation rules, the committed golden vectors,
and the LZ4 restatement against the system liblz4 (1.9.3).  No GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
f32 = np.float32


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ---- fixtures of the reference ------------------------------------------------------------------------------------------
def test_letter_f_fixture_hashes():
    # hashes of the reference's examples/quads/letter_f.png: file, decoded RGBA, and after the example's premultiply
    from PIL import Image
    path = os.path.join(GOLDEN, "letter_f.png")
    assert sha(open(path, "rb").read()) == "9f08fa8630b12d99696ec9c4ccdd3a2f7e3d37f50eb7fdded79ec6cff253191f"
    raw = np.array(Image.open(path).convert("RGBA"), dtype=np.uint8)
    assert raw.shape == (256, 256, 4) and (raw[..., 3] == 204).all()
    assert sha(raw.tobytes()) == "d6ff14de46e2a1e9e0ea27ca504011a93c6a3adcc4a3011fc05f0cb3957d09e9"
    assert tuple(raw[0, 0]) == (255, 0, 0, 204) and tuple(raw[128, 128]) == (255, 238, 0, 204) and tuple(raw[255, 255]) == (255, 0, 4, 204)
    pre = scenes.load_letter_f()
    assert sha(pre.tobytes()) == "aad1072597f3cc12240a9ad1724781e5bd0480ada29681ebbecc2447b5aded47"


def test_example_positions_golden():
    gold = json.load(open(os.path.join(GOLDEN, "example_positions.json")))
    for k, vals in gold.items():
        assert np.array_equal(scenes.example_positions(int(k)).reshape(-1), np.array(vals, dtype=f32))
    # frame 1 by hand: 0.3 + 0.015, -0.3 + 0.01, -0.5 + 0.005, -0.1 + 0.02 in f32
    p = scenes.example_positions(1).reshape(-1)
    assert p[0] == f32(f32(0.3) + f32(0.015)) and p[3] == f32(f32(-0.1) + f32(0.02))


def test_viewport_letterbox():
    from renderer_amd.viewport import Viewport
    v = Viewport(1., 1., 1920., 1080.)
    assert (v.width, v.height, v.margin_x, v.margin_y) == (1080.0, 1080.0, 420.0, 0.0)
    v = Viewport(16., 9., 1000., 1000.)
    assert v.width == 1000.0 and v.margin_x == 0.0 and abs(v.height - 562.5) < 1e-3 and abs(v.margin_y - 218.75) < 1e-3


def test_oracle_frames_golden(oracle):
    gold = json.load(open(os.path.join(GOLDEN, "oracle_frames.json")))
    for k in ("1", "60"):
        o = scenes.draw_oracle(scenes.example_scene(int(k), record=True), threads=2)
        assert sha(o["color"].tobytes()) == gold[k]["screen_bgra_sha256"]
        assert sha(o["recording"].tobytes()) == gold[k]["recording_rgba_sha256"]
        assert o["fragments"] == gold[k]["fragments"] == 2 * 108 * 108


# ---- rasterisation rules --------------------------------------------------------------------------------------------------
def flat_texture(rgba):
    return np.tile(np.array(rgba, dtype=np.uint8), (4, 4, 1))


def px_quad_scene(x0, y0, x1, y1, W=32, H=32, blend="additive", rgba=(10, 20, 30, 40), n=1, clear=(0., 0., 0., 0.)):
    """One axis-aligned quad with corners at pixel coordinates (x0, y0)-(x1, y1) (y down), strip order of the example."""
    def ndc(x, y):
        return [x / W * 2 - 1, 1 - y / H * 2]
    pos = np.array([ndc(x0, y1), ndc(x0, y0), ndc(x1, y1), ndc(x1, y0)], dtype=f32)
    return dict(shader="hello", width=W, height=H, target="texture", position=pos, uv=scenes.QUAD_UV,
                instances=np.zeros((n, 2), dtype=f32), texture=flat_texture(rgba), texture_format="rgba_u8",
                filter="nearest", blend=blend, topology="strip", clear=clear, viewport=None, record=False)


def test_top_left_rule_on_pixel_centres(oracle):
    # edges exactly on pixel centres: left/top centres are inside, right/bottom centres are outside
    o = scenes.draw_oracle(px_quad_scene(10.5, 5.5, 20.5, 9.5), threads=1)
    cov = o["color"][..., 3] > 0
    ys, xs = np.nonzero(cov)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (10, 19, 5, 8)
    assert cov.sum() == 10 * 4 == o["fragments"]


def test_shared_diagonal_covered_exactly_once(oracle):
    # additive blend of a constant colour: every covered pixel must hold exactly one layer, including the diagonal
    for (x0, y0, x1, y1) in [(3.0, 3.0, 29.0, 29.0), (2.25, 4.75, 30.5, 17.125), (0.0, 0.0, 32.0, 32.0)]:
        o = scenes.draw_oracle(px_quad_scene(x0, y0, x1, y1), threads=1)
        img = o["color"]
        covered = img[..., 3] > 0
        assert (img[covered] == np.array([10, 20, 30, 40], dtype=np.uint8)).all()
        w = int(np.ceil(x1 - 0.5) - np.ceil(x0 - 0.5)); h = int(np.ceil(y1 - 0.5) - np.ceil(y0 - 0.5))
        assert covered.sum() == w * h == o["fragments"]


def test_both_windings_drawn(oracle):
    s = px_quad_scene(4.0, 4.0, 12.0, 12.0)
    a = scenes.draw_oracle(s, threads=1)["color"]
    s["position"] = s["position"][[2, 3, 0, 1]].copy()      # mirrored strip: opposite winding
    s["uv"] = s["uv"][[2, 3, 0, 1]].copy()
    b = scenes.draw_oracle(s, threads=1)["color"]
    assert np.array_equal(a, b) and (a[..., 3] > 0).sum() == 64


def test_triangle_list_topology(oracle):
    s = px_quad_scene(4.0, 4.0, 12.0, 12.0)
    strip = scenes.draw_oracle(s, threads=1)["color"]
    s["position"] = s["position"][[0, 1, 2, 1, 3, 2]].copy()
    s["uv"] = s["uv"][[0, 1, 2, 1, 3, 2]].copy()
    s["topology"] = "list"
    assert np.array_equal(scenes.draw_oracle(s, threads=1)["color"], strip)


def test_bilinear_texel_centres_and_clamp(oracle):
    # a 2x2 texture drawn over a 4x4 target: pixel centres sample at u = 1/8, 3/8, 5/8, 7/8 -> unnormalised
    # -0.25, 0.25, 0.75, 1.25 -> weights 0 (clamped), .25, .75, clamped
    tex = np.zeros((2, 2, 4), dtype=np.uint8)
    tex[0, 0] = (0, 0, 0, 255); tex[0, 1] = (200, 0, 0, 255); tex[1, 0] = (0, 100, 0, 255); tex[1, 1] = (200, 100, 0, 255)
    s = px_quad_scene(0.0, 0.0, 4.0, 4.0, W=4, H=4, blend="additive")
    s["texture"] = tex; s["filter"] = "linear"
    img = scenes.draw_oracle(s, threads=1)["color"]
    assert list(img[0, :, 0]) == [0, 50, 150, 200]          # red ramps left to right
    assert list(img[:, 0, 1]) == [0, 25, 75, 100]           # green ramps top to bottom
    s["filter"] = "nearest"
    img = scenes.draw_oracle(s, threads=1)["color"]
    assert list(img[0, :, 0]) == [0, 0, 200, 200] and list(img[:, 0, 1]) == [0, 0, 100, 100]


def reference_blend(src, dst, mode):
    """numpy binary32 restatement of out = src*SF + dst*DF, re-quantised: computed on the 0..255 scale."""
    src = src.astype(f32); a = src[..., 3:4]
    df = f32(1.0) if mode == "additive" else (f32(1.0) - a)
    o = (dst.astype(f32) * df).astype(f32)
    o = (o.astype(np.float64) + (src * f32(255.0)).astype(f32).astype(np.float64))      # fma: one rounding
    o = np.clip(o.astype(f32), f32(0.0), f32(255.0))
    return np.rint(o).astype(np.uint8)


@pytest.mark.parametrize("mode", ["premultiplied", "additive"])
def test_blend_against_numpy_formula(oracle, mode):
    # one flat quad over a 256x256 target pre-filled with every (dst, dst2) byte pair; src sweeps over 8 colours
    rng = np.random.default_rng(5)
    W = H = 256
    dst = np.zeros((H, W, 4), dtype=np.uint8)
    dst[..., 0] = np.arange(256)[None, :]; dst[..., 1] = np.arange(256)[:, None]
    dst[..., 2] = rng.integers(0, 256, (H, W)); dst[..., 3] = rng.integers(0, 256, (H, W))
    for _ in range(8):
        rgba = tuple(int(v) for v in rng.integers(0, 256, 4))
        s = px_quad_scene(0.0, 0.0, float(W), float(H), W=W, H=H, blend=mode, rgba=rgba, clear=None)
        out = scenes.draw_oracle(s, threads=2, initial=dst)["color"]
        src = (np.array(rgba, dtype=f32) * f32(1.0 / 255.0)).astype(f32)          # sampler output
        expect = reference_blend(np.broadcast_to(src, dst.shape), dst, mode)
        assert np.array_equal(out, expect), (rgba, np.argwhere(out != expect)[:3])


def test_order_dependence(oracle):
    s = scenes.random_quads_scene(500, 24, 160, 120, seed=11)
    a = scenes.draw_oracle(s, threads=1)["color"]
    s2 = dict(s); s2["instances"] = s["instances"][::-1].copy()
    b = scenes.draw_oracle(s2, threads=1)["color"]
    assert not np.array_equal(a, b)                      # alpha blending depends on primitive order
    # additive without saturation does not (dim the texture so that sums stay below 255)
    s["blend"] = s2["blend"] = "additive"
    s["clear"] = s2["clear"] = (0., 0., 0., 0.)
    s["filter"] = s2["filter"] = "nearest"              # integer-valued sources: per-layer rounding cannot differ
    s["texture"] = s2["texture"] = (scenes.load_letter_f() // 64).astype(np.uint8)
    a = scenes.draw_oracle(s, threads=1)["color"]; b = scenes.draw_oracle(s2, threads=1)["color"]
    assert a.max() < 255 and np.array_equal(a, b)


def test_thread_count_invariance(oracle):
    s = scenes.random_quads_scene(3000, 16, 320, 200, seed=4)
    a = scenes.draw_oracle(s, threads=1)
    b = scenes.draw_oracle(s, threads=7)
    assert np.array_equal(a["color"], b["color"]) and a["fragments"] == b["fragments"]


def test_viewport_clip_switch(oracle):
    # the example's quads poke outside the letterboxed viewport: strict clipping removes those fragments
    s = scenes.example_scene(600)
    s["instances"] = np.array([[0.95, 0.0], [-0.95, 0.5]], dtype=f32)
    strict = scenes.draw_oracle(s, clip_to_viewport=True)
    loose = scenes.draw_oracle(s, clip_to_viewport=False)
    assert strict["fragments"] < loose["fragments"] == 2 * 108 * 108
    img = strict["color"]
    assert (img[:, :420, :3] == 0).all() and (img[:, 1500:, :3] == 0).all()
    assert (loose["color"][:, 1500:, :3] != 0).any()


def test_load_op_load_keeps_contents(oracle):
    s = px_quad_scene(8.0, 8.0, 16.0, 16.0, clear=None)
    init = np.full((32, 32, 4), 7, dtype=np.uint8)
    out = scenes.draw_oracle(s, threads=1, initial=init)["color"]
    assert (out[0, 0] == 7).all() and tuple(out[10, 10]) == (17, 27, 37, 47)


# ---- LZ4 frame codec ----------------------------------------------------------------------------------------------------------
def kat_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod.kat_inputs()


def test_lz4_restatement_matches_golden_streams(oracle):
    gold = json.load(open(os.path.join(GOLDEN, "lz4_kat.json")))
    assert gold["liblz4_version"] == 10903
    for name, writes in kat_cases().items():
        g = gold["cases"][name]
        assert sha(b"".join(writes)) == g["input_sha256"], "generator drifted"
        s = oracle.Lz4OracleStream()
        for w in writes:
            s.write(w)
        stream = s.finish()
        assert len(stream) == g["stream_len"] and sha(stream) == g["stream_sha256"], name
        if "stream_hex" in g:
            assert stream.hex() == g["stream_hex"]
        assert stream[:7].hex() == "04224d184040c0"          # magic, FLG, BD, HC
        dec, consumed, complete = oracle.lz4_decompress_frame(stream, sum(len(w) for w in writes) + 8)
        assert dec == b"".join(writes) and complete and consumed == len(stream)


def test_lz4_small_write_is_stored_block(oracle):
    s = oracle.Lz4OracleStream(); s.write(b"\x00" * 8); out = s.finish()
    assert len(out) == 7 + 12 + 4 and out[7:11] == (8 | 0x80000000).to_bytes(4, "little")


def test_lz4_restatement_matches_system_liblz4_random(oracle):
    rng = np.random.default_rng(99)
    for trial in range(40):
        a = oracle.Lz4OracleStream(); b = oracle.LibLz4Stream(0)
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.choice([5, 13, 64, 5000, 65536, 65541, 150000]))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            elif kind == 1:
                data = (rng.integers(0, 3, n, dtype=np.uint8) * 100).tobytes()
            else:
                data = (rng.integers(0, 256, 4, dtype=np.uint8).tobytes() * (n // 4 + 1))[:n]
            a.write(data); b.write(data)
        assert a.finish() == b.finish(), trial


def test_lz4_truncated_tail_is_tolerated(oracle):
    s = oracle.Lz4OracleStream()
    s.write(b"abcdefgh" * 4000); s.write(b"x" * 70000)
    stream = s.finish()
    dec, consumed, complete = oracle.lz4_decompress_frame(stream[:-9], 200000)
    assert not complete and (b"abcdefgh" * 4000 + b"x" * 70000).startswith(dec) and len(dec) >= 32000


def test_perspective_correct_interpolation_against_float64(oracle):
    """One projective quad (per-vertex w) textured with a ramp: the texel the oracle samples at every covered pixel must
    be the one perspective-correct interpolation gives in float64 (screen-space barycentrics divided by w)."""
    W, H = 96, 64
    ramp = np.zeros((1, 256, 4), dtype=np.uint8)
    ramp[0, :, 0] = np.arange(256); ramp[0, :, 3] = 255
    scene = scenes.tilted_quads_scene(1, W, H, seed=0, blend="additive", filter="nearest")
    cx, cy, half, tilt = 0.05, -0.1, 0.7, 0.6
    scene["instances"] = np.array([[cx, cy, half, tilt]], dtype=f32)
    scene["uniform"] = np.array([1, 1, 1, 1, 1, 1], dtype=f32)
    scene["texture"] = ramp
    scene["clear"] = (0., 0., 0., 0.)
    out = scenes.draw_oracle(scene)["color"]
    corners = scenes.UNIT_POSITION.astype(np.float64)
    w = 1.0 + corners[:, 0] * tilt
    ndc = np.stack([corners[:, 0] * half + cx, corners[:, 1] * half + cy], axis=1)
    sx = (ndc[:, 0] * 0.5 + 0.5) * W; sy = (0.5 - ndc[:, 1] * 0.5) * H
    u_attr = scenes.QUAD_UV[:, 0].astype(np.float64)
    shade_attr = corners[:, 1] * 0.25 + 0.75
    checked = 0
    for j in range(H):
        for i in range(W):
            if out[j, i, 3] == 0:
                continue
            px, py = i + 0.5, j + 0.5
            ok = False
            for tri in ((0, 1, 2), (1, 3, 2)):
                (x0, y0), (x1, y1), (x2, y2) = [(sx[k], sy[k]) for k in tri]
                den = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2)
                m0 = ((y1 - y2) * (px - x2) + (x2 - x1) * (py - y2)) / den
                m1 = ((y2 - y0) * (px - x2) + (x0 - x2) * (py - y2)) / den
                m = np.array([m0, m1, 1 - m0 - m1])
                if (m < -1e-6).any():
                    continue
                lam = m / w[list(tri)]; lam /= lam.sum()
                u = float((lam * u_attr[list(tri)]).sum()); shade = float((lam * shade_attr[list(tri)]).sum())
                texel = min(255, max(0, int(np.floor(u * 256))))
                want_r = texel * shade            # (texel/255 * 1) * shade * 255
                if abs(float(out[j, i, 0]) - want_r) <= 1.0 + 1.0 * shade:       # one texel either way + rounding
                    ok = True
            assert ok, f"pixel ({i}, {j}) = {out[j, i]}"
            checked += 1
    assert checked > 1500


# ---- clipping ---------------------------------------------------------------------------------------------------------------
def test_quad_reaching_behind_the_eye_is_cut_at_w_zero(oracle):
    """The unit quad under the homography (x, y, 0, 1 + 2x): its left part lies behind the eye (w < 0 for x < -1/2).
    What is in front projects to x_ndc = x / (1 + 2x) < 1/3 with |y_ndc| <= 1 - 2 x_ndc; the texel sampled at a covered
    pixel is the perspective-correct one, u = (x + 1) / 2 with x = x_ndc / (1 - 2 x_ndc) (closed form, float64)."""
    W, H = 240, 120
    ramp = np.zeros((1, 256, 4), dtype=np.uint8)
    ramp[0, :, 0] = np.arange(256); ramp[0, :, 3] = 255
    scene = scenes.space_quads_scene(1, W, H, seed=0, blend="additive", filter="nearest")
    m = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0], [2, 0, 0, 1]], dtype=np.float64)
    scene["uniform"] = np.concatenate([np.ones(4), m.T.reshape(-1)]).astype(f32)
    scene["instances"] = np.array([[0.0, 0.0, 1.0, 0.0]], dtype=f32)
    scene["texture"] = ramp
    scene["clear"] = (0., 0., 0., 0.)
    out = scenes.draw_oracle(scene)["color"]
    covered = out[..., 3] > 0
    ii, jj = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    x_ndc = (ii / W - 0.5) * 2.0; y_ndc = (0.5 - jj / H) * 2.0
    slack = 2.5 / H                                      # a pixel either way
    sure_in = (x_ndc < 1.0 / 3.0 - 2.0 / W) & (np.abs(y_ndc) < 1.0 - 2.0 * x_ndc - 2.0 * slack)
    sure_out = (x_ndc > 1.0 / 3.0 + 2.0 / W) | (np.abs(y_ndc) > 1.0 - 2.0 * x_ndc + 2.0 * slack)
    assert covered[sure_in].all() and not covered[sure_out].any() and sure_in.sum() > 15000
    x = x_ndc / (1.0 - 2.0 * x_ndc)
    u = (x + 1.0) / 2.0
    lo = np.clip(np.floor((u - 0.004) * 256.0), 0, 255); hi = np.clip(np.floor((u + 0.004) * 256.0), 0, 255)
    got = out[..., 0].astype(np.float64)
    assert ((got >= lo - 1) & (got <= hi + 1))[sure_in].all()


def test_depth_clipping_keeps_zero_to_w(oracle):
    """Quads in space under a matrix that maps z := x and leaves w = 1: only the part with 0 <= x_ndc <= 1 survives."""
    W, H = 128, 64
    scene = scenes.space_quads_scene(1, W, H, seed=0, blend="additive", filter="nearest")
    m = np.eye(4); m[2, :] = [1, 0, 0, 0]                       # z = x
    scene["uniform"] = np.concatenate([np.ones(4), m.T.reshape(-1)]).astype(f32)
    scene["instances"] = np.array([[-0.031, 0.0, 0.9, 0.0]], dtype=f32)     # x_ndc in [-0.931, 0.869]
    scene["clear"] = (0., 0., 0., 0.)
    covered = scenes.draw_oracle(scene)["color"][..., 3] > 0
    cols = np.nonzero(covered.any(axis=0))[0]
    assert cols.min() == W // 2 and cols.max() == int(np.floor((0.869 * 0.5 + 0.5) * W - 0.5))
    # the same quad pushed beyond z = w disappears, a NaN coordinate drops the triangle
    m2 = m.copy(); m2[2, 3] = 2.0
    scene["uniform"] = np.concatenate([np.ones(4), m2.T.reshape(-1)]).astype(f32)
    assert not (scenes.draw_oracle(scene)["color"][..., 3] > 0).any()


# ---- multisampling ----------------------------------------------------------------------------------------------------------
# Standard sample locations inside a pixel (Vulkan 1.3, "Multisampling"), x right, y down.
STANDARD_SAMPLES = {
    2: [(0.75, 0.75), (0.25, 0.25)],
    4: [(0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875)],
    8: [(0.5625, 0.3125), (0.4375, 0.6875), (0.8125, 0.5625), (0.3125, 0.1875),
        (0.1875, 0.8125), (0.0625, 0.4375), (0.6875, 0.9375), (0.9375, 0.0625)],
}


@pytest.mark.parametrize("samples", [2, 4, 8])
def test_multisample_coverage_of_an_axis_aligned_quad_by_hand(oracle, samples):
    """An opaque quad [x0, x1) x [y0, y1) with edges off the sample grid: every stored sample is set exactly when its
    standard location lies inside the rectangle, each covered sample holds exactly one layer (the shared diagonal is
    hit once per sample), and the resolve is the mean rounded to nearest, ties to even."""
    x0, y0, x1, y1 = 2.2, 1.6, 6.2, 5.6
    s = px_quad_scene(x0, y0, x1, y1, W=8, H=8, rgba=(255, 255, 255, 255), clear=(0., 0., 0., 0.))
    s["samples"] = samples
    o = scenes.draw_oracle(s, threads=1)
    expect = np.zeros((8, 8, samples), dtype=np.uint8)
    for j in range(8):
        for i in range(8):
            for k, (sx, sy) in enumerate(STANDARD_SAMPLES[samples]):
                expect[j, i, k] = 255 if (x0 <= i + sx < x1 and y0 <= j + sy < y1) else 0
    assert np.array_equal(o["msaa"][..., 0], expect)
    assert np.array_equal(o["msaa"][..., 3], expect)
    mean = expect.astype(np.float64).sum(axis=2) / samples
    assert np.array_equal(o["color"][..., 0], np.rint(mean).astype(np.uint8))        # np.rint: ties to even
    if samples == 4:
        assert o["color"][3, 2, 0] == 191 and o["color"][3, 6, 0] == 64 and o["color"][1, 3, 0] == 128 and o["color"][1, 2, 0] == 64


def test_multisample_edges_on_sample_positions_follow_the_top_left_rule(oracle):
    # left edge exactly through sample 2 of 4x (x = 0.125): a left edge takes it; the right edge at x = 4.125 does not
    s = px_quad_scene(2.125, 1.0, 4.125, 3.0, W=8, H=4, rgba=(255, 255, 255, 255))
    s["samples"] = 4
    o = scenes.draw_oracle(s, threads=1)
    assert o["msaa"][1, 2, 2, 0] == 255 and o["msaa"][1, 4, 2, 0] == 0
    assert o["color"][1, 2, 0] == 255 and o["color"][1, 4, 0] == 0 and o["color"][1, 3, 0] == 255


def test_multisample_area_is_preserved(oracle):
    """Size-independent property: an opaque quad of area A pixels adds 255*A to the resolved channel sum, up to the
    sampling error of its four edges and 0.5 per edge pixel of rounding."""
    for samples in (2, 4, 8):
        s = px_quad_scene(3.3, 2.7, 27.9, 25.1, rgba=(255, 255, 255, 255))
        s["samples"] = samples
        o = scenes.draw_oracle(s, threads=1)
        area = (27.9 - 3.3) * (25.1 - 2.7)
        total = o["color"][..., 0].astype(np.float64).sum() / 255.0
        assert abs(total - area) < 0.35 * (2 * (24.6 + 22.4)) , (samples, total, area)
        single = scenes.draw_oracle(px_quad_scene(3.3, 2.7, 27.9, 25.1, rgba=(255, 255, 255, 255)), threads=1)
        assert abs(total - area) <= abs(single["color"][..., 0].astype(np.float64).sum() / 255.0 - area) + 1e-9


def test_multisample_load_continues_from_the_sample_buffer(oracle):
    s = px_quad_scene(2.2, 1.6, 6.2, 5.6, W=8, H=8, blend="additive", rgba=(10, 20, 30, 40))
    s["samples"] = 4
    one = scenes.draw_oracle(s, threads=1)
    two = scenes.draw_oracle(dict(s, clear=None), threads=1, msaa=one["msaa"], initial=np.full((8, 8, 4), 99, np.uint8))
    assert np.array_equal(two["msaa"], one["msaa"] * 2)
    assert two["color"][3, 3, 0] == 20 and two["color"][0, 0, 0] == 0        # the target's old contents are gone


def test_multisample_thread_invariance(oracle):
    s = scenes.random_quads_scene(400, 9, 96, 64, seed=3)
    s["samples"] = 4
    a = scenes.draw_oracle(s, threads=1)
    b = scenes.draw_oracle(s, threads=5)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["msaa"], b["msaa"]) and a["fragments"] == b["fragments"]


# ---- render target formats -----------------------------------------------------------------------------------------------------
def test_float_to_half_matches_numpy(oracle):
    """The f32 -> f16 rounding RGBA16F targets use (nearest even, overflow to infinity, denormals kept) against numpy."""
    import ctypes as C
    from oracle import loader
    lib = loader.raster_lib()
    lib.orc_float_to_half.argtypes = [C.c_float]; lib.orc_float_to_half.restype = C.c_uint16
    rng = np.random.default_rng(0)
    bits = np.concatenate([rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32),
                           np.arange(0x33000000 - 5, 0x33000000 + 5, dtype=np.uint32),          # around 2^-25 (rounds to 0 / min denormal)
                           np.arange(0x477fefff - 3, 0x477ff003, dtype=np.uint32),              # around 65520 (rounds to max / infinity)
                           np.arange(0x38800000 - 4, 0x38800000 + 4, dtype=np.uint32)])         # around 2^-14 (denormal / normal)
    vals = np.concatenate([bits.view(np.float32), (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 12, 20000))).astype(np.float32)])
    with np.errstate(all="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        if not np.isnan(v):
            assert lib.orc_float_to_half(float(v)) == w, v


def test_float_and_r8_targets_by_hand(oracle):
    # three additive layers of (0.4, 0.8, 1.0, 0.5) (texture bytes 102, 204, 255, 128 / 255)
    s = px_quad_scene(2.0, 2.0, 6.0, 6.0, W=8, H=8, rgba=(102, 204, 255, 128), n=3, clear=(0.5, 0., 0., 0.))
    c = [f32(v) * f32(1.0 / 255.0) for v in (102, 204, 255, 128)]
    # RGBA32F: unclamped f32 accumulation on top of the clear colour
    o = scenes.draw_oracle(dict(s, target_format="rgba_f32"), threads=1)["color"].view(np.float32).reshape(8, 8, 4)
    want = [f32(0.5), f32(0), f32(0), f32(0)]
    for _ in range(3):
        want = [f32(w + ck) for w, ck in zip(want, c)]
    assert np.array_equal(o[3, 3], np.array(want, dtype=f32)) and o[3, 3, 2] == 3.0 and np.array_equal(o[0, 0], np.array([0.5, 0, 0, 0], dtype=f32))
    # RGBA16F: the same, rounded to f16 after every layer
    o = scenes.draw_oracle(dict(s, target_format="rgba_f16"), threads=1)["color"].view(np.float16).reshape(8, 8, 4)
    want = [np.float16(0.5), np.float16(0), np.float16(0), np.float16(0)]
    for _ in range(3):
        want = [np.float16(f32(w) + ck) for w, ck in zip(want, c)]
    assert np.array_equal(o[3, 3], np.array(want, dtype=np.float16))
    # R8: red only, saturating at 255: 128 + 3 * 102 > 255
    o = scenes.draw_oracle(dict(s, target_format="r_u8"), threads=1)["color"]
    assert o.shape == (8, 8, 1) and o[3, 3, 0] == 255 and o[0, 0, 0] == 128
    o = scenes.draw_oracle(dict(s, target_format="r_u8", instances=np.zeros((1, 2), dtype=f32)), threads=1)["color"]
    assert o[3, 3, 0] == 128 + 102
