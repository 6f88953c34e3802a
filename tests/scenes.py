"""Scene descriptions shared by the parity tests, smoke() and bench.py: each scene is a plain dict that can be drawn
both by the HIP renderer (through renderer_amd, i.e. the C ABI) and by the CPU oracle.

Scene keys
  shader        'hello' | 'sized' | 'tilted' (projective quads; needs 'uniform': float32 [6]) |
                'space' (quads in space; 'uniform': float32 [20] = tint + column-major 4x4 matrix) |
                'layered' (hello sampling layer uniform[0] of a 2D-array texture; needs 'uniform' and 'texture_layers')
  width,height  window / target size
  target        'screen' (BGRA8 offscreen screen) | 'texture' (RGBA8 texture target)
  position,uv   per-vertex attributes (location 0 and 1), float32 [V, 2]
  instances     float32 [N, 2] ('hello') or [N, 4] ('sized')
  texture       uint8 [H, W, 4] (or [H, W] for r_u8), texture_format, filter ('linear' | 'nearest')
  blend         'premultiplied' | 'additive'
  topology      'strip' | 'list'
  clear         None | (r, g, b, a)
  viewport      None | (aspect_x, aspect_y)
  record        False | True  (adds the RGBA8 recording attachment)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def load_letter_f():
    """The reference example's texture (examples/quads/letter_f.png, committed as tests/golden/letter_f.png),
    premultiplied the way examples/quads_main_thread.rs:181-189 does: c * (a / 255) in f32, rounded half away from
    zero (Rust f32::round)."""
    from PIL import Image
    img = np.array(Image.open(os.path.join(HERE, "golden", "letter_f.png")).convert("RGBA"), dtype=np.uint8)
    out = img.copy()
    alpha = img[..., 3].astype(f32) / f32(255.0)
    for c in range(3):
        v = img[..., c].astype(f32) * alpha
        out[..., c] = np.floor(v + f32(0.5)).astype(np.uint8)      # values are >= 0, so this is round-half-away
    return out


def example_positions(frame):
    """Instance offsets of the quads example at frame `frame` (1-based): the f32 bounce recurrence of
    examples/quads_main_thread.rs:122-141."""
    state = [[f32(0.3), f32(0.015)], [f32(-0.3), f32(0.01)], [f32(-0.5), f32(0.005)], [f32(-0.1), f32(0.02)]]
    for _ in range(frame):
        for s in state:
            s[0] = f32(s[0] + s[1])
            if s[0] > f32(0.9) or s[0] < f32(-0.9):
                s[1] = f32(s[1] * f32(-1.0))
    return np.array([state[0][0], state[1][0], state[2][0], state[3][0]], dtype=f32).reshape(2, 2)


QUAD_POSITION = np.array([[-0.1, -0.1], [-0.1, 0.1], [0.1, -0.1], [0.1, 0.1]], dtype=f32)   # quads_main_thread.rs:86
QUAD_UV = np.array([[0., 1.], [0., 0.], [1., 1.], [1., 0.]], dtype=f32)                      # quads_main_thread.rs:87
UNIT_POSITION = np.array([[-1., -1.], [-1., 1.], [1., -1.], [1., 1.]], dtype=f32)


def example_scene(frame, width=1920, height=1080, record=False):
    """BASELINE config[0]: the quads example at 1920x1080 with a 1:1 letterboxed viewport."""
    return dict(shader="hello", width=width, height=height, target="screen", position=QUAD_POSITION, uv=QUAD_UV,
                instances=example_positions(frame), texture=load_letter_f(), texture_format="rgba_u8",
                filter="linear", blend="premultiplied", topology="strip", clear=(0., 0., 0., 1.),
                viewport=(1., 1.), record=record)


def random_quads_scene(n, side_px, width, height, seed, blend="premultiplied", filter="linear", target="texture",
                       clear=(0., 0., 0., 1.), texture=None):
    """BASELINE config[1] family: n quads of side_px x side_px pixels, positions uniform with random sub-pixel
    offsets, fully inside the target."""
    rng = np.random.default_rng(seed)
    hx, hy = f32(side_px / width), f32(side_px / height)         # half extents in NDC: (side/2) / (size/2)
    position = np.array([[-hx, -hy], [-hx, hy], [hx, -hy], [hx, hy]], dtype=f32)
    cx = rng.uniform(-1.0 + float(hx), 1.0 - float(hx), n).astype(f32)
    cy = rng.uniform(-1.0 + float(hy), 1.0 - float(hy), n).astype(f32)
    return dict(shader="hello", width=width, height=height, target=target, position=position, uv=QUAD_UV,
                instances=np.stack([cx, cy], axis=1).astype(f32),
                texture=load_letter_f() if texture is None else texture, texture_format="rgba_u8",
                filter=filter, blend=blend, topology="strip", clear=clear, viewport=None, record=False)


def tiny_quads_scene(n, width, height, seed, lo=2, hi=8):
    """BASELINE config[2]: n quads with integer side in [lo, hi] px, additive blending, vec4 instance records."""
    rng = np.random.default_rng(seed)
    side = rng.integers(lo, hi + 1, n)
    hw = (side / width).astype(f32); hh = (side / height).astype(f32)
    cx = rng.uniform(-0.98, 0.98, n).astype(f32); cy = rng.uniform(-0.98, 0.98, n).astype(f32)
    return dict(shader="sized", width=width, height=height, target="texture", position=UNIT_POSITION, uv=QUAD_UV,
                instances=np.stack([cx, cy, hw, hh], axis=1).astype(f32), texture=load_letter_f(),
                texture_format="rgba_u8", filter="linear", blend="additive", topology="strip",
                clear=(0., 0., 0., 1.), viewport=None, record=False)


# target_format -> (oracle format id = reference src/format.rs order, bytes per texel)
TARGET_FORMATS = {"r_u8": (0, 1), "bgra_u8": (1, 4), "rgba_u8": (2, 4), "rgba_f16": (3, 8), "rgba_f32": (4, 16)}


# ---- drawing with the oracle -------------------------------------------------------------------------------------------------
def draw_oracle(scene, threads=0, clip_to_viewport=True, initial=None, msaa=None, initial_recording=None):
    """Returns {'color': uint8 [H, W, 4] in the target's byte order, 'recording': ... (if record), 'fragments': n}.
    scene['samples'] > 1: multisampled; `msaa` is the pipeline's multisampled texture [H, W, S, bytes per texel] (zeros when None),
    returned under 'msaa'."""
    from oracle import loader as O
    W, H = scene["width"], scene["height"]
    fmt = O.FMT_BGRA_U8 if scene["target"] == "screen" else TARGET_FORMATS[scene.get("target_format", "rgba_u8")][0]
    bpt = 4 if scene["target"] == "screen" else TARGET_FORMATS[scene.get("target_format", "rgba_u8")][1]
    color = np.zeros((H, W, bpt), dtype=np.uint8) if initial is None else initial.copy()
    targets = [dict(data=color, format=fmt, clear=scene["clear"])]
    rec = None
    if scene.get("record"):
        rec = np.zeros((H, W, 4), dtype=np.uint8) if initial_recording is None else initial_recording.copy()
        targets.append(dict(data=rec, format=O.FMT_RGBA_U8, clear=scene.get("record_clear", scene["clear"])))
    viewport = None
    if scene["viewport"] is not None:
        from renderer_amd.viewport import Viewport
        v = Viewport(scene["viewport"][0], scene["viewport"][1], float(W), float(H))
        viewport = (v.margin_x, v.margin_y, v.width, v.height)
    tex = scene["texture"]
    tfmt = {"rgba_u8": O.FMT_RGBA_U8, "bgra_u8": O.FMT_BGRA_U8, "r_u8": O.FMT_R_U8, "rgba_f16": O.FMT_RGBA_F16,
            "rgba_f32": O.FMT_RGBA_F32}[scene["texture_format"]]
    layers = scene.get("texture_layers", 1)
    th, tw = (tex.shape[0] // layers, tex.shape[1])
    blend = {"premultiplied": (O.BF_ONE, O.BF_ONE_MINUS_SRC_ALPHA), "additive": (O.BF_ONE, O.BF_ONE)}[scene["blend"]]
    samples = scene.get("samples", 1)
    if samples > 1:
        msaa = np.zeros((H, W, samples, bpt), dtype=np.uint8) if msaa is None else msaa.copy()
    n = O.render(shader={"hello": O.SHADER_HELLO, "sized": O.SHADER_SIZED, "tilted": O.SHADER_TILTED, "space": O.SHADER_SPACE,
                         "layered": O.SHADER_HELLO_LAYER}[scene["shader"]],
                 uniforms=[scene["uniform"]] if "uniform" in scene else (),
                 topology=O.TOPO_TRIANGLE_STRIP if scene["topology"] == "strip" else O.TOPO_TRIANGLE_LIST,
                 vertex_count=len(scene["position"]), instance_count=len(scene["instances"]),
                 attrs=[(scene["position"], 2), (scene["uv"], 2)], storage=[scene["instances"]],
                 textures=[(tex, tw, th, layers, tfmt, O.FILTER_LINEAR if scene["filter"] == "linear" else O.FILTER_NEAREST)],
                 targets=targets, blend=blend, viewport=viewport, clip_to_viewport=clip_to_viewport, threads=threads,
                 samples=samples, msaa=msaa if samples > 1 else None)
    out = {"color": color, "fragments": n}
    if samples > 1:
        out["msaa"] = msaa
    if rec is not None:
        out["recording"] = rec
    return out


# ---- drawing with the HIP renderer ----------------------------------------------------------------------------------------------
def shader_code(name):
    import renderer_amd
    d = os.path.join(os.path.dirname(renderer_amd.__file__), "shaders")
    with open(os.path.join(d, f"{name}.vert.co"), "rb") as f:
        vert = f.read()
    frag_path = os.environ.get("RH_FRAG_CO_OVERRIDE") or os.path.join(d, f"{name}.frag.co")     # kernel experiments
    with open(frag_path, "rb") as f:
        frag = f.read()
    return vert, frag


class HipScene:
    """Builds the renderer objects for a scene the way examples/quads_main_thread.rs:37-88 does."""

    def __init__(self, scene, renderer=None, clip_to_viewport=True):
        from renderer_amd import Renderer
        self.scene = scene
        W, H = scene["width"], scene["height"]
        self.renderer = r = renderer or Renderer(W, H)
        self.owns = renderer is None
        r.set_clip_to_viewport(clip_to_viewport)
        vert, frag = shader_code(scene["shader"])
        tex = scene["texture"]
        layers = scene.get("texture_layers", 1)
        th, tw = tex.shape[0] // layers, tex.shape[1]
        fmt = {"rgba_u8": Renderer.rgba_u8(), "bgra_u8": Renderer.bgra_u8(), "r_u8": Renderer.red_u8(),
               "rgba_f16": Renderer.rgba_f16(), "rgba_f32": Renderer.rgba_f32()}[scene["texture_format"]]
        flt = Renderer.linear_filtering() if scene["filter"] == "linear" else Renderer.nearest_filtering()
        a_position = r.attribute(0, 2)
        a_tex_coord = r.attribute(1, 2)
        i_offset = r.instanced()
        t_texture = r.texture(tw, th, layers, flt, fmt, False, False, True)
        uniforms = []
        if "uniform" in scene:
            self.u_params = r.uniform()
            uniforms = [(self.u_params, Renderer.visible_to_both_shaders())]
        program = r.program(vert, frag, [a_position, a_tex_coord], [i_offset], uniforms,
                            [(t_texture, Renderer.visible_to_fragment_shader())])
        texture_slot = (0, 1 + len(uniforms))
        blend = Renderer.pre_multiplied_blend() if scene["blend"] == "premultiplied" else Renderer.additive_blend()
        primitive = Renderer.triangle_strip_primitive() if scene["topology"] == "strip" else Renderer.triangle_primitive()
        if scene["target"] == "screen":
            self.target_texture = None
            target = Renderer.screen_target()
        else:
            tfmt = {"rgba_u8": Renderer.rgba_u8(), "bgra_u8": Renderer.bgra_u8(), "r_u8": Renderer.red_u8(),
                    "rgba_f16": Renderer.rgba_f16(), "rgba_f32": Renderer.rgba_f32()}[scene.get("target_format", "rgba_u8")]
            self.target_texture = r.texture(W, H, 1, Renderer.nearest_filtering(), tfmt, True, True, False)
            target = Renderer.texture_target(self.target_texture)
        self.pipeline = r.pipeline(program, blend, primitive, scene.get("samples", 1), [target])
        r.set_attribute(self.pipeline, 0, scene["position"])
        r.set_attribute(self.pipeline, 1, scene["uv"])
        layer_px = tex.reshape(layers, -1)
        r.set_texture(self.pipeline, texture_slot, [layer_px[i] for i in range(layers)])
        if uniforms:
            r.set_uniform(self.pipeline, (0, 1), scene["uniform"])
        r.set_instanced(self.pipeline, (0, 0), scene["instances"])
        self.clear = Renderer.clear_color(*scene["clear"]) if scene["clear"] is not None else None
        self.viewport = r.viewport(*scene["viewport"]) if scene["viewport"] is not None else None
        self.count = (len(scene["instances"]), len(scene["position"]))
        self.frames = []
        if scene.get("record"):
            rc = scene.get("record_clear", scene["clear"])
            r.start_recording([self.pipeline], Renderer.clear_color(*rc) if rc is not None else None, 500.,
                              lambda frame: self.frames.append(frame))

    def set_initial(self, image):
        """Pre-fills the colour target (for LoadOp::Load scenes)."""
        from renderer_amd import _ffi
        h = self.target_texture.handle if self.target_texture is not None else self.renderer.screen_handle()
        image = np.ascontiguousarray(image)
        _ffi.check(_ffi.lib().rh_texture_write(h, 0, 0, 0, image.shape[1], image.shape[0], image.ctypes.data,
                                               image.nbytes, image.shape[1] * image.shape[2]))

    def draw(self):
        self.renderer.render(self.pipeline, self.clear, self.viewport, self.count)
        self.renderer.finish_frame()

    def read_color(self):
        if self.target_texture is not None:
            return self.target_texture.read()
        return self.renderer.read_screen()

    def read_recording(self):
        return self.renderer.recorder.recording_texture.read()

    def close(self):
        if self.owns:
            self.renderer.close()


def draw_hip(scene, clip_to_viewport=True, initial=None, twice=False):
    """`twice` (scenes that clear their targets): the frame is drawn, thrown away and drawn again, so that the checked
    frame is one the device queued before its pair count was known (the first pass of a device waits for it)."""
    hs = HipScene(scene, clip_to_viewport=clip_to_viewport)
    try:
        if initial is not None:
            hs.set_initial(initial)
        elif twice:
            hs.draw()
        hs.draw()
        out = {"color": hs.read_color(), "stats": hs.renderer.last_pass_stats()}
        if scene.get("samples", 1) > 1:
            out["msaa"] = hs.pipeline.msaa_texture.read()
        if scene.get("record"):
            out["recording"] = hs.read_recording()
        return out
    finally:
        hs.close()


def tilted_quads_scene(n, width, height, seed, blend="premultiplied", filter="linear"):
    """Projective quads: per-vertex w (perspective-correct interpolation), three varyings, a uniform read by both
    stages."""
    rng = np.random.default_rng(seed)
    cx = rng.uniform(-0.9, 0.9, n).astype(f32); cy = rng.uniform(-0.9, 0.9, n).astype(f32)
    half = rng.uniform(0.02, 0.25, n).astype(f32)
    tilt = rng.uniform(-0.8, 0.8, n).astype(f32)
    tilt[::7] = 0.0                                   # some quads stay affine (w = 1 at every vertex)
    scene = dict(shader="tilted", width=width, height=height, target="texture", position=UNIT_POSITION, uv=QUAD_UV,
                 instances=np.stack([cx, cy, half, tilt], axis=1).astype(f32), texture=load_letter_f(),
                 texture_format="rgba_u8", filter=filter, blend=blend, topology="strip", clear=(0.1, 0.1, 0.1, 1.0),
                 viewport=None, record=False)
    scene["uniform"] = np.array([1.0, 0.8, 0.6, 0.9, 1.0, 0.75], dtype=f32)
    return scene


def perspective_matrix(fov_y_deg, aspect, near, far):
    """Column-major projection with the 0..1 depth range (the camera looks down -z), as float32 [16]."""
    f = 1.0 / np.tan(np.radians(fov_y_deg) / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = f / aspect; m[1, 1] = f
    m[2, 2] = far / (near - far); m[2, 3] = near * far / (near - far)
    m[3, 2] = -1.0
    return m


def space_quads_scene(n, width, height, seed, blend="premultiplied", filter="linear", near=0.5, far=6.0, tilt_deg=70.0,
                      depth_range=(-7.0, 1.0), half_range=(0.05, 1.5), topology="strip", spread=3.0):
    """Quads in space: a ground-plane-like set of quads tilted away from the camera, many of them crossing the near and
    far planes, some reaching behind the eye (w <= 0) and some far outside the guard band."""
    rng = np.random.default_rng(seed)
    a = np.radians(tilt_deg)
    rot = np.array([[1, 0, 0, 0], [0, np.cos(a), -np.sin(a), 0], [0, np.sin(a), np.cos(a), 0], [0, 0, 0, 1]], dtype=np.float64)
    trans = np.eye(4); trans[2, 3] = -2.0; trans[1, 3] = -0.3
    m = perspective_matrix(60.0, width / height, near, far) @ trans @ rot
    cx = rng.uniform(-spread, spread, n).astype(f32); cy = rng.uniform(-spread, spread, n).astype(f32)
    half = rng.uniform(*half_range, n).astype(f32)
    depth = rng.uniform(*depth_range, n).astype(f32)
    if topology == "strip":
        position, uv = UNIT_POSITION, QUAD_UV
    else:
        position = np.array([[-1., -1.], [-1., 1.], [1., -1.], [1., -1.], [-1., 1.], [1., 1.]], dtype=f32)
        uv = np.array([[0., 1.], [0., 0.], [1., 1.], [1., 1.], [0., 0.], [1., 0.]], dtype=f32)
    scene = dict(shader="space", width=width, height=height, target="texture", position=position, uv=uv,
                 instances=np.stack([cx, cy, half, depth], axis=1).astype(f32), texture=load_letter_f(),
                 texture_format="rgba_u8", filter=filter, blend=blend, topology=topology, clear=(0.1, 0.1, 0.1, 1.0),
                 viewport=None, record=False)
    scene["uniform"] = np.concatenate([np.array([1.0, 0.9, 0.8, 0.9]), m.T.reshape(-1)]).astype(f32)
    return scene
