"""GPU parity at the binning depths the benchmark configurations reach.  The per-tile lists come out of a stable sort
of (tile, quad) pairs by tile number; its work grows with the number of tiles (more key bits) and with the number of
pairs, and the order it leaves inside a tile is the blend order.  Small targets exercise one sorting pass only, so
these scenes use targets of more than 256 and more than 65536 tiles under enough pairs that every pass moves many of
them, drawn both in the form that waits for the pair count and queued before the count is known (the second draw of a
device).  Bar: bit-exact against the CPU oracle."""
import numpy as np
import pytest

import scenes
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def draw_both_forms(scene, what, tickets=False):
    want = scenes.draw_oracle(scene)["color"]
    hs = scenes.HipScene(scene)
    try:
        hs.renderer.set_binning_tickets(tickets)
        hs.draw()                                   # first pass of the device: waits for its pair count
        assert_same(hs.read_color(), want, what + " (waiting form)")
        hs.draw()                                   # queued whole before the count is known
        got = hs.read_color()
        stats = hs.renderer.last_pass_stats()
    finally:
        hs.close()
    assert_same(got, want, what + " (queued before the count)")
    return stats


def test_100k_alpha_quads_on_1024_tiles(oracle):
    """2048x1024 target (64 x 32 tiles of 32x32): tile numbers need 11 bits; ~12 layers of pre-multiplied alpha, so
    every pixel depends on the order of its quads."""
    scene = scenes.random_quads_scene(100_000, 16, 2048, 1024, seed=101)
    stats = draw_both_forms(scene, "100k quads / 1024 tiles")
    assert stats["tiles_x"] * stats["tiles_y"] > 256 and stats["bin_entries"] > 100_000


@pytest.mark.parametrize("blend", ["additive", "premultiplied"])
def test_300k_tiny_quads_at_4k(oracle, blend):
    """BASELINE configs[2] shape at a size the oracle draws in a second: 2..8 px quads at 3840x2160 (8160 tiles, 13 bits);
    additive as the configuration has it, and pre-multiplied so that the order inside a tile shows."""
    scene = scenes.tiny_quads_scene(300_000, 3840, 2160, seed=102)
    scene["blend"] = blend
    stats = draw_both_forms(scene, f"300k tiny quads at 4K / {blend}")
    assert stats["tiles_x"] * stats["tiles_y"] == 120 * 68


def test_headline_shape_200k_quads_at_4k(oracle):
    """BASELINE configs[1] shape with a fifth of the quads: 16 px quads, pre-multiplied alpha, linear filter, 4K."""
    scene = scenes.random_quads_scene(200_000, 16, 3840, 2160, seed=103)
    stats = draw_both_forms(scene, "200k quads at 4K")
    assert stats["bin_entries"] > 200_000


def test_more_than_65536_tiles(oracle):
    """8192 x 8224 target: 256 x 257 = 65792 tiles, tile numbers need 17 bits (BASELINE configs[3]: 16384^2, 18 bits)."""
    scene = scenes.random_quads_scene(6000, 48, 8192, 8224, seed=104)
    # a few columns of overlapping quads so that some tiles hold deep, order-sensitive lists
    rng = np.random.default_rng(104)
    stack = scene["instances"][:1500]
    stack[:, 0] = (rng.integers(0, 4, 1500) * 0.5 - 0.75 + rng.uniform(-0.004, 0.004, 1500)).astype(np.float32)
    stats = draw_both_forms(scene, "65792 tiles")
    assert stats["tiles_x"] * stats["tiles_y"] > 65536


def test_quads_over_many_tiles_in_the_queued_form(oracle):
    """Quads of 300 px touch about a hundred tiles each: their pairs are written by a whole wave, not by the thread that
    owns the quad; small quads in between keep the two ways of writing side by side."""
    scene = scenes.random_quads_scene(1500, 300, 2048, 1024, seed=105)
    small = scenes.random_quads_scene(1500, 10, 2048, 1024, seed=106)
    scene["instances"][::2] = small["instances"][::2]
    stats = draw_both_forms(scene, "quads over ~100 tiles")
    assert stats["bin_entries"] > 50_000


@pytest.mark.parametrize("shape", ["1024 tiles", "65792 tiles"])
def test_chain_places_from_tickets(oracle, shape):
    """The queued form with places in the look-back chains taken from ticket counters (what launches too large to fit the
    device at once do), at two and three sorting passes."""
    if shape == "1024 tiles":
        scene = scenes.random_quads_scene(100_000, 16, 2048, 1024, seed=107)
    else:
        scene = scenes.random_quads_scene(20_000, 48, 8192, 8224, seed=108)
    draw_both_forms(scene, "tickets / " + shape, tickets=True)


def tiny_then(r, W, H):
    """Leaves the device with the smallest room it ever gives the pair arrays of a queued pass."""
    tiny = scenes.random_quads_scene(40, 12, W, H, seed=1)
    hs = scenes.HipScene(tiny, renderer=r)
    hs.draw(); hs.draw()
    assert_same(hs.read_color(), scenes.draw_oracle(tiny)["color"], "tiny scene")
    return hs


def test_pass_that_did_not_fit_runs_again_load_additive_recording(oracle):
    """A dense pass queued with the room of a tiny predecessor: its pairs do not fit, its kernels must write NOTHING
    (the colour target is loaded, not cleared, and blended additively: one stray write would show), and the pass runs
    again; the recording attachment rides along."""
    from renderer_amd import Renderer
    W, H = 640, 360
    r = Renderer(W, H)
    try:
        tiny_then(r, W, H)
        before = r.last_pass_stats()["reruns"]
        scene = scenes.random_quads_scene(40_000, 12, W, H, seed=2, blend="additive", clear=None)
        scene["record"] = True
        scene["record_clear"] = (0.0, 0.0, 0.0, 1.0)
        init = np.random.default_rng(3).integers(0, 120, (H, W, 4), dtype=np.uint8)
        want = scenes.draw_oracle(scene, initial=init)
        hs = scenes.HipScene(scene, renderer=r)
        hs.set_initial(init)
        r.set_timing(True); r.drain_timings()
        hs.draw()
        timings = r.drain_timings(); r.set_timing(False)
        stats = r.last_pass_stats()
        assert stats["reruns"] == before + 1 and stats["bin_entries"] > 40_000
        assert timings["passes"] == 1               # the attempt that wrote nothing is not counted as a pass
        assert_same(hs.read_color(), want["color"], "loaded additive target after the re-run")
        assert_same(hs.read_recording(), want["recording"], "recording attachment after the re-run")
    finally:
        r.close()


def test_pass_that_did_not_fit_runs_again_multisampled(oracle):
    from renderer_amd import Renderer
    W, H = 333, 217
    r = Renderer(W, H)
    try:
        tiny_then(r, W, H)
        before = r.last_pass_stats()["reruns"]
        scene = scenes.random_quads_scene(30_000, 12, W, H, seed=5, blend="additive", clear=None)
        scene["samples"] = 4
        junk = np.full((H, W, 4), 77, dtype=np.uint8)
        want = scenes.draw_oracle(scene, initial=junk)          # Load continues from the (zeroed) multisampled texture
        hs = scenes.HipScene(scene, renderer=r)
        hs.set_initial(junk)
        hs.draw()
        assert r.last_pass_stats()["reruns"] == before + 1
        assert_same(hs.pipeline.msaa_texture.read().reshape(H, W * 4, 4), want["msaa"].reshape(H, W * 4, 4), "samples after the re-run")
        assert_same(hs.read_color(), want["color"], "resolve after the re-run")
    finally:
        r.close()


def test_unset_and_oversized_instance_buffers_read_zeros(oracle):
    """wgpu zero-initialises buffers: instances beyond the uploaded data (the headroom a grown Buffer keeps,
    reference src/buffer.rs:57-67) and a buffer that was never set read as zeros, also when the allocation is a
    recycled one."""
    from renderer_amd import Renderer
    W, H = 320, 200
    r = Renderer(W, H)
    try:
        # fill the allocation cache with junk-holding blocks of the sizes the instanced buffer will ask for
        junk_scene = scenes.random_quads_scene(100, 12, W, H, seed=9)
        for _ in range(2):
            hs = scenes.HipScene(junk_scene, renderer=r)
            hs.draw()
            r.synchronize()
        scene = scenes.random_quads_scene(100, 12, W, H, seed=10)
        hs = scenes.HipScene(scene, renderer=r)
        hs.count = (120, 4)                         # 20 instances past the uploaded 100: still inside the 2048-byte buffer
        hs.draw()
        padded = dict(scene, instances=np.concatenate([scene["instances"], np.zeros((20, 2), dtype=np.float32)]))
        assert_same(hs.read_color(), scenes.draw_oracle(padded)["color"], "instances past the data read zeros")
        # 300 instances: 44 of them past the END of the 2048-byte buffer (256 records): zeros as well, nothing outside is read
        hs.count = (300, 4)
        hs.draw()
        padded = dict(scene, instances=np.concatenate([scene["instances"], np.zeros((200, 2), dtype=np.float32)]))
        assert_same(hs.read_color(), scenes.draw_oracle(padded)["color"], "instances past the buffer read zeros")
    finally:
        r.close()


def test_first_clipped_pass_of_a_pipeline_queued_without_its_count(oracle):
    """A pipeline's first pass with a triangle across a clip plane, queued whole (the device has seen passes before): the
    usual binning kernel reports it through the pair count, every kernel behind it writes NOTHING (Load + additive: a
    stray write would show), the pass runs again with the clipping kernels, and the pipeline keeps those afterwards."""
    from renderer_amd import Renderer
    W, H = 640, 400
    r = Renderer(W, H)
    try:
        tiny_then(r, W, H)
        before = r.last_pass_stats()["reruns"]
        scene = scenes.space_quads_scene(300, W, H, seed=45, blend="additive")
        scene["clear"] = None
        scene["record"] = True
        scene["record_clear"] = (0.0, 0.0, 0.0, 1.0)
        init = np.random.default_rng(6).integers(0, 100, (H, W, 4), dtype=np.uint8)
        want = scenes.draw_oracle(scene, initial=init)
        hs = scenes.HipScene(scene, renderer=r)
        hs.set_initial(init)
        hs.draw()
        assert r.last_pass_stats()["reruns"] == before + 1
        assert_same(hs.read_color(), want["color"], "clipped scene after the re-run")
        assert_same(hs.read_recording(), want["recording"], "recording after the re-run")
        # the next frame of the pipeline is queued whole with the clipping kernels: no further re-run
        hs.set_initial(init)
        hs.draw()
        assert r.last_pass_stats()["reruns"] == before + 1
        assert_same(hs.read_color(), want["color"], "clipped scene, second frame")
    finally:
        r.close()


def test_pipeline_goes_back_to_the_usual_kernels_when_nothing_is_clipped(oracle):
    """A pipeline that met a crossing triangle runs the clipping kernels; after a run of passes that clipped nothing it
    returns to the usual ones (seen from outside: the next crossing triangle costs a re-run again), with the same
    pixels all along."""
    from renderer_amd import Renderer
    W, H = 320, 200
    r = Renderer(W, H)
    try:
        tiny_then(r, W, H)
        scene = scenes.space_quads_scene(80, W, H, seed=47)
        want_clipped = scenes.draw_oracle(scene)["color"]
        flat = dict(scene)
        m = np.eye(4); m[2, :] = [0, 0, 0, 0.5]                     # z = 0.5, w = 1: nothing crosses a plane
        flat["uniform"] = np.concatenate([np.ones(4), m.T.reshape(-1)]).astype(np.float32)
        want_flat = scenes.draw_oracle(flat)["color"]
        hs = scenes.HipScene(scene, renderer=r)
        before = r.last_pass_stats()["reruns"]
        hs.draw()
        assert r.last_pass_stats()["reruns"] == before + 1
        assert_same(hs.read_color(), want_clipped, "clipped")
        r.set_uniform(hs.pipeline, (0, 1), flat["uniform"])
        for _ in range(20):
            hs.draw()
        assert r.last_pass_stats()["reruns"] == before + 1
        assert_same(hs.read_color(), want_flat, "flat, after twenty passes")
        r.set_uniform(hs.pipeline, (0, 1), scene["uniform"])
        hs.draw()
        assert r.last_pass_stats()["reruns"] == before + 2             # the usual kernels met the crossing triangle again
        assert_same(hs.read_color(), want_clipped, "clipped again")
    finally:
        r.close()


def test_headroom_of_a_buffer_created_between_two_passes_reads_zeros(oracle):
    """A buffer created between two passes takes a recycled allocation and is cleared by work queued on the main stream,
    behind the tile kernel of the pass before; the next pass's binning, which may run beside that kernel on a stream of
    its own, must not read the buffer before it has been cleared.  Here the instance buffer grows into a block that held
    other floats, and the draw covers more instances than were uploaded: the rest must read as zeros (quads at the
    centre), as wgpu's zero-initialised buffers give."""
    import ctypes as C
    from renderer_amd import Renderer, _ffi
    from renderer_amd.buffer import HEADROOM, next_power_of_two
    W, H = 1920, 1080
    heavy = scenes.random_quads_scene(1000, 96, W, H, seed=131)        # a tile kernel long enough to run beside
    r = Renderer(W, H)
    try:
        hs = scenes.HipScene(heavy, renderer=r)
        hs.draw(); hs.draw()                                           # from now on passes are queued whole
        uploaded, drawn = 3000, 4000
        size = next_power_of_two(uploaded * 8 + HEADROOM)
        assert drawn * 8 <= size and size > next_power_of_two(1000 * 8 + HEADROOM)
        # a block of exactly that size, full of offsets that would put quads all over the target, goes back to the cache
        junk = np.random.default_rng(132).uniform(-0.9, 0.9, size // 4).astype(np.float32)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().rh_buffer_create(r.device, size, 2, junk.ctypes.data, junk.nbytes, C.byref(h)))
        hs.draw()
        _ffi.lib().rh_buffer_release(h)
        hs.draw()                                                      # its tile kernel is at work ...
        inst = scenes.random_quads_scene(uploaded, 96, W, H, seed=133)["instances"]
        r.set_instanced(hs.pipeline, (0, 0), inst)                     # ... while the buffer grows into the junk block
        r.render(hs.pipeline, hs.clear, hs.viewport, (drawn, 4))
        r.finish_frame()
        got = hs.read_color()
        scene = dict(heavy, instances=np.concatenate([inst, np.zeros((drawn - uploaded, 2), np.float32)]))
        assert_same(got, scenes.draw_oracle(scene)["color"], "instances beyond the uploaded ones")
    finally:
        r.close()


def test_animated_instances_keep_early_binning_and_match_the_oracle(oracle):
    """Instances re-uploaded every frame (what an animated scene does, reference examples/quads_main_thread.rs:144-151):
    the upload goes to a block of its own on the copy stream, so the pass that needs it still bins beside the tile
    kernel of the pass before, which keeps reading the old block.  Every frame matches the oracle's frame of the same
    instances; also a partial write (the rest of the buffer comes over from the old block) and a uniform rewritten every
    frame (read by both stages)."""
    from renderer_amd import Renderer, _ffi
    W, H = 1280, 720
    rng = np.random.default_rng(141)
    scene = scenes.random_quads_scene(60_000, 20, W, H, seed=142)
    r = Renderer(W, H)
    try:
        hs = scenes.HipScene(scene, renderer=r)
        hs.draw(); hs.draw()
        early0 = r.last_pass_stats()["early"]
        inst = scene["instances"].copy()
        frames = []
        for f in range(6):
            if f == 3:
                # a partial write: only the second half of the instances moves (rh_buffer_write at an offset)
                half = len(inst) // 2
                inst[half:] = inst[half:] + rng.uniform(-0.02, 0.02, inst[half:].shape).astype(np.float32)
                buf = hs.pipeline.program.instances[0].buffer
                part = np.ascontiguousarray(inst[half:])
                _ffi.check(_ffi.lib().rh_buffer_write(buf.handle, half * 8, part.ctypes.data, part.nbytes))
            else:
                inst = (inst + rng.uniform(-0.02, 0.02, inst.shape)).astype(np.float32)
                r.set_instanced(hs.pipeline, (0, 0), inst)
            r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
            r.finish_frame()                          # (no read-back in between: the frames queue up behind one another)
            frames.append(inst.copy())
        got_last = hs.read_color()
        assert r.last_pass_stats()["early"] - early0 >= 5, "the passes of the animated frames should bin early"
        assert_same(got_last, scenes.draw_oracle(dict(scene, instances=frames[-1]))["color"], "last animated frame")
        # every frame, one at a time
        for f, want_inst in enumerate(frames[:4]):
            r.set_instanced(hs.pipeline, (0, 0), want_inst)
            hs.draw()
            assert_same(hs.read_color(), scenes.draw_oracle(dict(scene, instances=want_inst))["color"], f"frame {f}")
    finally:
        r.close()
    # a uniform read by both stages, rewritten every frame
    tilted = scenes.tilted_quads_scene(4000, 640, 360, seed=143)
    r = Renderer(640, 360)
    try:
        hs = scenes.HipScene(tilted, renderer=r)
        hs.draw(); hs.draw()
        for f in range(4):
            u = tilted["uniform"].copy(); u[0] = 0.4 + 0.15 * f; u[4] = 1.0 - 0.05 * f
            r.set_uniform(hs.pipeline, (0, 1), u)
            r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
            r.finish_frame()
        assert_same(hs.read_color(), scenes.draw_oracle(dict(tilted, uniform=u))["color"], "uniform of the last frame")
    finally:
        r.close()
