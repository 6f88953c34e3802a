"""rh_submit_static, the entry point that allocates nothing (its passes work in a static device array), against
rh_submit: same command buffers, bit-identical targets, two calls in a row on one device, the second with a smaller
input; and the limits are enforced with an error."""
import numpy as np
import pytest

import scenes
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def draw_both(r, scene):
    hs = scenes.HipScene(scene, renderer=r)
    out = {}
    for name in ("submit", "static"):
        hs.renderer.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        if name == "submit":
            hs.renderer.finish_frame()
        else:
            hs.renderer.flush_static()
        out[name] = {"color": hs.read_color()}
        if scene.get("record"):
            out[name]["recording"] = hs.read_recording()
        # scribble over the target so that the second way has to draw every pixel again
        hs.set_initial(np.full_like(out[name]["color"], 0x5a))
    return out


def test_static_submit_equals_submit_twice_in_a_row(oracle):
    from renderer_amd import Renderer
    W, H = 640, 360
    r = Renderer(W, H)
    try:
        big = scenes.random_quads_scene(30_000, 14, W, H, seed=71)
        got = draw_both(r, big)
        assert_same(got["static"]["color"], got["submit"]["color"], "first call")
        assert_same(got["static"]["color"], scenes.draw_oracle(big)["color"], "first call against the oracle")
        # second call: a smaller input (fewer pairs, fewer records: the array keeps the first call's contents)
        small = scenes.tiny_quads_scene(900, W, H, seed=72)
        small["blend"] = "premultiplied"
        got = draw_both(r, small)
        assert_same(got["static"]["color"], got["submit"]["color"], "second, smaller call")
        assert_same(got["static"]["color"], scenes.draw_oracle(small)["color"], "second call against the oracle")
        # a recorded frame (two attachments) and the example's letterboxed viewport
        ex = scenes.example_scene(37, W, H, record=True)
        r2 = Renderer(W, H)
        try:
            got = draw_both(r2, ex)
            assert_same(got["static"]["color"], got["submit"]["color"], "example frame")
            assert_same(got["static"]["recording"], got["submit"]["recording"], "recording attachment")
        finally:
            r2.close()
    finally:
        r.close()


def test_static_submit_rejects_a_pass_beyond_its_limits(oracle):
    from renderer_amd import Renderer, RendererError
    W, H = 640, 360
    r = Renderer(W, H)
    try:
        scene = scenes.random_quads_scene(70_000, 4, W, H, seed=73)          # more than 65536 triangle pairs
        hs = scenes.HipScene(scene, renderer=r)
        r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        with pytest.raises(RendererError, match="allocates nothing"):
            r.flush_static()
        # the device goes on working
        hs.draw()
        assert_same(hs.read_color(), scenes.draw_oracle(scene)["color"], "after the rejected call")
    finally:
        r.close()


def test_static_submit_rejects_a_pass_that_needs_clipping(oracle):
    """The kernels that clip use private memory, which the runtime allocates: rh_submit_static says so instead of running
    them, whether it meets the crossing triangle itself or the pipeline has met one before."""
    from renderer_amd import Renderer, RendererError
    W, H = 320, 200
    r = Renderer(W, H)
    try:
        scene = scenes.space_quads_scene(50, W, H, seed=74)
        hs = scenes.HipScene(scene, renderer=r)
        r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        with pytest.raises(RendererError, match="clip plane"):
            r.flush_static()
        hs.draw()                                   # rh_submit clips
        assert_same(hs.read_color(), scenes.draw_oracle(scene)["color"], "after the rejected call")
        r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
        with pytest.raises(RendererError, match="clip plane"):
            r.flush_static()
    finally:
        r.close()


def test_static_submit_behind_a_pass_that_has_to_run_again(oracle):
    """rh_submit leaves its last pass queued without waiting for its pair count; when that pass did not fit the room it
    was given it has to run again, which allocates.  rh_submit_static must not do that behind the caller's back: it waits
    for the count and reports an error; after rh_device_synchronize (which runs the pass again) it works."""
    from renderer_amd import Renderer, RendererError, _ffi
    W, H = 640, 360
    r = Renderer(W, H)
    try:
        small = scenes.random_quads_scene(300, 14, W, H, seed=75)
        hs_small = scenes.HipScene(small, renderer=r)
        hs_small.draw()                                   # the device's first pass: sizes the room for the next ones
        big = scenes.random_quads_scene(30_000, 14, W, H, seed=76)
        hs = scenes.HipScene(big, renderer=r)
        # (rh_submit does not look at the count before it returns: the pass stays queued with too little room)
        before = r.last_pass_stats()["reruns"]
        _ffi.check(_ffi.lib().rh_device_set_option(r.device, _ffi.OPTION_EAGER_COUNT, 0))
        hs.draw()
        r.render(hs_small.pipeline, hs_small.clear, hs_small.viewport, hs_small.count)
        with pytest.raises(RendererError, match="has to run again"):
            r.flush_static()
        r.synchronize()                                   # runs the big pass again, with room
        assert r.last_pass_stats()["reruns"] == before + 1
        assert_same(hs.read_color(), scenes.draw_oracle(big)["color"], "the pass that ran again")
        r.render(hs_small.pipeline, hs_small.clear, hs_small.viewport, hs_small.count)
        r.flush_static()
        assert_same(hs_small.read_color(), scenes.draw_oracle(small)["color"], "static submit afterwards")
    finally:
        r.close()


def test_static_submits_of_two_devices_on_one_gpu_do_not_overlap(oracle):
    """Two rh_devices on one GPU share that GPU's static array, each with a stream of its own; a call returns with its
    kernels queued.  The second device's call waits (on the GPU) for the first one's kernels."""
    from renderer_amd import Renderer, _ffi
    import ctypes as C
    W, H = 640, 360
    a, b = Renderer(W, H), Renderer(W, H)
    try:
        sa = scenes.random_quads_scene(30_000, 14, W, H, seed=77)
        sb = scenes.random_quads_scene(20_000, 10, W, H, seed=78)
        ha, hb = scenes.HipScene(sa, renderer=a), scenes.HipScene(sb, renderer=b)
        ha.draw(); hb.draw()                             # footprint copies made, rooms sized
        a.synchronize(); b.synchronize()
        for _ in range(3):
            for r, hs in ((a, ha), (b, hb)):
                hs.set_initial(np.full((H, W, 4), 0x33, np.uint8))
                r.render(hs.pipeline, hs.clear, hs.viewport, hs.count)
            # both calls return at once: no host synchronisation between them
            cmds = []
            for r in (a, b):
                n = len(r.commands)
                arr = (C.c_void_p * n)(*[c.value for c in r.commands])
                cmds.append((r, list(r.commands)))
                r.commands = []; r.flushes += 1
                _ffi.check(_ffi.lib().rh_submit_static(r.device, arr, n))
            a.synchronize(); b.synchronize()
            for r, cs in cmds:
                for c in cs:
                    _ffi.lib().rh_cmdbuf_release(c)
            assert_same(ha.read_color(), scenes.draw_oracle(sa)["color"], "first device")
            assert_same(hb.read_color(), scenes.draw_oracle(sb)["color"], "second device")
    finally:
        a.close(); b.close()
