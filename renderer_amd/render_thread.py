"""Renderer proxy that owns a Renderer on a worker thread (reference src/render_thread.rs:4-333): calls are queued as
messages and executed in order on that thread; objects are handed out as index handles (PipelineRef, TextureRef, ...);
only the calls that return something wait for the worker.  A panic on the worker is re-raised on the calling thread at
the next call that waits (synchronize, a factory, adapter_info, join)."""
import queue
import threading

from .viewport import Viewport


class _Ref:
    __slots__ = ("index",)

    def __init__(self, index):
        self.index = index

    def __repr__(self):
        return f"{type(self).__name__}({self.index})"


class PipelineRef(_Ref): pass
class AttributeRef(_Ref): pass
class InstancedRef(_Ref): pass
class UniformRef(_Ref): pass
class TextureRef(_Ref): pass
class ProgramRef(_Ref): pass


class TargetRef:
    """TargetRef::Screen | TargetRef::TextureRef(texture) (reference src/render_thread.rs:57)."""

    def __init__(self, texture=None):
        self.texture = texture

    def to_target(self, textures):
        from .target import Target
        return Target.Screen() if self.texture is None else Target.Texture(textures[self.texture.index])


class RenderThread:
    def __init__(self, width, height, hip_ordinal=0):
        self._window_size = (int(width), int(height))
        self._calls = queue.Queue()                 # unbounded, like the reference's function-call channel
        self._returns = queue.Queue(maxsize=1)      # bounded(1) return-value channel
        self._error = None
        self._thread = threading.Thread(target=self._run, args=(hip_ordinal,), daemon=True)
        self._thread.start()

    # ---- worker ----------------------------------------------------------------------------------------------------------
    def _run(self, hip_ordinal):
        from .renderer import Renderer
        renderer = None
        tables = {"pipeline": [], "attribute": [], "instanced": [], "uniform": [], "texture": [], "program": []}
        try:
            renderer = Renderer(self._window_size[0], self._window_size[1], hip_ordinal=hip_ordinal)
            while True:
                message = self._calls.get()
                if message is None:
                    break
                name, args, wants_reply = message
                try:
                    value = self._dispatch(renderer, tables, name, args)
                except BaseException as e:            # noqa: BLE001 - a panic on the render thread
                    self._error = e
                    value = e
                if wants_reply:
                    self._returns.put(value)
        except BaseException as e:                    # noqa: BLE001
            self._error = e
        finally:
            for t in tables.values():
                t.clear()
            if renderer is not None:
                renderer.close()

    @staticmethod
    def _dispatch(r, t, name, a):
        P, A, I, U, T, G = t["pipeline"], t["attribute"], t["instanced"], t["uniform"], t["texture"], t["program"]
        if name == "synchronize":
            return True
        if name == "resize_swap_chain":
            return r.resize_swap_chain(a[0])
        if name == "resize_texture":
            return r.resize_texture(T[a[0].index], a[1])
        if name == "render":
            return r.render(P[a[0].index], a[1], a[2], a[3])
        if name == "render_to":
            return r.render_to([x.to_target(T) for x in a[0]], P[a[1].index], a[2], a[3], a[4])
        if name in ("finish_frame", "flush"):
            return getattr(r, name)()
        if name in ("set_attribute", "set_instanced", "set_uniform", "set_texture"):
            return getattr(r, name)(P[a[0].index], a[1], a[2])
        if name == "set_part_of_texture":
            return r.set_part_of_texture(P[a[0].index], a[1], a[2], a[3], a[4])
        if name == "set_vsync":
            return r.set_vsync(a[0])
        if name == "set_msaa_samples":
            return r.set_msaa_samples(P[a[0].index], a[1])
        if name == "start_recording":
            return r.start_recording([P[x.index] for x in a[0]], a[1], a[2], a[3])
        if name == "stop_recording":
            return r.stop_recording([P[x.index] for x in a[0]])
        if name == "adapter_info":
            return r.adapter_info()
        if name == "pipeline":
            P.append(r.pipeline(G[a[0].index], a[1], a[2], a[3], [x.to_target(T) for x in a[4]]))
            return PipelineRef(len(P) - 1)
        if name == "attribute":
            A.append(r.attribute(a[0], a[1])); return AttributeRef(len(A) - 1)
        if name == "instanced":
            I.append(r.instanced()); return InstancedRef(len(I) - 1)
        if name == "uniform":
            U.append(r.uniform()); return UniformRef(len(U) - 1)
        if name == "texture":
            T.append(r.texture(*a)); return TextureRef(len(T) - 1)
        if name == "program":
            G.append(r.program(a[0], a[1], [A[x.index] for x in a[2]], [I[x.index] for x in a[3]],
                               [(U[x.index], v) for x, v in a[4]], [(T[x.index], v) for x, v in a[5]]))
            return ProgramRef(len(G) - 1)
        raise ValueError(name)

    # ---- caller side -------------------------------------------------------------------------------------------------------
    def _send(self, name, *args):
        self._calls.put((name, args, False))

    def _call(self, name, *args):
        self._calls.put((name, args, True))
        value = self._returns.get()
        if isinstance(value, BaseException):
            raise value
        if self._error is not None:
            raise self._error
        return value

    def join(self):
        """Ends the worker after everything queued has run (reference src/render_thread.rs:173-176)."""
        self._calls.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error

    def synchronize(self):
        self._call("synchronize")

    def window_size(self):
        return self._window_size

    def resize_swap_chain(self, new_size):
        self._window_size = (int(new_size[0]), int(new_size[1]))
        self._send("resize_swap_chain", self._window_size)

    def resize_texture(self, texture, new_size): self._send("resize_texture", texture, new_size)
    def render(self, pipeline, clear_color, viewport, count): self._send("render", pipeline, clear_color, viewport, count)
    def render_to(self, targets, pipeline, clear_color, viewport, count): self._send("render_to", targets, pipeline, clear_color, viewport, count)
    def finish_frame(self): self._send("finish_frame")
    def flush(self): self._send("flush")
    def set_attribute(self, pipeline, location, data): self._send("set_attribute", pipeline, location, data)
    def set_instanced(self, pipeline, index_tuple, data): self._send("set_instanced", pipeline, index_tuple, data)
    def set_uniform(self, pipeline, index_tuple, data): self._send("set_uniform", pipeline, index_tuple, data)
    def set_texture(self, pipeline, index_tuple, layers_data): self._send("set_texture", pipeline, index_tuple, layers_data)
    def set_part_of_texture(self, pipeline, index_tuple, offset, size, data): self._send("set_part_of_texture", pipeline, index_tuple, offset, size, data)
    def set_vsync(self, boolean): self._send("set_vsync", boolean)
    def set_msaa_samples(self, pipeline, msaa_samples): self._send("set_msaa_samples", pipeline, msaa_samples)

    def start_recording(self, pipelines, clear_color, max_buffer_size_in_megabytes, process_function):
        """process_function runs on the render thread inside finish_frame (it must be safe to call from there)."""
        self._send("start_recording", pipelines, clear_color, max_buffer_size_in_megabytes, process_function)

    def stop_recording(self, pipelines): self._send("stop_recording", pipelines)
    def adapter_info(self): return self._call("adapter_info")
    def pipeline(self, program, blend_mode, primitive, msaa_samples, targets): return self._call("pipeline", program, blend_mode, primitive, msaa_samples, targets)
    def attribute(self, location, size): return self._call("attribute", location, size)
    def instanced(self): return self._call("instanced")
    def uniform(self): return self._call("uniform")

    def texture(self, width, height, layers, filter_mode, format, renderable, copyable, with_sampler):
        return self._call("texture", width, height, layers, filter_mode, format, renderable, copyable, with_sampler)

    def program(self, vert, frag, attributes, instances, uniforms, textures):
        return self._call("program", vert, frag, attributes, instances, uniforms, textures)

    def viewport(self, aspect_x, aspect_y):
        return Viewport(aspect_x, aspect_y, float(self._window_size[0]), float(self._window_size[1]))

    screen_target = staticmethod(lambda: TargetRef())
    texture_target = staticmethod(lambda texture: TargetRef(texture))
