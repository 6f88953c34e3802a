"""Shader program: two code objects plus the ordered attribute / instanced / uniform / texture lists
(reference src/program.rs:23-48).  `vert` and `frag` are the bytes of gfx950 code objects produced by
Compiler.compile_shaders, loaded with hipModuleLoadData instead of wgpu's create_shader_module."""
import ctypes as C

from . import _ffi


class Program:
    def __init__(self, device, vert, frag, attributes, instances, uniforms, textures):
        self.device = device
        self.vertex_shader = self._create_shader_module(vert)
        self.fragment_shader = self._create_shader_module(frag)
        self.attributes = list(attributes)
        self.instances = list(instances)
        self.uniforms = list(uniforms)          # [(Uniform, Visibility)]
        self.textures = list(textures)          # [(Texture, Visibility)]

    def _create_shader_module(self, code):
        code = bytes(code)
        out = C.c_void_p()
        buf = C.create_string_buffer(code, len(code))
        _ffi.check(_ffi.lib().rh_module_load(self.device, buf, len(code), C.byref(out)))
        return out

    def latest_generations(self):
        """Generations of every bound resource, in binding order (reference src/program.rs:33-40)."""
        return ([a.buffer.generation for a in self.attributes] + [i.buffer.generation for i in self.instances] +
                [u.buffer.generation for (u, _) in self.uniforms] + [t.generation for (t, _) in self.textures])

    def __del__(self):
        try:
            for name in ("vertex_shader", "fragment_shader"):
                h = getattr(self, name, None)
                if h:
                    _ffi.lib().rh_module_release(h)
                    setattr(self, name, None)
        except Exception:
            pass
