"""The reference ships two SPIR-V modules for its quads example (examples/quads/hello.vert.spirv, hello.frag.spirv: the
only shader artefacts it holds).  tools/make_spirv_facts.py decoded them in the build container into
tests/golden/spirv_facts.json; these tests check the HIP restatements (renderer_amd/shaders/hello.*.hip) and the slot
numbering of the host mirror against those facts."""
import json
import os
import re
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FACTS = json.load(open(os.path.join(HERE, "golden", "spirv_facts.json")))
SHADERS = os.path.join(ROOT, "renderer_amd", "shaders")


def source(name):
    text = open(os.path.join(SHADERS, name)).read()
    return "\n".join(line for line in text.split("\n") if not line.lstrip().startswith("//"))      # code only


def variables(module, **match):
    return [v for v in FACTS[module]["variables"] if all(v.get(k) == val for k, val in match.items())]


def test_vertex_module_facts():
    m = FACTS["hello.vert.spirv"]
    assert m["entry_point"] == {"model": "Vertex", "name": "main"} and m["version"] == "1.0"
    inputs = {v["location"]: v["type"] for v in variables("hello.vert.spirv", storage="Input") if "location" in v}
    assert inputs == {0: "vec2", 1: "vec2"}                       # a_position, a_tex_coord
    outputs = {v["location"]: v["type"] for v in variables("hello.vert.spirv", storage="Output") if "location" in v}
    assert outputs == {0: "vec2"}                                 # v_tex_coord
    (ssbo,) = [v for v in m["variables"] if v.get("block") == "BufferBlock"]
    assert (ssbo["set"], ssbo["binding"]) == (0, 0)
    assert ssbo["members"] == [{"ArrayStride": 8, "NonWritable": True, "Offset": 0, "type": "vec2[]"}]
    assert variables("hello.vert.spirv", builtin="InstanceIndex")
    assert m["fadd"] == 1 and m["image_sample_implicit_lod"] == 0


def test_fragment_module_facts():
    m = FACTS["hello.frag.spirv"]
    assert m["entry_point"] == {"model": "Fragment", "name": "main"}
    (tex,) = variables("hello.frag.spirv", type="texture2D")
    (smp,) = variables("hello.frag.spirv", type="sampler")
    assert (tex["set"], tex["binding"]) == (0, 1) and (smp["set"], smp["binding"]) == (0, 2)
    outputs = {v["location"]: v["type"] for v in variables("hello.frag.spirv", storage="Output")}
    assert outputs == {0: "vec4", 1: "vec4"}                      # f_color, f_recording
    inputs = {v["location"]: v["type"] for v in variables("hello.frag.spirv", storage="Input")}
    assert inputs == {0: "vec2"}
    assert m["image_sample_implicit_lod"] == 1 and m["sampled_image"] == 1 and m["stores"] == 2


def test_hip_vertex_restatement_agrees():
    src = source("hello.vert.hip")
    # attributes by location, both vec2
    assert sorted(int(x) for x in re.findall(r"attribute2\((\d+)\)", src)) == [0, 1]
    # the storage buffer: set 0, binding 0, elements of 8 bytes, indexed by the instance index, never written
    (elem, s, b), = re.findall(r"storage<(\w+)>\((\d+), (\d+)\)", src)
    assert (elem, int(s), int(b)) == ("float2", 0, 0)
    stride = {"float": 4, "float2": 8, "float4": 16}[elem]
    (ssbo,) = [v for v in FACTS["hello.vert.spirv"]["variables"] if v.get("block") == "BufferBlock"]
    assert stride == ssbo["members"][0]["ArrayStride"]
    assert re.search(r"storage<float2>\(0, 0\)\[c\.instance_index\]", src)
    assert "const float2 i_offset" in src                                  # read-only use
    # one varying, a vec2, and exactly one addition per component of the position
    assert re.search(r"struct Varyings \{ float2 tex_coord; \};", src)
    assert src.count(" + ") == 2                                           # vec2 FAdd = one add per component
    assert "0.0f, 1.0f)" in src                                            # gl_Position = (x, y, 0, 1)


def test_hip_fragment_restatement_agrees():
    src = source("hello.frag.hip")
    (tex,) = re.findall(r"texture2d\((\d+), (\d+)\)", src)
    (smp,) = re.findall(r"sampler\((\d+), (\d+)\)", src)
    assert tuple(map(int, tex)) == (0, 1) and tuple(map(int, smp)) == (0, 2)
    assert src.count(".sample(") == FACTS["hello.frag.spirv"]["image_sample_implicit_lod"]
    assert re.search(r"#define RH_FRAGMENT_OUTPUTS 2", src)
    assert sorted(int(x) for x in re.findall(r"out\.color\[(\d+)\] =", src)) == [0, 1]      # both outputs get the sample


def test_slot_numbering_of_the_example_matches_the_bindings():
    """instanced..., uniforms..., then per texture: texture [, sampler]; (set, binding) = (flat / 4, flat % 4)
    (reference src/pipeline.rs:90-131): the example's program puts i_offset at (0, 0), t_texture at (0, 1), s_sampler at
    (0, 2), which is what the modules declare."""
    from renderer_amd.pipeline import BINDINGS_PER_GROUP
    from renderer_amd.renderer import texture_index
    program = types.SimpleNamespace(instances=[object()], uniforms=[], textures=[(types.SimpleNamespace(sampler=object()), None)])
    flat = {"i_offset": 0, "t_texture": len(program.instances) + len(program.uniforms)}
    flat["s_sampler"] = flat["t_texture"] + 1
    where = {k: (v // BINDINGS_PER_GROUP, v % BINDINGS_PER_GROUP) for k, v in flat.items()}
    (ssbo,) = [v for v in FACTS["hello.vert.spirv"]["variables"] if v.get("block") == "BufferBlock"]
    (tex,) = variables("hello.frag.spirv", type="texture2D")
    (smp,) = variables("hello.frag.spirv", type="sampler")
    assert where == {"i_offset": (ssbo["set"], ssbo["binding"]), "t_texture": (tex["set"], tex["binding"]), "s_sampler": (smp["set"], smp["binding"])}
    assert texture_index(flat["t_texture"], program) == 0
