#!/usr/bin/env python3
"""Reads the two SPIR-V modules the reference ships for its quads example (examples/quads/hello.vert.spirv,
hello.frag.spirv: what src/program.rs:43-48 loads) and writes the facts a shader restatement has to agree with to
tests/golden/spirv_facts.json: entry points, interface variables with their locations / descriptor sets / bindings,
array strides, NonWritable decorations, and the sampling instructions.  Run in the build container:
    python tools/make_spirv_facts.py /root/reference/examples/quads tests/golden/spirv_facts.json
The JSON is data about the binaries (no source text); the tests read only the JSON."""
import json
import struct
import sys

OP_NAME, OP_MEMBER_NAME, OP_ENTRY_POINT, OP_TYPE_FLOAT, OP_TYPE_VECTOR, OP_TYPE_RUNTIME_ARRAY, OP_TYPE_ARRAY = 5, 6, 15, 22, 23, 29, 28
OP_TYPE_STRUCT, OP_TYPE_POINTER, OP_VARIABLE, OP_DECORATE, OP_MEMBER_DECORATE = 30, 32, 59, 71, 72
OP_IMAGE_SAMPLE_IMPLICIT_LOD, OP_SAMPLED_IMAGE, OP_TYPE_IMAGE, OP_TYPE_SAMPLER, OP_STORE, OP_FADD = 87, 86, 25, 26, 62, 129
DECORATIONS = {2: "Block", 3: "BufferBlock", 6: "ArrayStride", 11: "BuiltIn", 24: "NonWritable", 30: "Location", 33: "Binding", 34: "DescriptorSet", 35: "Offset"}
STORAGE = {0: "UniformConstant", 1: "Input", 2: "Uniform", 3: "Output", 12: "StorageBuffer"}
MODELS = {0: "Vertex", 4: "Fragment"}
BUILTINS = {0: "Position", 43: "InstanceIndex", 42: "VertexIndex"}


def string_at(words, i):
    raw = b"".join(struct.pack("<I", w) for w in words[i:])
    return raw.split(b"\0", 1)[0].decode()


def parse(path):
    data = open(path, "rb").read()
    words = list(struct.unpack("<%dI" % (len(data) // 4), data))
    assert words[0] == 0x07230203, "not a SPIR-V module"
    facts = {"bytes": len(data), "version": "%d.%d" % ((words[1] >> 16) & 0xff, (words[1] >> 8) & 0xff), "generator": "0x%08x" % words[2]}
    names, decos, member_decos, variables, pointers, types, ops = {}, {}, {}, {}, {}, {}, {}
    i = 5
    while i < len(words):
        op, n = words[i] & 0xffff, words[i] >> 16
        a = words[i + 1:i + n]
        ops[op] = ops.get(op, 0) + 1
        if op == OP_NAME:
            names[a[0]] = string_at(a, 1)
        elif op == OP_ENTRY_POINT:
            facts["entry_point"] = {"model": MODELS.get(a[0], a[0]), "name": string_at(a, 2)}
        elif op == OP_DECORATE:
            decos.setdefault(a[0], {})[DECORATIONS.get(a[1], str(a[1]))] = a[2] if len(a) > 2 else True
        elif op == OP_MEMBER_DECORATE:
            member_decos.setdefault(a[0], {}).setdefault(a[1], {})[DECORATIONS.get(a[2], str(a[2]))] = a[3] if len(a) > 3 else True
        elif op == OP_TYPE_POINTER:
            pointers[a[0]] = (a[1], a[2])
        elif op == OP_TYPE_VECTOR:
            types[a[0]] = ("vector", a[1], a[2])
        elif op == OP_TYPE_FLOAT:
            types[a[0]] = ("float", a[1])
        elif op == OP_TYPE_RUNTIME_ARRAY:
            types[a[0]] = ("runtime_array", a[1])
        elif op == OP_TYPE_STRUCT:
            types[a[0]] = ("struct", list(a[1:]))
        elif op == OP_TYPE_IMAGE:
            types[a[0]] = ("image", a[2], a[4])           # dim, arrayed
        elif op == OP_TYPE_SAMPLER:
            types[a[0]] = ("sampler",)
        elif op == OP_VARIABLE:
            variables[a[1]] = (a[0], a[2])                # result type (a pointer), storage class
        i += n

    def describe(type_id):
        t = types.get(type_id)
        if t is None:
            return "?"
        if t[0] == "float":
            return "float"
        if t[0] == "vector":
            return "vec%d" % t[2]
        if t[0] == "image":
            return "texture2D" + ("Array" if t[2] else "") if t[1] == 1 else "image"
        if t[0] == "sampler":
            return "sampler"
        if t[0] == "runtime_array":
            return describe(t[1]) + "[]"
        if t[0] == "struct":
            return "struct{" + ",".join(describe(m) for m in t[1]) + "}"
        return t[0]

    out = []
    for vid, (ptr, storage) in sorted(variables.items()):
        pointee = pointers[ptr][1]
        d = decos.get(vid, {})
        v = {"name": names.get(vid, ""), "storage": STORAGE.get(storage, storage), "type": describe(pointee)}
        for key in ("Location", "DescriptorSet", "Binding"):
            if key in d:
                v[key.lower() if key != "DescriptorSet" else "set"] = d[key]
        if "BuiltIn" in d:
            v["builtin"] = BUILTINS.get(d["BuiltIn"], d["BuiltIn"])
        if types.get(pointee, ("",))[0] == "struct":
            sd = decos.get(pointee, {})
            v["block"] = "BufferBlock" if sd.get("BufferBlock") else ("Block" if sd.get("Block") else None)
            members = []
            for k, m in enumerate(types[pointee][1]):
                md = dict(member_decos.get(pointee, {}).get(k, {}))
                if types.get(m, ("",))[0] == "runtime_array":
                    md["ArrayStride"] = decos.get(m, {}).get("ArrayStride")
                    if md.get("BuiltIn") is not None:
                        md["BuiltIn"] = BUILTINS.get(md["BuiltIn"], md["BuiltIn"])
                elif "BuiltIn" in md:
                    md["BuiltIn"] = BUILTINS.get(md["BuiltIn"], md["BuiltIn"])
                members.append({"type": describe(m), **{k2: v2 for k2, v2 in md.items()}})
            v["members"] = members
        out.append(v)
    facts["variables"] = out
    facts["image_sample_implicit_lod"] = ops.get(OP_IMAGE_SAMPLE_IMPLICIT_LOD, 0)
    facts["sampled_image"] = ops.get(OP_SAMPLED_IMAGE, 0)
    facts["stores"] = ops.get(OP_STORE, 0)
    facts["fadd"] = ops.get(OP_FADD, 0)
    return facts


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    facts = {"hello.vert.spirv": parse(src + "/hello.vert.spirv"), "hello.frag.spirv": parse(src + "/hello.frag.spirv")}
    json.dump(facts, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(facts, indent=1, sort_keys=True))
