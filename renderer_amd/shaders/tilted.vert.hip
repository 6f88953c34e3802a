// Projective quads (test shader for the perspective path, a third varying and a uniform read by both stages).
//   layout(set=0, binding=0) readonly buffer Records { vec4 r[]; };   // {cx, cy, half, tilt}
//   layout(set=0, binding=1) uniform Params { vec4 tint; vec2 scale; };
//   w = a_position.x * tilt + 1
//   gl_Position = vec4((a_position.x * scale.x * half + cx) * w, (a_position.y * scale.y * half + cy) * w, 0, w)
//   v_tex_coord = a_tex_coord;  v_shade = a_position.y * 0.25 + 0.75
struct Varyings { float2 tex_coord; float shade; };

RH_DEVICE float4 vertex_main(const rh::VertexContext& c, Varyings& v) {
    const float2 a_position = c.attribute2(0);
    const float2 a_tex_coord = c.attribute2(1);
    const float4 r = c.storage<float4>(0, 0)[c.instance_index];
    const float* u = c.uniform<float>(0, 1);
    const float w = __builtin_fmaf(a_position.x, r.w, 1.0f);
    const float x = __builtin_fmaf(a_position.x * u[4], r.z, r.x), y = __builtin_fmaf(a_position.y * u[5], r.z, r.y);
    v.tex_coord = a_tex_coord;
    v.shade = __builtin_fmaf(a_position.y, 0.25f, 0.75f);
    return make_float4(x * w, y * w, 0.0f, w);
}
