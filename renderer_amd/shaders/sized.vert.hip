// hello.vert with a per-instance size: the storage record is vec4 {x, y, half_width, half_height} and
//   gl_Position = vec4(a_position * half_size + xy, 0.0, 1.0)
// (a_position is the unit quad +-1).  Used by the tiny-quad benchmark scene.
struct Varyings { float2 tex_coord; };

RH_DEVICE float4 vertex_main(const rh::VertexContext& c, Varyings& v) {
    const float2 a_position = c.attribute2(0);
    const float2 a_tex_coord = c.attribute2(1);
    const float4 r = c.storage<float4>(0, 0)[c.instance_index];
    v.tex_coord = a_tex_coord;
    return make_float4(__builtin_fmaf(a_position.x, r.z, r.x), __builtin_fmaf(a_position.y, r.w, r.y), 0.0f, 1.0f);
}
