// (same vertex stage as hello.vert; paired with crowd.frag, which binds twenty slots over five bind groups)
//   layout(set=0, binding=0) readonly buffer _0 { vec2 i_offset[]; };
//   layout(location=0) in vec2 a_position;  layout(location=1) in vec2 a_tex_coord;
//   v_tex_coord = a_tex_coord;  gl_Position = vec4(a_position + i_offset[gl_InstanceIndex], 0.0, 1.0);
struct Varyings { float2 tex_coord; };

RH_DEVICE float4 vertex_main(const rh::VertexContext& c, Varyings& v) {
    const float2 a_position = c.attribute2(0);
    const float2 a_tex_coord = c.attribute2(1);
    const float2 i_offset = c.storage<float2>(0, 0)[c.instance_index];
    v.tex_coord = a_tex_coord;
    return make_float4(a_position.x + i_offset.x, a_position.y + i_offset.y, 0.0f, 1.0f);
}
