// Quads in space (test shader for the clip volume: triangles that cross the w = 0 and the depth planes).
//   layout(set=0, binding=0) readonly buffer Records { vec4 r[]; };   // {cx, cy, half, depth}
//   layout(set=0, binding=1) uniform Params { vec4 tint; mat4 m; };
//   gl_Position = m * vec4(a_position.x * half + cx, a_position.y * half + cy, depth, 1)
//   v_tex_coord = a_tex_coord;  v_shade = 1
struct Varyings { float2 tex_coord; float shade; };

RH_DEVICE float4 vertex_main(const rh::VertexContext& c, Varyings& v) {
    const float2 a_position = c.attribute2(0);
    const float2 a_tex_coord = c.attribute2(1);
    const float4 r = c.storage<float4>(0, 0)[c.instance_index];
    const float* m = c.uniform<float>(0, 1) + 4;
    const float lx = __builtin_fmaf(a_position.x, r.z, r.x), ly = __builtin_fmaf(a_position.y, r.z, r.y), lz = r.w;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = __builtin_fmaf(m[i], lx, __builtin_fmaf(m[4 + i], ly, __builtin_fmaf(m[8 + i], lz, m[12 + i])));
    v.tex_coord = a_tex_coord;
    v.shade = 1.0f;
    return make_float4(o[0], o[1], o[2], o[3]);
}
