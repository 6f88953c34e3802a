// Fragment stage of the quads-in-space test shader (the projective-quad one):
//   layout(set=0, binding=2) uniform texture2D t_texture;  layout(set=0, binding=3) uniform sampler s_sampler;
//   f_color = (texture(sampler2D(t_texture, s_sampler), v_tex_coord) * tint) * v_shade;  f_recording = f_color
#define RH_FRAGMENT_OUTPUTS 2

RH_DEVICE void fragment_main(const rh::FragmentContext& c, const Varyings& v, rh::Outputs& out) {
    const float4 t = c.texture2d(0, 2).sample(c.sampler(0, 3), v.tex_coord);
    const float* u = c.uniform<float>(0, 1);
    const float4 color = make_float4((t.x * u[0]) * v.shade, (t.y * u[1]) * v.shade, (t.z * u[2]) * v.shade, (t.w * u[3]) * v.shade);
    out.color[0] = color;
    out.color[1] = color;
}
