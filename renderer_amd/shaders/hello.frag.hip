// HIP restatement of the reference's examples/quads/hello.frag:
//   layout(set=0, binding=1) uniform texture2D t_texture;  layout(set=0, binding=2) uniform sampler s_sampler;
//   f_color = texture(sampler2D(t_texture, s_sampler), v_tex_coord);  f_recording = f_color;
#define RH_FRAGMENT_OUTPUTS 2

RH_DEVICE void fragment_main(const rh::FragmentContext& c, const Varyings& v, rh::Outputs& out) {
    const float4 color = c.texture2d(0, 1).sample(c.sampler(0, 2), v.tex_coord);
    out.color[0] = color;      // f_color
    out.color[1] = color;      // f_recording
}
