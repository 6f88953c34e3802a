// hello.frag sampling one layer of a 2D-array texture, the layer index coming from a uniform:
//   layout(set=0, binding=1) uniform Params { float layer; };
//   layout(set=0, binding=2) uniform texture2DArray t_texture;  layout(set=0, binding=3) uniform sampler s_sampler;
//   f_color = texture(sampler2DArray(t_texture, s_sampler), vec3(v_tex_coord, layer));  f_recording = f_color
// (the reference creates a D2Array view for textures with more than one layer, src/texture.rs:130-135)
#define RH_FRAGMENT_OUTPUTS 2

RH_DEVICE void fragment_main(const rh::FragmentContext& c, const Varyings& v, rh::Outputs& out) {
    const float lf = c.uniform<float>(0, 1)[0];
    const uint32_t layer = lf > 0.0f ? (uint32_t)lf : 0u;
    const float4 color = c.texture2d(0, 2).sample(c.sampler(0, 3), v.tex_coord, layer);
    out.color[0] = color;
    out.color[1] = color;
}
