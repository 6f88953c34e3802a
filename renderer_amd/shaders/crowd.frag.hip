// A fragment stage that reaches past the sixteenth binding slot (the reference numbers slots flat, four to a bind group,
// eight groups: src/pipeline.rs:25-27, 90-131): one instanced buffer (slot 0), seven uniforms (slots 1..7: a gain, then
// one vec4 weight per texture) and six textures with their samplers (slots 8..19), i.e. bind groups 0..4.
//   layout(set=0, binding=1) uniform Gain { float gain; };
//   layout(set=(1+i)/4, binding=(1+i)%4) uniform Weight_i { vec4 w_i; };                       i = 1..6
//   layout(set=(6+2i)/4, binding=(6+2i)%4) uniform texture2D t_i;  (next slot) uniform sampler s_i;   i = 1..6
//   f_color = gain * sum_i texture(sampler2D(t_i, s_i), v_tex_coord) * w_i   (accumulated in the order i = 1..6, one fma per channel)
#define RH_FRAGMENT_OUTPUTS 1

RH_DEVICE void fragment_main(const rh::FragmentContext& c, const Varyings& v, rh::Outputs& out) {
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) {
        const uint32_t us = 2u + i, ts = 8u + 2u * i;          // flat slots of the weight and of the texture (its sampler: the next one)
        const float4 w = c.uniform<float4>(us / 4u, us % 4u)[0];
        const float4 t = c.texture2d(ts / 4u, ts % 4u).sample(c.sampler((ts + 1u) / 4u, (ts + 1u) % 4u), v.tex_coord);
        acc.x = __builtin_fmaf(t.x, w.x, acc.x); acc.y = __builtin_fmaf(t.y, w.y, acc.y);
        acc.z = __builtin_fmaf(t.z, w.z, acc.z); acc.w = __builtin_fmaf(t.w, w.w, acc.w);
    }
    const float gain = c.uniform<float>(0, 1)[0];
    out.color[0] = make_float4(acc.x * gain, acc.y * gain, acc.z * gain, acc.w * gain);
}
