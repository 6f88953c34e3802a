// The quads example driven through the RenderThread proxy: every call is a message to a worker thread that owns the
// Renderer, objects are index handles.  Mirrors reference examples/quads_render_thread.rs (same scene and motion as
// quads_main_thread); instead of opening a window it renders <frames> frames offscreen and writes the last screen as a
// PNG.  usage: quads_render_thread <shader dir> <letter_f.png> <out dir> [frames] [width] [height] [msaa samples]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "png_codec.hpp"
#include "render_thread.hpp"

using renderer::Renderer;
using renderer::RenderThread;

static std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// the example multiplies colour by alpha on the CPU, in f32 with round half away from zero
static void premultiply_alpha(std::vector<uint8_t>& rgba) {
    for (size_t i = 0; i + 3 < rgba.size(); i += 4) {
        const float alpha = (float)rgba[i + 3] / 255.0f;
        for (int k = 0; k < 3; k++) rgba[i + k] = (uint8_t)std::round((float)rgba[i + k] * alpha);
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <shader dir> <letter_f.png> <out dir> [frames] [width] [height] [msaa samples]\n", argv[0]); return 2; }
    const std::string shader_dir = argv[1], image_path = argv[2], out_dir = argv[3];
    const int frames = argc > 4 ? atoi(argv[4]) : 60;
    const uint32_t win_w = argc > 5 ? (uint32_t)atoi(argv[5]) : 1920, win_h = argc > 6 ? (uint32_t)atoi(argv[6]) : 1080;
    const uint32_t msaa_samples = argc > 7 ? (uint32_t)atoi(argv[7]) : 1;
    try {
        RenderThread renderer(win_w, win_h);

        uint32_t width = 0, height = 0;
        std::vector<uint8_t> image = renderer::decode_png_rgba8(read_file(image_path), width, height);
        premultiply_alpha(image);

        const auto a_position = renderer.attribute(0, 2);
        const auto a_tex_coord = renderer.attribute(1, 2);
        const auto i_offset = renderer.instanced();
        const auto t_texture = renderer.texture(width, height, 1, Renderer::linear_filtering(), Renderer::rgba_u8(), false, false, true);
        const auto program = renderer.program(read_file(shader_dir + "/hello.vert.co"), read_file(shader_dir + "/hello.frag.co"),
                                              { a_position, a_tex_coord }, { i_offset }, {}, { { t_texture, Renderer::visible_to_fragment_shader() } });
        const auto pipeline = renderer.pipeline(program, Renderer::pre_multiplied_blend(), Renderer::triangle_strip_primitive(), msaa_samples,
                                                { RenderThread::screen_target() });
        const auto clear_color = Renderer::clear_color(0., 0., 0., 1.);

        renderer.set_attribute(pipeline, 0, { -0.1f, -0.1f, -0.1f, 0.1f, 0.1f, -0.1f, 0.1f, 0.1f });
        renderer.set_attribute(pipeline, 1, { 0.f, 1.f, 0.f, 0.f, 1.f, 1.f, 1.f, 0.f });
        renderer.set_texture(pipeline, { 0, 1 }, { image });

        float x1[2] = { 0.3f, 0.015f }, y1[2] = { -0.3f, 0.01f }, x2[2] = { -0.5f, 0.005f }, y2[2] = { -0.1f, 0.02f };
        for (int frame = 0; frame < frames; frame++) {
            x1[0] += x1[1]; if (x1[0] > 0.9f || x1[0] < -0.9f) x1[1] *= -1.f;
            y1[0] += y1[1]; if (y1[0] > 0.9f || y1[0] < -0.9f) y1[1] *= -1.f;
            x2[0] += x2[1]; if (x2[0] > 0.9f || x2[0] < -0.9f) x2[1] *= -1.f;
            y2[0] += y2[1]; if (y2[0] > 0.9f || y2[0] < -0.9f) y2[1] *= -1.f;
            renderer.set_instanced(pipeline, { 0, 0 }, { x1[0], y1[0], x2[0], y2[0] });
            const auto viewport = renderer.viewport(1.f, 1.f);
            renderer.render(pipeline, &clear_color, &viewport, { 2, 4 });
            renderer.finish_frame();
        }
        renderer.synchronize();                               // the queue has drained: every frame is rendered

        std::vector<uint8_t> screen = renderer.read_screen();     // BGRA
        for (size_t i = 0; i + 3 < screen.size(); i += 4) std::swap(screen[i], screen[i + 2]);
        const std::vector<uint8_t> png = renderer::PngEncoder::encode_rows(screen.data(), win_w, win_h, (size_t)win_w * 4, (size_t)win_w * 4);
        const std::string png_path = out_dir + "/frame_" + std::to_string(frames) + ".png";
        std::ofstream(png_path, std::ios::binary).write(reinterpret_cast<const char*>(png.data()), (std::streamsize)png.size());
        printf("%s: %d frames through the render thread on %s, wrote %s (%zu bytes)\n", argv[0], frames, renderer.adapter_info().c_str(), png_path.c_str(), png.size());
        renderer.join();
    } catch (const renderer::Panic& p) {
        fprintf(stderr, "panic: %s\n", p.what());
        return 1;
    }
    return 0;
}
