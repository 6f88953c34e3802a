// The reference's quads example (reference examples/quads_main_thread.rs:12-164) on the C++ host mirror: same calls in
// the same order.  Differences forced by the platform: there is no window or event loop (the frames are rendered
// offscreen in a plain loop), Program takes gfx950 code objects (hello.vert.co / hello.frag.co, built by
// Compiler::compile_shaders = hipcc --genco) instead of SPIR-V, and the recorded frames are compressed on the GPU.
//
// usage: quads_main_thread <shader dir> <letter_f.png> <out dir> [frames=60] [width=1920] [height=1080]
// Writes <out dir>/frame_<n>.png (the screen after the last frame, from the recording of that frame) and the
// compressed recording <out dir>/<timestamp>--<i>.sz.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

#include "png_codec.hpp"
#include "renderer.hpp"

using renderer::Renderer;

constexpr uint32_t A_POSITION = 0, A_TEX_COORD = 1;
const std::pair<size_t, size_t> I_OFFSET = { 0, 0 };      // set 0, binding 0
const std::pair<size_t, size_t> T_TEXTURE = { 0, 1 };     // set 0, binding 1; the next set begins after 4 bindings

static std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

// chunk[c] = round(chunk[c] * (alpha / 255)) in f32, round half away from zero (examples/quads_main_thread.rs:181-189)
static void premultiply_alpha(std::vector<uint8_t>& buffer) {
    for (size_t i = 0; i + 3 < buffer.size(); i += 4) {
        const float alpha = (float)buffer[i + 3] / 255.0f;
        for (int c = 0; c < 3; c++) buffer[i + c] = (uint8_t)roundf((float)buffer[i + c] * alpha);
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <shader dir> <letter_f.png> <out dir> [frames] [width] [height] [msaa samples]\n", argv[0]); return 2; }
    const std::string shader_dir = argv[1], image_path = argv[2], out_dir = argv[3];
    const int frames = argc > 4 ? atoi(argv[4]) : 60;
    const uint32_t win_w = argc > 5 ? (uint32_t)atoi(argv[5]) : 1920, win_h = argc > 6 ? (uint32_t)atoi(argv[6]) : 1080;
    // (not in the reference's example) more than one sample: a multisampled pass cannot carry the single-sampled
    // recording attachment, so nothing is recorded and the PNG is the final screen instead of the last captured frame
    const uint32_t msaa_arg = argc > 7 ? (uint32_t)atoi(argv[7]) : 1;
    try {
        Renderer renderer(win_w, win_h);

        const auto vert = read_file(shader_dir + "/hello.vert.co");
        const auto frag = read_file(shader_dir + "/hello.frag.co");

        uint32_t width = 0, height = 0;
        std::vector<uint8_t> image = renderer::decode_png_rgba8(read_file(image_path), width, height);
        premultiply_alpha(image);

        const auto format = Renderer::rgba_u8();
        const auto filter = Renderer::linear_filtering();
        auto a_position = renderer.attribute(A_POSITION, 2);
        auto a_tex_coord = renderer.attribute(A_TEX_COORD, 2);
        auto i_offset = renderer.instanced();
        auto t_texture = renderer.texture(width, height, 1, filter, format, false, false, true);

        auto program = renderer.program(vert, frag, { a_position, a_tex_coord }, { i_offset }, {},
                                        { { t_texture, Renderer::visible_to_fragment_shader() } });
        const auto blend_mode = Renderer::pre_multiplied_blend();
        const auto primitive = Renderer::triangle_strip_primitive();
        const uint32_t msaa_samples = msaa_arg;
        const auto target = Renderer::screen_target();
        auto pipeline = renderer.pipeline(program, blend_mode, primitive, msaa_samples, { target });
        const auto clear_color = Renderer::clear_color(0., 0., 0., 1.);

        renderer.set_attribute(pipeline, A_POSITION, { -0.1f, -0.1f, -0.1f, 0.1f, 0.1f, -0.1f, 0.1f, 0.1f });
        renderer.set_attribute(pipeline, A_TEX_COORD, { 0.f, 1.f, 0.f, 0.f, 1.f, 1.f, 1.f, 0.f });
        renderer.set_texture(pipeline, T_TEXTURE, { &image });

        // 1) compress the raw frame data to disk (on the GPU: the frames never leave device memory uncompressed);
        //    the last frame is also kept mapped on the host to write it out as a PNG below
        renderer::Compressor compressor(renderer, out_dir, "1970-01-01T00_00_00.000Z", 0, 4);
        std::vector<uint8_t> last_png;
        if (msaa_samples == 1) renderer.start_recording({ &pipeline }, &clear_color, 500.f, [&](renderer::VideoFrame video_frame) {
            if ((int)video_frame.frame_number == frames) last_png = renderer::PngEncoder::encode_to_bytes(video_frame);
            compressor.compress_to_disk(std::move(video_frame));
        });

        // the start position of each quad and its velocity in the x, y directions
        float x1[2] = { 0.3f, 0.015f }, y1[2] = { -0.3f, 0.01f }, x2[2] = { -0.5f, 0.005f }, y2[2] = { -0.1f, 0.02f };
        for (int frame = 0; frame < frames; frame++) {
            x1[0] += x1[1]; if (x1[0] > 0.9f || x1[0] < -0.9f) x1[1] *= -1.f;
            y1[0] += y1[1]; if (y1[0] > 0.9f || y1[0] < -0.9f) y1[1] *= -1.f;
            x2[0] += x2[1]; if (x2[0] > 0.9f || x2[0] < -0.9f) x2[1] *= -1.f;
            y2[0] += y2[1]; if (y2[0] > 0.9f || y2[0] < -0.9f) y2[1] *= -1.f;

            renderer.set_instanced(pipeline, I_OFFSET, { x1[0], y1[0], x2[0], y2[0] });
            const auto viewport = renderer.viewport(1.f, 1.f);
            renderer.render(pipeline, &clear_color, &viewport, { 2, 4 });
            renderer.finish_frame();
        }
        // frames are delivered during a later finish_frame at the earliest: drain what is still in flight
        renderer.synchronize();
        if (msaa_samples == 1) {
            renderer.recorder->process_mapped_buffers();
            renderer.stop_recording({ &pipeline });
        } else {
            std::vector<uint8_t> screen = renderer.read_screen();                    // BGRA
            for (size_t i = 0; i + 3 < screen.size(); i += 4) std::swap(screen[i], screen[i + 2]);
            last_png = renderer::PngEncoder::encode_rows(screen.data(), win_w, win_h, (size_t)win_w * 4, (size_t)win_w * 4);
        }
        compressor.finish();

        const std::string png_path = out_dir + "/frame_" + std::to_string(frames) + ".png";
        std::ofstream(png_path, std::ios::binary).write(reinterpret_cast<const char*>(last_png.data()), (std::streamsize)last_png.size());
        printf("%s: %d frames on %s, %zu captured, %zu raw bytes -> %zu compressed, wrote %s (%zu bytes)\n", argv[0], frames,
               renderer.adapter_info().c_str(), compressor.frames_captured, compressor.raw_video_size, compressor.compressed_size,
               png_path.c_str(), last_png.size());
    } catch (const renderer::Panic& p) {
        fprintf(stderr, "panic: %s\n", p.what());
        return 1;
    }
    return 0;
}
