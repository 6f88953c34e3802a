// C++ host-side mirror of the reference crate's public API (reference src/lib.rs:23-59) above the C ABI of
// librenderer_hip.so (include/renderer_hip.h).  The reference is Rust; every type here keeps the reference's name,
// constructor arguments and panics (thrown as renderer::Panic), so a program written against the crate reads the same:
// compare examples/quads_main_thread.cpp with the reference's examples/quads_main_thread.rs.
//
// Rust's `#[derive(Clone)]` handle types over Rc (Buffer, Texture, Program, Attribute, Instanced, Uniform) are value
// types over std::shared_ptr here.  There is no window: Renderer(width, height) replaces Renderer::new(window) and
// Target::Screen is an offscreen BGRA8 image of that size.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "renderer_hip.h"

namespace renderer {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(int rc) { if (rc != 0) throw Panic(rh_last_error()); }

// ---- enums and small value types ---------------------------------------------------------------------------------------
// reference src/format.rs:3-42 (declaration order = bincode discriminant)
enum class Format : uint32_t { RU8 = 0, BgraU8 = 1, RgbaU8 = 2, RgbaF16 = 3, RgbaF32 = 4 };
inline uint32_t channels(Format f) { return f == Format::RU8 ? 1u : 4u; }
inline uint32_t bytes_per_channel(Format f) { return f == Format::RgbaF16 ? 2u : (f == Format::RgbaF32 ? 4u : 1u); }
inline uint32_t bytes_per_texel(Format f) { return channels(f) * bytes_per_channel(f); }

enum class FilterMode : uint32_t { Linear = RH_FILTER_LINEAR, Nearest = RH_FILTER_NEAREST };          // src/filter_mode.rs:8-20
enum class Primitive : uint32_t { Triangle = RH_TOPOLOGY_TRIANGLE_LIST, TriangleStrip = RH_TOPOLOGY_TRIANGLE_STRIP };   // src/primitive.rs:8-13
enum class Visibility : uint32_t { VertexShader = RH_STAGE_VERTEX, FragmentShader = RH_STAGE_FRAGMENT,
                                   BothShaders = RH_STAGE_VERTEX | RH_STAGE_FRAGMENT };                // src/visibility.rs:9-15

// reference src/blend_mode.rs:8-40: out = src * src_factor + dst * dst_factor, colour and alpha alike
struct BlendMode {
    uint32_t src_factor, dst_factor;
    static BlendMode additive() { return { RH_BLEND_ONE, RH_BLEND_ONE }; }
    static BlendMode pre_multiplied_alpha() { return { RH_BLEND_ONE, RH_BLEND_ONE_MINUS_SRC_ALPHA }; }
    rh_color_target_desc state(Format format) const { return { (uint32_t)format, src_factor, dst_factor }; }
};

struct ClearColor { double inner[4]; };                                                               // src/clear_color.rs:7-16

// reference src/viewport.rs:10-28: letterbox, all f32, no rounding
struct Viewport {
    float margin_x = 0, margin_y = 0, width = 0, height = 0;
    Viewport(float aspect_x, float aspect_y, float max_width, float max_height) {
        const float current_aspect = max_width / max_height, desired_aspect = aspect_x / aspect_y;
        width = max_width; height = max_height;
        if (current_aspect > desired_aspect) { width = height * desired_aspect; margin_x = (max_width - width) / 2.0f; }
        else { height = width / desired_aspect; margin_y = (max_height - height) / 2.0f; }
    }
};

// ---- Buffer: growable f32 device buffer with a generation counter (reference src/buffer.rs:16-67) -----------------------
class Buffer {
public:
    static constexpr size_t INITIAL_SIZE = 4 * 16, HEADROOM = 4 * 256;
    struct Inner {
        rh_device* device; uint32_t usage; rh_buffer* handle = nullptr; size_t size = INITIAL_SIZE;
        uint32_t generation = 0; uint64_t previous = UINT64_MAX;
        ~Inner() { if (handle) rh_buffer_release(handle); }
    };
    std::shared_ptr<Inner> inner;

    Buffer() = default;
    Buffer(rh_device* device, uint32_t usage) : inner(std::make_shared<Inner>()) {
        inner->device = device; inner->usage = usage;
        check(rh_buffer_create(device, INITIAL_SIZE, usage, nullptr, 0, &inner->handle));
    }
    void set_data(const float* data, size_t count, uint64_t flushes) const {
        Inner& b = *inner;
        if (flushes == b.previous) throw Panic("Wasteful call to buffer.set_data(). The previous data would be overridden.");
        b.previous = flushes;
        const size_t bytes = count * sizeof(float);
        if (bytes > b.size) {
            size_t new_size = 1; while (new_size < bytes + HEADROOM) new_size <<= 1;      // next_power_of_two
            rh_buffer* fresh = nullptr;
            check(rh_buffer_create(b.device, new_size, b.usage, data, bytes, &fresh));
            rh_buffer_release(b.handle);                       // recorded command buffers keep the old one alive
            b.handle = fresh; b.size = new_size; b.generation += 1;
        } else if (bytes) check(rh_buffer_write(b.handle, 0, data, bytes));
    }
};

struct Attribute {                                                                                    // src/attribute.rs:10-29
    Buffer buffer; uint32_t location = 0, size = 0;
    Attribute() = default;
    Attribute(rh_device* device, uint32_t location, uint32_t size)
        : buffer(device, RH_BUFFER_VERTEX | RH_BUFFER_COPY_DST), location(location), size(size) {
        if (size < 1 || size > 4) throw Panic("Attribute size must be 1, 2, 3 or 4.");
    }
};

struct Instanced {                                                                                    // src/instanced.rs:9-38
    Buffer buffer;
    Instanced() = default;
    explicit Instanced(rh_device* device) : buffer(device, RH_BUFFER_STORAGE | RH_BUFFER_COPY_DST) {}
    rh_slot_desc binding() const { return { RH_SLOT_STORAGE, RH_STAGE_VERTEX, buffer.inner->handle }; }
};

struct Uniform {                                                                                      // src/uniform.rs:9-38
    Buffer buffer;
    Uniform() = default;
    explicit Uniform(rh_device* device) : buffer(device, RH_BUFFER_UNIFORM | RH_BUFFER_COPY_DST) {}
    rh_slot_desc binding(Visibility v) const { return { RH_SLOT_UNIFORM, (uint32_t)v, buffer.inner->handle }; }
};

// ---- Texture (reference src/texture.rs:23-158) ---------------------------------------------------------------------------
class Texture {
public:
    struct Inner {
        rh_device* device; uint32_t width, height, layers; FilterMode filter_mode; Format format; uint32_t msaa_samples;
        bool renderable, copyable; uint32_t generation = 0; rh_texture* handle = nullptr; rh_sampler* sampler = nullptr;
        ~Inner() { if (handle) rh_texture_release(handle); if (sampler) rh_sampler_release(sampler); }
        uint32_t usage() const {
            return RH_TEXTURE_BINDING | RH_TEXTURE_COPY_DST | (renderable ? RH_TEXTURE_RENDER_ATTACHMENT : 0) | (copyable ? RH_TEXTURE_COPY_SRC : 0);
        }
        void create() { check(rh_texture_create(device, width, height, layers, (uint32_t)format, msaa_samples, usage(), &handle)); }
    };
    std::shared_ptr<Inner> inner;

    Texture() = default;
    Texture(rh_device* device, uint32_t width, uint32_t height, uint32_t layers, FilterMode filter_mode, Format format,
            uint32_t msaa_samples, bool renderable, bool copyable, bool with_sampler) : inner(std::make_shared<Inner>()) {
        Inner& t = *inner;
        t.device = device; t.width = width; t.height = height; t.layers = layers; t.filter_mode = filter_mode; t.format = format;
        t.msaa_samples = msaa_samples; t.renderable = renderable; t.copyable = copyable;
        t.create();
        if (with_sampler) check(rh_sampler_create(device, (uint32_t)filter_mode, &t.sampler));
    }
    // reallocates (contents are lost) and bumps the generation, only when width or height change (src/texture.rs:34-43)
    void resize(uint32_t width, uint32_t height, uint32_t layers) const {
        Inner& t = *inner;
        if (t.width == width && t.height == height) return;
        if (width == 0 || height == 0 || layers == 0) return;
        t.width = width; t.height = height; t.layers = layers;
        rh_texture* old = t.handle; t.handle = nullptr;
        t.create();
        rh_texture_release(old);
        t.generation += 1;
    }
    // uploads tightly packed rows into (x, y, layer); size (0, 0) = the whole layer (src/texture.rs:45-56)
    void set_data(uint32_t x, uint32_t y, uint32_t layer, uint32_t w, uint32_t h, const uint8_t* data, size_t len) const {
        const Inner& t = *inner;
        if (w == 0 && h == 0) { w = t.width; h = t.height; }
        check(rh_texture_write(t.handle, x, y, layer, w, h, data, len, w * bytes_per_texel(t.format)));
    }
    std::vector<uint8_t> read(uint32_t layer = 0) const {
        const Inner& t = *inner;
        std::vector<uint8_t> out((size_t)t.width * t.height * bytes_per_texel(t.format));
        check(rh_texture_read(t.handle, layer, out.data(), out.size()));
        return out;
    }
    rh_slot_desc texture_binding(Visibility v) const { return { RH_SLOT_TEXTURE, (uint32_t)v, inner->handle }; }
    rh_slot_desc sampler_binding(Visibility v) const { return { RH_SLOT_SAMPLER, (uint32_t)v, inner->sampler }; }
};

// reference src/target.rs:8-27
struct Target {
    bool screen = true; Texture texture;
    static Target Screen() { return Target{}; }
    static Target TextureTarget(Texture t) { Target x; x.screen = false; x.texture = std::move(t); return x; }
    Format format() const { return screen ? Format::BgraU8 : texture.inner->format; }
};

// ---- Program (reference src/program.rs:23-48): two gfx950 code objects plus the ordered binding lists -------------------
class Program {
public:
    struct Inner {
        rh_module* vertex_shader = nullptr; rh_module* fragment_shader = nullptr;
        std::vector<Attribute> attributes; std::vector<Instanced> instances;
        std::vector<std::pair<Uniform, Visibility>> uniforms; std::vector<std::pair<Texture, Visibility>> textures;
        ~Inner() { if (vertex_shader) rh_module_release(vertex_shader); if (fragment_shader) rh_module_release(fragment_shader); }
    };
    std::shared_ptr<Inner> inner;

    Program() = default;
    Program(rh_device* device, const std::vector<uint8_t>& vert, const std::vector<uint8_t>& frag, std::vector<Attribute> attributes,
            std::vector<Instanced> instances, std::vector<std::pair<Uniform, Visibility>> uniforms,
            std::vector<std::pair<Texture, Visibility>> textures) : inner(std::make_shared<Inner>()) {
        check(rh_module_load(device, vert.data(), vert.size(), &inner->vertex_shader));
        check(rh_module_load(device, frag.data(), frag.size(), &inner->fragment_shader));
        inner->attributes = std::move(attributes); inner->instances = std::move(instances);
        inner->uniforms = std::move(uniforms); inner->textures = std::move(textures);
    }
    // generations of every bound resource, in binding order (src/program.rs:33-40)
    std::vector<uint32_t> latest_generations() const {
        std::vector<uint32_t> g;
        for (auto& a : inner->attributes) g.push_back(a.buffer.inner->generation);
        for (auto& i : inner->instances) g.push_back(i.buffer.inner->generation);
        for (auto& u : inner->uniforms) g.push_back(u.first.buffer.inner->generation);
        for (auto& t : inner->textures) g.push_back(t.first.inner->generation);
        return g;
    }
};

// ---- Pipeline (reference src/pipeline.rs:21-237) --------------------------------------------------------------------------
enum class RecordingPosition { None, NotLast, Last };       // only the Last pipeline of a frame copies the recording out
constexpr size_t BINDINGS_PER_GROUP = 4;                   // src/pipeline.rs:25-27

class Pipeline {
public:
    struct Inner {
        rh_device* device; Program program; BlendMode blend_mode; Primitive primitive; uint32_t msaa_samples;
        RecordingPosition position_in_recording = RecordingPosition::None; std::vector<Target> targets;
        std::vector<uint32_t> seen_generations; rh_pipeline* handle = nullptr;
        uint32_t window_width = 0, window_height = 0;
        Texture msaa_texture;                                 // Some(..) <=> msaa_samples > 1 (src/pipeline.rs:31)
        ~Inner() { if (handle) rh_pipeline_release(handle); }

        // One multisampled texture per pipeline, shaped after the FIRST target (src/pipeline.rs:164-176); every pass of the
        // pipeline renders into it (clear and load act on it) and resolves into the target.
        void create_msaa_texture() {
            if (msaa_samples <= 1) { msaa_texture = Texture(); return; }
            const Target& t = targets.at(0);
            const uint32_t w = t.screen ? window_width : t.texture.inner->width, h = t.screen ? window_height : t.texture.inner->height;
            msaa_texture = Texture(device, w, h, 1, FilterMode::Nearest, t.format(), msaa_samples, true, false, false);
        }
        void resize_msaa_texture(uint32_t ww, uint32_t wh, const std::vector<Target>& to) {        // src/pipeline.rs:178-187
            if (!msaa_texture.inner || to.empty()) return;
            const Target& t = to[0];
            msaa_texture.resize(t.screen ? ww : t.texture.inner->width, t.screen ? wh : t.texture.inner->height, 1);
        }

        // flat slot list: instanced..., uniforms..., then per texture: texture [, sampler]; slot i = (set i / 4, binding i % 4)
        std::vector<rh_slot_desc> create_bind_groups() const {
            std::vector<rh_slot_desc> slots;
            for (auto& i : program.inner->instances) slots.push_back(i.binding());
            for (auto& u : program.inner->uniforms) slots.push_back(u.first.binding(u.second));
            for (auto& t : program.inner->textures) {
                slots.push_back(t.first.texture_binding(t.second));
                if (t.first.inner->sampler) slots.push_back(t.first.sampler_binding(t.second));
            }
            return slots;
        }
        // one state per target, + 1 RGBA8 state while recording (src/pipeline.rs:133-142)
        std::vector<rh_color_target_desc> create_color_target_states() const {
            std::vector<rh_color_target_desc> states;
            for (auto& t : targets) states.push_back(blend_mode.state(t.format()));
            if (position_in_recording != RecordingPosition::None) states.push_back(blend_mode.state(Format::RgbaU8));
            return states;
        }
        void rebuild() {
            auto slots = create_bind_groups();
            auto states = create_color_target_states();
            std::vector<rh_attribute_desc> attrs;
            for (auto& a : program.inner->attributes) attrs.push_back({ a.location, a.size });
            rh_pipeline_desc d{};
            d.vertex_module = program.inner->vertex_shader; d.fragment_module = program.inner->fragment_shader;
            d.n_attributes = (uint32_t)attrs.size(); d.attributes = attrs.data();
            d.n_slots = (uint32_t)slots.size(); d.slots = slots.data();
            d.topology = (uint32_t)primitive; d.samples = msaa_samples;
            d.n_targets = (uint32_t)states.size(); d.targets = states.data();
            rh_pipeline* fresh = nullptr;
            check(rh_pipeline_create(device, &d, &fresh));
            if (handle) rh_pipeline_release(handle);
            handle = fresh;
        }
    };
    std::shared_ptr<Inner> inner;

    Pipeline() = default;
    Pipeline(rh_device* device, uint32_t window_width, uint32_t window_height, Program program, BlendMode blend_mode, Primitive primitive,
             uint32_t msaa_samples, std::vector<Target> targets) : inner(std::make_shared<Inner>()) {
        inner->device = device; inner->program = std::move(program); inner->blend_mode = blend_mode; inner->primitive = primitive;
        inner->msaa_samples = msaa_samples; inner->targets = std::move(targets);
        inner->window_width = window_width; inner->window_height = window_height;
        inner->create_msaa_texture();
        inner->rebuild();
        inner->seen_generations = inner->program.latest_generations();
    }
    void recreate_on_buffer_or_texture_resize(uint32_t window_width, uint32_t window_height, const std::vector<Target>& targets) const {   // src/pipeline.rs:44-62
        inner->resize_msaa_texture(window_width, window_height, targets);
        auto actual = inner->program.latest_generations();
        if (actual == inner->seen_generations) return;
        inner->rebuild();
        inner->window_width = window_width; inner->window_height = window_height;
        inner->seen_generations = std::move(actual);
    }
    void set_msaa_samples(uint32_t n) const { inner->msaa_samples = n; inner->create_msaa_texture(); inner->rebuild(); }   // src/pipeline.rs:64-76
    void set_stream_position(RecordingPosition p) const { inner->position_in_recording = p; inner->rebuild(); }   // src/pipeline.rs:78-88
};

// ---- captured frames (reference src/video_frame.rs:6-104) ----------------------------------------------------------------
enum class FrameStatus : uint32_t { Captured = 0, Dropped = 1, Missing = 2 };

struct VideoFrame {
    FrameStatus status = FrameStatus::Missing;
    rh_buffer* image_data = nullptr;          // Some(ImageData::Buffer) when not null; one reference owned
    uint32_t width = 0, height = 0; Format format = Format::RgbaU8;
    uint32_t unpadded_bytes_per_row = 0, padded_bytes_per_row = 0;
    size_t frame_number = 0, frame_size_in_bytes = 0;
    std::shared_ptr<size_t> buffer_size_in_bytes;      // the in-flight capture byte budget shared with the recorder

    VideoFrame() = default;
    VideoFrame(const VideoFrame&) = delete;
    VideoFrame& operator=(const VideoFrame&) = delete;
    VideoFrame(VideoFrame&& o) noexcept { *this = std::move(o); }
    VideoFrame& operator=(VideoFrame&& o) noexcept {
        release();
        status = o.status; image_data = o.image_data; o.image_data = nullptr; width = o.width; height = o.height; format = o.format;
        unpadded_bytes_per_row = o.unpadded_bytes_per_row; padded_bytes_per_row = o.padded_bytes_per_row;
        frame_number = o.frame_number; frame_size_in_bytes = o.frame_size_in_bytes; buffer_size_in_bytes = std::move(o.buffer_size_in_bytes);
        return *this;
    }
    ~VideoFrame() { release(); }
    // Drop returns the frame's bytes to the budget (src/video_frame.rs:70-77)
    void release() {
        if (image_data) { if (buffer_size_in_bytes) *buffer_size_in_bytes -= frame_size_in_bytes; rh_buffer_release(image_data); image_data = nullptr; }
    }
    // the mapped pixels (pinned host memory), valid while the frame lives
    std::pair<const uint8_t*, size_t> bytes() const {
        if (!image_data) throw Panic("The buffer is no longer available.");
        const uint8_t* p = nullptr; size_t n = 0;
        check(rh_buffer_mapped_ptr(image_data, &p, &n));
        return { p, n };
    }
    // bincode 2 `config::standard()`: varint little endian, fields in declaration order; ImageData encodes to zero bytes
    static void varint(std::vector<uint8_t>& out, uint64_t v) {
        if (v < 251) { out.push_back((uint8_t)v); return; }
        const int width = v < (1ull << 16) ? 2 : (v < (1ull << 32) ? 4 : 8);
        out.push_back(width == 2 ? 251 : (width == 4 ? 252 : 253));
        for (int i = 0; i < width; i++) out.push_back((uint8_t)(v >> (8 * i)));
    }
    std::vector<uint8_t> encode() const {
        std::vector<uint8_t> out;
        varint(out, (uint64_t)status);
        out.push_back(image_data ? 1 : 0);
        varint(out, width); varint(out, height); varint(out, (uint64_t)format);
        varint(out, unpadded_bytes_per_row); varint(out, padded_bytes_per_row);
        varint(out, frame_number); varint(out, frame_size_in_bytes);
        varint(out, buffer_size_in_bytes ? *buffer_size_in_bytes : 0);
        return out;
    }
};

class Renderer;

// ---- VideoRecorder (reference src/video_recorder.rs:25-188) ----------------------------------------------------------------
class VideoRecorder {
public:
    using ProcessFunction = std::function<void(VideoFrame)>;
    rh_device* device; Texture recording_texture; bool has_clear; ClearColor clear_color; bool cleared_this_frame = false;
    size_t max_buffer_size_in_bytes; ProcessFunction process_function; bool keep_on_device;
    std::shared_ptr<size_t> buffer_size_in_bytes = std::make_shared<size_t>(0);
    std::deque<VideoFrame> video_frames;
    std::deque<std::shared_ptr<int>> frame_states;        // 0 = dropped, 1 = mapping, 2 = mapped, 3 = failed to map
    size_t frame_number = 0;

    VideoRecorder(rh_device* device, uint32_t width, uint32_t height, const ClearColor* clear, size_t max_bytes, ProcessFunction f, bool keep_on_device)
        : device(device), recording_texture(device, width, height, 1, FilterMode::Nearest, Format::RgbaU8, 1, true, true, false),
          has_clear(clear != nullptr), clear_color(clear ? *clear : ClearColor{}), max_buffer_size_in_bytes(max_bytes),
          process_function(std::move(f)), keep_on_device(keep_on_device) {}

    rh_attachment_desc color_attachment() {                // cleared on first use per frame only (src/video_recorder.rs:43-60)
        rh_attachment_desc a{};
        a.texture = recording_texture.inner->handle;
        if (cleared_this_frame || !has_clear) a.load_op = RH_LOAD_OP_LOAD;
        else { cleared_this_frame = true; a.load_op = RH_LOAD_OP_CLEAR; memcpy(a.clear, clear_color.inner, sizeof a.clear); }
        return a;
    }
    void finish_frame() { cleared_this_frame = false; }

    void create_buffer_if_within_memory_limit(const Viewport* viewport) {      // src/video_recorder.rs:66-105
        const Texture::Inner& tex = *recording_texture.inner;
        const uint32_t width = viewport ? (uint32_t)std::floor(viewport->width) : tex.width;
        const uint32_t height = viewport ? (uint32_t)std::floor(viewport->height) : tex.height;
        const uint32_t unpadded = width * bytes_per_texel(tex.format);
        const uint32_t alignment = RH_COPY_BYTES_PER_ROW_ALIGNMENT;
        const uint32_t padded = unpadded + (alignment - unpadded % alignment) % alignment;
        const size_t frame_size = (size_t)padded * height;
        const size_t prev_size = *buffer_size_in_bytes; *buffer_size_in_bytes += frame_size;
        const bool drop_frame = prev_size > max_buffer_size_in_bytes;
        rh_buffer* buffer = nullptr;
        if (drop_frame) *buffer_size_in_bytes -= frame_size;
        else check(rh_buffer_create(device, frame_size, RH_BUFFER_COPY_DST | RH_BUFFER_MAP_READ, nullptr, 0, &buffer));
        frame_number += 1;                                 // incremented whether or not the frame is dropped
        VideoFrame f;
        f.status = drop_frame ? FrameStatus::Dropped : FrameStatus::Captured; f.image_data = buffer;
        f.width = width; f.height = height; f.format = tex.format; f.unpadded_bytes_per_row = unpadded; f.padded_bytes_per_row = padded;
        f.frame_number = frame_number; f.frame_size_in_bytes = frame_size;
        if (buffer) f.buffer_size_in_bytes = buffer_size_in_bytes;
        video_frames.push_back(std::move(f));
    }
    void copy_texture_to_buffer_if_present(rh_cmd* encoder, const Viewport* viewport) {                 // src/video_recorder.rs:107-128
        VideoFrame& f = video_frames.back();
        if (!f.image_data) return;
        const uint32_t margin_x = viewport ? (uint32_t)std::ceil(viewport->margin_x) : 0, margin_y = viewport ? (uint32_t)std::ceil(viewport->margin_y) : 0;
        const Texture::Inner& tex = *recording_texture.inner;
        // u32 arithmetic in the reference (src/video_recorder.rs:124-125): a viewport thinner than one pixel panics
        if (2 * margin_x > tex.width || 2 * margin_y > tex.height) throw Panic("attempt to subtract with overflow");
        check(rh_cmd_copy_texture_to_buffer(encoder, tex.handle, margin_x, margin_y, f.image_data, f.padded_bytes_per_row, f.height,
                                            tex.width - 2 * margin_x, tex.height - 2 * margin_y));
    }
    static void on_mapped(void* user, int ok) { *static_cast<int*>(user) = ok ? 2 : 3; }
    void initiate_buffer_mapping() {                       // src/video_recorder.rs:130-150
        for (size_t i = frame_states.size(); i < video_frames.size(); i++) {
            VideoFrame& f = video_frames[i];
            auto state = std::make_shared<int>(0);
            if (f.image_data && keep_on_device) *state = 2;           // the consumer reads device memory
            else if (f.image_data) { *state = 1; check(rh_buffer_map_async(f.image_data, &VideoRecorder::on_mapped, state.get())); }
            frame_states.push_back(state);
        }
    }
    void process_mapped_buffers() {                        // in order; stops at the first frame still mapping (src/video_recorder.rs:152-176)
        check(rh_poll(device, 0));
        while (!video_frames.empty()) {
            const int state = *frame_states.front();
            if (state == 0 || state == 2) {
                VideoFrame f = std::move(video_frames.front());
                video_frames.pop_front(); frame_states.pop_front();
                process_function(std::move(f));
            } else if (state == 1) break;
            else throw Panic("Failed to memory map buffer data for a video frame.");
        }
    }
};

// ---- Renderer facade (reference src/renderer.rs:26-378) + RenderPass (reference src/render_pass.rs:13-82) -------------------
class Renderer {
public:
    rh_device* device = nullptr; uint32_t window_width, window_height; bool frame = true;
    std::vector<rh_cmdbuf*> commands; std::unique_ptr<VideoRecorder> recorder; uint64_t flushes = 0;

    Renderer(uint32_t width, uint32_t height, int hip_ordinal = 0) : window_width(width), window_height(height) {
        check(rh_device_create(hip_ordinal, width, height, &device));
    }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    ~Renderer() {
        for (rh_cmdbuf* c : commands) rh_cmdbuf_release(c);
        recorder.reset();
        if (device) rh_device_destroy(device);
    }

    void resize_swap_chain(uint32_t width, uint32_t height) {                                          // src/renderer.rs:56-66
        if (width == 0 || height == 0) return;
        window_width = width; window_height = height; frame = false;
        check(rh_screen_resize(device, width, height));
    }
    rh_texture* screen_handle() const { rh_texture* t = nullptr; check(rh_screen_texture(device, &t)); return t; }
    std::vector<uint8_t> read_screen() const {              // BGRA bytes, [height][width][4]
        std::vector<uint8_t> out((size_t)window_width * window_height * 4);
        check(rh_texture_read(screen_handle(), 0, out.data(), out.size()));
        return out;
    }

    // count = (instance_count, vertices_per_instance) (src/renderer.rs:72-74)
    void render(const Pipeline& pipeline, const ClearColor* clear_color, const Viewport* viewport, std::pair<uint32_t, uint32_t> count) {
        render_to(pipeline.inner->targets, pipeline, clear_color, viewport, count);
    }
    // nothing executes until flush() (src/renderer.rs:76-90)
    void render_to(const std::vector<Target>& targets, const Pipeline& pipeline, const ClearColor* clear_color, const Viewport* viewport,
                   std::pair<uint32_t, uint32_t> count) {
        for (auto& t : targets) if (t.screen) frame = true;
        commands.push_back(render_pass(targets, pipeline, clear_color, viewport, count));
    }
    void finish_frame() {                                   // src/renderer.rs:102-117
        flush();
        if (recorder) { recorder->initiate_buffer_mapping(); recorder->process_mapped_buffers(); recorder->finish_frame(); }
        frame = false;                                      // present() is a no-op without a display
    }
    void flush() {                                          // queue.submit of every recorded command buffer, in order (src/renderer.rs:119-122)
        std::vector<rh_cmdbuf*> list; list.swap(commands);
        flushes += 1;
        check(rh_submit(device, list.data(), list.size()));
    }
    void synchronize() { check(rh_device_synchronize(device)); }

    // uploads, routed by location / (set, binding) (src/renderer.rs:124-162)
    void set_attribute(const Pipeline& p, uint32_t location, const std::vector<float>& data) {
        for (auto& a : p.inner->program.inner->attributes) if (a.location == location) { a.buffer.set_data(data.data(), data.size(), flushes); return; }
        throw Panic("no attribute at that location");
    }
    void set_instanced(const Pipeline& p, std::pair<size_t, size_t> index_tuple, const std::vector<float>& data) {
        const size_t index = index_tuple.first * BINDINGS_PER_GROUP + index_tuple.second;
        p.inner->program.inner->instances.at(index).buffer.set_data(data.data(), data.size(), flushes);
    }
    void set_uniform(const Pipeline& p, std::pair<size_t, size_t> index_tuple, const std::vector<float>& data) {
        const size_t index = index_tuple.first * BINDINGS_PER_GROUP + index_tuple.second;
        const Program::Inner& prog = *p.inner->program.inner;
        prog.uniforms.at(index - prog.instances.size()).first.buffer.set_data(data.data(), data.size(), flushes);      // src/renderer.rs:356-358
    }
    void set_texture(const Pipeline& p, std::pair<size_t, size_t> index_tuple, const std::vector<const std::vector<uint8_t>*>& layers_data) {
        for (size_t layer = 0; layer < layers_data.size(); layer++)
            set_part_of_texture(p, index_tuple, 0, 0, (uint32_t)layer, 0, 0, *layers_data[layer]);
    }
    void set_part_of_texture(const Pipeline& p, std::pair<size_t, size_t> index_tuple, uint32_t x, uint32_t y, uint32_t layer, uint32_t w, uint32_t h,
                             const std::vector<uint8_t>& data) {
        const size_t index = index_tuple.first * BINDINGS_PER_GROUP + index_tuple.second;
        const Program::Inner& prog = *p.inner->program.inner;
        prog.textures.at(texture_index(index, prog)).first.set_data(x, y, layer, w, h, data.data(), data.size());
    }
    // walks the textures skipping sampler slots (src/renderer.rs:360-378)
    static size_t texture_index(size_t index, const Program::Inner& prog) {
        long remaining = (long)index - (long)prog.instances.size() - (long)prog.uniforms.size();
        for (size_t i = 0; i < prog.textures.size(); i++) {
            if (remaining == 0) return i;
            remaining -= 1;
            if (prog.textures[i].first.inner->sampler) remaining -= 1;
            if (remaining < 0) throw Panic("Tried to get a texture but a sampler is in that slot.");
        }
        throw Panic("Tried to a get a texture but nothing is in that slot.");
    }
    void set_msaa_samples(const Pipeline& p, uint32_t n) { p.set_msaa_samples(n); }

    // recording (src/renderer.rs:178-197); keep_on_device hands frames over unmapped, for the GPU compressor
    void start_recording(const std::vector<const Pipeline*>& pipelines, const ClearColor* clear_color, float max_buffer_size_in_megabytes,
                         VideoRecorder::ProcessFunction process_function, bool keep_on_device = false) {
        const size_t max_bytes = (size_t)(max_buffer_size_in_megabytes * 1024.0f * 1024.0f);
        recorder.reset(new VideoRecorder(device, window_width, window_height, clear_color, max_bytes, std::move(process_function), keep_on_device));
        for (size_t i = 0; i < pipelines.size(); i++)
            pipelines[i]->set_stream_position(i + 1 == pipelines.size() ? RecordingPosition::Last : RecordingPosition::NotLast);
    }
    void stop_recording(const std::vector<const Pipeline*>& pipelines) {
        recorder.reset();
        for (auto* p : pipelines) p->set_stream_position(RecordingPosition::None);
    }
    std::string adapter_info() const { char buf[256]; check(rh_device_info(device, buf, sizeof buf)); return buf; }

    // factories (src/renderer.rs:203-298)
    Pipeline pipeline(Program program, BlendMode blend_mode, Primitive primitive, uint32_t msaa_samples, std::vector<Target> targets) {
        return Pipeline(device, window_width, window_height, std::move(program), blend_mode, primitive, msaa_samples, std::move(targets));
    }
    Attribute attribute(uint32_t location, uint32_t size) { return Attribute(device, location, size); }
    Instanced instanced() { return Instanced(device); }
    Uniform uniform() { return Uniform(device); }
    Texture texture(uint32_t width, uint32_t height, uint32_t layers, FilterMode filter_mode, Format format, bool renderable, bool copyable, bool with_sampler) {
        return Texture(device, width, height, layers, filter_mode, format, 1, renderable, copyable, with_sampler);
    }
    Program program(const std::vector<uint8_t>& vert, const std::vector<uint8_t>& frag, std::vector<Attribute> attributes, std::vector<Instanced> instances,
                    std::vector<std::pair<Uniform, Visibility>> uniforms, std::vector<std::pair<Texture, Visibility>> textures) {
        return Program(device, vert, frag, std::move(attributes), std::move(instances), std::move(uniforms), std::move(textures));
    }
    Viewport viewport(float aspect_x, float aspect_y) const { return Viewport(aspect_x, aspect_y, (float)window_width, (float)window_height); }
    static Target screen_target() { return Target::Screen(); }
    static Target texture_target(Texture t) { return Target::TextureTarget(std::move(t)); }
    static Format bgra_u8() { return Format::BgraU8; }
    static Format red_u8() { return Format::RU8; }
    static Format rgba_u8() { return Format::RgbaU8; }
    static Format rgba_f16() { return Format::RgbaF16; }
    static Format rgba_f32() { return Format::RgbaF32; }
    static FilterMode linear_filtering() { return FilterMode::Linear; }
    static FilterMode nearest_filtering() { return FilterMode::Nearest; }
    static Visibility visible_to_vertex_shader() { return Visibility::VertexShader; }
    static Visibility visible_to_fragment_shader() { return Visibility::FragmentShader; }
    static Visibility visible_to_both_shaders() { return Visibility::BothShaders; }
    static BlendMode additive_blend() { return BlendMode::additive(); }
    static BlendMode pre_multiplied_blend() { return BlendMode::pre_multiplied_alpha(); }
    static Primitive triangle_primitive() { return Primitive::Triangle; }
    static Primitive triangle_strip_primitive() { return Primitive::TriangleStrip; }
    static ClearColor clear_color(double r, double g, double b, double a) { return ClearColor{ { r, g, b, a } }; }

private:
    // encodes exactly one render pass with one instanced draw; after the Last recorded pipeline appends the capture copy
    rh_cmdbuf* render_pass(const std::vector<Target>& targets, const Pipeline& pipeline, const ClearColor* clear, const Viewport* viewport,
                           std::pair<uint32_t, uint32_t> count) {
        Pipeline::Inner& pl = *pipeline.inner;
        pipeline.recreate_on_buffer_or_texture_resize(window_width, window_height, targets);
        if (recorder) recorder->recording_texture.resize(window_width, window_height, 1);
        rh_cmd* encoder = nullptr;
        check(rh_cmd_begin(device, &encoder));
        if (!targets.empty()) {                             // src/render_pass.rs:26
            std::vector<rh_attachment_desc> attachments;
            for (auto& t : targets) {
                rh_attachment_desc a{};
                rh_texture* view = t.screen ? screen_handle() : t.texture.inner->handle;
                // src/render_pass.rs:76-79: render into the pipeline's multisampled texture, resolve into the target
                if (pl.msaa_samples != 1) { a.texture = pl.msaa_texture.inner->handle; a.resolve_target = view; }
                else a.texture = view;
                if (clear) { a.load_op = RH_LOAD_OP_CLEAR; memcpy(a.clear, clear->inner, sizeof a.clear); } else a.load_op = RH_LOAD_OP_LOAD;
                attachments.push_back(a);
            }
            if (pl.position_in_recording != RecordingPosition::None) attachments.push_back(recorder->color_attachment());
            std::vector<rh_buffer*> vbs;
            for (auto& a : pl.program.inner->attributes) vbs.push_back(a.buffer.inner->handle);       // slot = list index (src/render_pass.rs:35-37)
            rh_pass_desc d{};
            d.pipeline = pl.handle;
            d.n_attachments = (uint32_t)attachments.size(); d.attachments = attachments.data();
            d.n_vertex_buffers = (uint32_t)vbs.size(); d.vertex_buffers = vbs.data();
            if (viewport) {                                 // src/render_pass.rs:39-41
                d.has_viewport = 1;
                const float v[6] = { viewport->margin_x, viewport->margin_y, viewport->width, viewport->height, 0.0f, 1.0f };
                memcpy(d.viewport, v, sizeof v);
            }
            d.instance_count = count.first; d.vertex_count = count.second;
            check(rh_cmd_render_pass(encoder, &d));
            if (pl.position_in_recording == RecordingPosition::Last) {
                recorder->create_buffer_if_within_memory_limit(viewport);
                recorder->copy_texture_to_buffer_if_present(encoder, viewport);
            }
        }
        rh_cmdbuf* cbuffer = nullptr;
        check(rh_cmd_finish(encoder, &cbuffer));
        return cbuffer;
    }
};

// ---- Compressor (reference src/compressor.rs:15-148): N LZ4F stream files `<timestamp>--<i>.sz`, compressed on the GPU ------
class Compressor {
public:
    rh_compressor* handle = nullptr; uint32_t n_streams; std::vector<FILE*> files; std::vector<VideoFrame> queued;
    size_t frames_captured = 0, frames_dropped = 0, raw_video_size = 0, compressed_size = 0;

    Compressor(Renderer& renderer, const std::string& directory, const std::string& timestamp, int lz4_compression_level = 0, uint32_t n_streams = 64)
        : n_streams(n_streams) {
        if (lz4_compression_level > 12) throw Panic("Please choose a compression level in the range 0..=12");
        check(rh_compressor_create(renderer.device, n_streams, lz4_compression_level, &handle));
        for (uint32_t i = 0; i < n_streams; i++) {
            const std::string path = directory + "/" + timestamp + "--" + std::to_string(i) + ".sz";
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) throw Panic("cannot create " + path);
            files.push_back(f);
        }
    }
    Compressor(const Compressor&) = delete;
    ~Compressor() { try { finish(); } catch (...) {} }

    // frame f goes to stream (f - 1) % n_streams; packet = [packet_len u64 BE][metadata_len u64 BE][bincode][image] (src/compressor.rs:101-131)
    void compress_to_disk(VideoFrame frame) {
        const uint32_t stream = (uint32_t)((frame.frame_number - 1) % n_streams);
        const std::vector<uint8_t> meta = frame.encode();
        check(rh_compressor_write_packet(handle, stream, meta.data(), meta.size(), frame.image_data));
        if (frame.image_data) { frames_captured++; raw_video_size += frame.frame_size_in_bytes; } else frames_dropped++;
        queued.push_back(std::move(frame));
        if (queued.size() >= n_streams) submit();
    }
    void submit() { check(rh_compressor_submit(handle)); queued.clear(); drain(); }
    void flush() { check(rh_compressor_flush(handle)); queued.clear(); drain(); }
    void finish() {                                         // end marks, close (src/compressor.rs:46-62)
        if (!handle) return;
        flush();
        for (uint32_t i = 0; i < n_streams; i++) check(rh_compressor_finish_stream(handle, i));
        drain();
        for (FILE* f : files) fclose(f);
        files.clear();
        rh_compressor_destroy(handle); handle = nullptr;
    }
private:
    void drain() {                                          // straight from the library's pinned memory to the files
        for (uint32_t i = 0; i < n_streams; i++)
            for (;;) {
                const uint8_t* p = nullptr; size_t n = 0;
                check(rh_compressor_peek(handle, i, &p, &n));
                if (!n) break;
                if (fwrite(p, 1, n, files[i]) != n) throw Panic("short write to a stream file");
                check(rh_compressor_consume(handle, i, n));
                compressed_size += n;
            }
    }
};

} // namespace renderer
