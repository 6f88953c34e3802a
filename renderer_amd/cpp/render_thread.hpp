// Renderer proxy that owns a Renderer on a worker thread (reference src/render_thread.rs:4-333): calls are queued as
// messages and executed in order on that thread, objects are handed out as index handles (PipelineRef, TextureRef, ...),
// and only the calls that return something wait for the worker.  The worker thread is the only thread that touches
// the device, which is the threading rule of the C ABI.  A panic on the worker is rethrown on the calling thread by the
// next call that waits (synchronize, a factory, adapter_info, join).
#pragma once
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "renderer.hpp"

namespace renderer {

struct PipelineRef { size_t index; };
struct AttributeRef { size_t index; };
struct InstancedRef { size_t index; };
struct UniformRef { size_t index; };
struct TextureRef { size_t index; };
struct ProgramRef { size_t index; };
// TargetRef::Screen | TargetRef::TextureRef(texture) (src/render_thread.rs:56)
struct TargetRef {
    bool screen = true; TextureRef texture{ 0 };
    Target to_target(const std::vector<Texture>& textures) const { return screen ? Target::Screen() : Target::TextureTarget(textures.at(texture.index)); }
};

class RenderThread {
    struct Tables {
        std::vector<Pipeline> pipelines; std::vector<Attribute> attributes; std::vector<Instanced> instances;
        std::vector<Uniform> uniforms; std::vector<Texture> textures; std::vector<Program> programs;
    };
    using Call = std::function<void(Renderer&, Tables&)>;

    uint32_t window_width, window_height;
    std::mutex mutex;
    std::condition_variable wake;
    std::deque<Call> calls;                 // unbounded, like the reference's function-call channel
    bool closing = false;
    std::exception_ptr error;               // the worker's panic, if any
    std::thread worker;

    void run(int hip_ordinal) {
        std::unique_ptr<Renderer> renderer;
        Tables tables;
        try { renderer.reset(new Renderer(window_width, window_height, hip_ordinal)); }
        catch (...) { std::lock_guard<std::mutex> lock(mutex); error = std::current_exception(); }
        for (;;) {
            Call call;
            {   std::unique_lock<std::mutex> lock(mutex);
                wake.wait(lock, [&] { return closing || !calls.empty(); });
                if (calls.empty()) break;
                call = std::move(calls.front()); calls.pop_front();
            }
            if (!renderer) { call = nullptr; continue; }            // creation failed: waiting callers see the error
            try { call(*renderer, tables); }
            catch (...) { std::lock_guard<std::mutex> lock(mutex); if (!error) error = std::current_exception(); }
        }
        tables = Tables();                  // objects go before the device
        renderer.reset();
    }

    // fire and forget (src/render_thread.rs: the calls that send a FunctionCall and return)
    void send(Call call) {
        {   std::lock_guard<std::mutex> lock(mutex); calls.push_back(std::move(call)); }
        wake.notify_one();
    }
    // calls that return something: wait for the worker (the reference's bounded(1) return channel)
    template <typename T>
    T ask(std::function<T(Renderer&, Tables&)> f) {
        auto promise = std::make_shared<std::promise<T>>();
        std::future<T> result = promise->get_future();
        auto fulfilled = std::make_shared<bool>(false);
        // answers even if the worker drops the call without running it (renderer creation failed): the deleter runs when
        // the last copy of the queued call is destroyed
        std::shared_ptr<void> guard(nullptr, [promise, fulfilled, this](void*) {
            if (*fulfilled) return;
            std::lock_guard<std::mutex> lock(mutex);
            promise->set_exception(error ? error : std::make_exception_ptr(Panic("the render thread is gone")));
        });
        send([promise, fulfilled, guard, f](Renderer& r, Tables& t) {
            try { T value = f(r, t); *fulfilled = true; promise->set_value(std::move(value)); }
            catch (...) { if (!*fulfilled) { *fulfilled = true; promise->set_exception(std::current_exception()); } throw; }
        });
        guard.reset();
        return result.get();
    }

public:
    RenderThread(uint32_t width, uint32_t height, int hip_ordinal = 0) : window_width(width), window_height(height) {
        worker = std::thread([this, hip_ordinal] { run(hip_ordinal); });
    }
    RenderThread(const RenderThread&) = delete;
    RenderThread& operator=(const RenderThread&) = delete;
    ~RenderThread() { try { join(); } catch (...) {} }

    // src/render_thread.rs:173-184
    void join() {
        if (!worker.joinable()) return;
        {   std::lock_guard<std::mutex> lock(mutex); closing = true; }
        wake.notify_one();
        worker.join();
        if (error) { std::exception_ptr e = error; error = nullptr; std::rethrow_exception(e); }
    }
    void synchronize() { ask<bool>([](Renderer&, Tables&) { return true; }); }

    std::pair<uint32_t, uint32_t> window_size() const { return { window_width, window_height }; }
    void resize_swap_chain(uint32_t width, uint32_t height) {
        window_width = width; window_height = height;
        send([=](Renderer& r, Tables&) { r.resize_swap_chain(width, height); });
    }
    void resize_texture(TextureRef texture, uint32_t width, uint32_t height, uint32_t layers) {
        send([=](Renderer&, Tables& t) { t.textures.at(texture.index).resize(width, height, layers); });
    }
    // Option<ClearColor> / Option<Viewport> travel by value (src/render_thread.rs:201-209)
    void render(PipelineRef pipeline, const ClearColor* clear_color, const Viewport* viewport, std::pair<uint32_t, uint32_t> count) {
        const bool has_clear = clear_color != nullptr, has_viewport = viewport != nullptr;
        const ClearColor clear = has_clear ? *clear_color : ClearColor{};
        const Viewport view = has_viewport ? *viewport : Viewport(1.f, 1.f, 1.f, 1.f);
        send([=](Renderer& r, Tables& t) { r.render(t.pipelines.at(pipeline.index), has_clear ? &clear : nullptr, has_viewport ? &view : nullptr, count); });
    }
    void render_to(std::vector<TargetRef> targets, PipelineRef pipeline, const ClearColor* clear_color, const Viewport* viewport,
                   std::pair<uint32_t, uint32_t> count) {
        const bool has_clear = clear_color != nullptr, has_viewport = viewport != nullptr;
        const ClearColor clear = has_clear ? *clear_color : ClearColor{};
        const Viewport view = has_viewport ? *viewport : Viewport(1.f, 1.f, 1.f, 1.f);
        send([=](Renderer& r, Tables& t) {
            std::vector<Target> to;
            for (const TargetRef& x : targets) to.push_back(x.to_target(t.textures));
            r.render_to(to, t.pipelines.at(pipeline.index), has_clear ? &clear : nullptr, has_viewport ? &view : nullptr, count);
        });
    }
    void finish_frame() { send([](Renderer& r, Tables&) { r.finish_frame(); }); }
    void flush() { send([](Renderer& r, Tables&) { r.flush(); }); }

    void set_attribute(PipelineRef pipeline, uint32_t location, std::vector<float> data) {
        send([=](Renderer& r, Tables& t) { r.set_attribute(t.pipelines.at(pipeline.index), location, data); });
    }
    void set_instanced(PipelineRef pipeline, std::pair<size_t, size_t> index_tuple, std::vector<float> data) {
        send([=](Renderer& r, Tables& t) { r.set_instanced(t.pipelines.at(pipeline.index), index_tuple, data); });
    }
    void set_uniform(PipelineRef pipeline, std::pair<size_t, size_t> index_tuple, std::vector<float> data) {
        send([=](Renderer& r, Tables& t) { r.set_uniform(t.pipelines.at(pipeline.index), index_tuple, data); });
    }
    void set_texture(PipelineRef pipeline, std::pair<size_t, size_t> index_tuple, std::vector<std::vector<uint8_t>> layers_data) {
        send([=](Renderer& r, Tables& t) {
            std::vector<const std::vector<uint8_t>*> layers;
            for (const auto& l : layers_data) layers.push_back(&l);
            r.set_texture(t.pipelines.at(pipeline.index), index_tuple, layers);
        });
    }
    void set_part_of_texture(PipelineRef pipeline, std::pair<size_t, size_t> index_tuple, uint32_t x, uint32_t y, uint32_t layer, uint32_t w, uint32_t h,
                             std::vector<uint8_t> data) {
        send([=](Renderer& r, Tables& t) { r.set_part_of_texture(t.pipelines.at(pipeline.index), index_tuple, x, y, layer, w, h, data); });
    }
    void set_vsync(bool) {}                                  // no display: nothing to pace (src/render_thread.rs:246-249)
    void set_msaa_samples(PipelineRef pipeline, uint32_t msaa_samples) {
        send([=](Renderer& r, Tables& t) { r.set_msaa_samples(t.pipelines.at(pipeline.index), msaa_samples); });
    }
    // process_function runs on the render thread inside finish_frame (src/render_thread.rs:256-259: FnMut + Send)
    void start_recording(std::vector<PipelineRef> pipelines, const ClearColor* clear_color, float max_buffer_size_in_megabytes,
                         VideoRecorder::ProcessFunction process_function, bool keep_on_device = false) {
        const bool has_clear = clear_color != nullptr;
        const ClearColor clear = has_clear ? *clear_color : ClearColor{};
        send([=](Renderer& r, Tables& t) {
            std::vector<const Pipeline*> ps;
            for (PipelineRef p : pipelines) ps.push_back(&t.pipelines.at(p.index));
            r.start_recording(ps, has_clear ? &clear : nullptr, max_buffer_size_in_megabytes, process_function, keep_on_device);
        });
    }
    void stop_recording(std::vector<PipelineRef> pipelines) {
        send([=](Renderer& r, Tables& t) {
            // frames still in flight are delivered before the recorder goes away
            r.synchronize();
            if (r.recorder) r.recorder->process_mapped_buffers();
            std::vector<const Pipeline*> ps;
            for (PipelineRef p : pipelines) ps.push_back(&t.pipelines.at(p.index));
            r.stop_recording(ps);
        });
    }
    std::string adapter_info() { return ask<std::string>([](Renderer& r, Tables&) { return r.adapter_info(); }); }
    std::vector<uint8_t> read_screen() { return ask<std::vector<uint8_t>>([](Renderer& r, Tables&) { r.synchronize(); return r.read_screen(); }); }

    // factories: the object stays on the render thread, the caller gets its index (src/render_thread.rs:274-320)
    PipelineRef pipeline(ProgramRef program, BlendMode blend_mode, Primitive primitive, uint32_t msaa_samples, std::vector<TargetRef> targets) {
        return ask<PipelineRef>([=](Renderer& r, Tables& t) {
            std::vector<Target> to;
            for (const TargetRef& x : targets) to.push_back(x.to_target(t.textures));
            t.pipelines.push_back(r.pipeline(t.programs.at(program.index), blend_mode, primitive, msaa_samples, to));
            return PipelineRef{ t.pipelines.size() - 1 };
        });
    }
    AttributeRef attribute(uint32_t location, uint32_t size) {
        return ask<AttributeRef>([=](Renderer& r, Tables& t) { t.attributes.push_back(r.attribute(location, size)); return AttributeRef{ t.attributes.size() - 1 }; });
    }
    InstancedRef instanced() {
        return ask<InstancedRef>([](Renderer& r, Tables& t) { t.instances.push_back(r.instanced()); return InstancedRef{ t.instances.size() - 1 }; });
    }
    UniformRef uniform() {
        return ask<UniformRef>([](Renderer& r, Tables& t) { t.uniforms.push_back(r.uniform()); return UniformRef{ t.uniforms.size() - 1 }; });
    }
    TextureRef texture(uint32_t width, uint32_t height, uint32_t layers, FilterMode filter_mode, Format format, bool renderable, bool copyable, bool with_sampler) {
        return ask<TextureRef>([=](Renderer& r, Tables& t) {
            t.textures.push_back(r.texture(width, height, layers, filter_mode, format, renderable, copyable, with_sampler));
            return TextureRef{ t.textures.size() - 1 };
        });
    }
    ProgramRef program(std::vector<uint8_t> vert, std::vector<uint8_t> frag, std::vector<AttributeRef> attributes, std::vector<InstancedRef> instances,
                       std::vector<std::pair<UniformRef, Visibility>> uniforms, std::vector<std::pair<TextureRef, Visibility>> textures) {
        return ask<ProgramRef>([=](Renderer& r, Tables& t) {
            std::vector<Attribute> a; std::vector<Instanced> i;
            std::vector<std::pair<Uniform, Visibility>> u; std::vector<std::pair<Texture, Visibility>> x;
            for (AttributeRef ref : attributes) a.push_back(t.attributes.at(ref.index));
            for (InstancedRef ref : instances) i.push_back(t.instances.at(ref.index));
            for (const auto& ref : uniforms) u.emplace_back(t.uniforms.at(ref.first.index), ref.second);
            for (const auto& ref : textures) x.emplace_back(t.textures.at(ref.first.index), ref.second);
            t.programs.push_back(r.program(vert, frag, a, i, u, x));
            return ProgramRef{ t.programs.size() - 1 };
        });
    }
    Viewport viewport(float aspect_x, float aspect_y) const { return Viewport(aspect_x, aspect_y, (float)window_width, (float)window_height); }
    static TargetRef screen_target() { return TargetRef{}; }
    static TargetRef texture_target(TextureRef texture) { TargetRef t; t.screen = false; t.texture = texture; return t; }
};

} // namespace renderer
