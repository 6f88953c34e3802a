// librenderer_hip.so -- host side of the C ABI declared in include/renderer_hip.h.
//
// Objects are reference counted PODs around HIP allocations.  Command buffers record passes and copies and are
// executed, in order, on the device's single HIP stream by rh_submit (the reference defers all GPU work to
// queue.submit: src/renderer.rs:119-122).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <deque>

#include "../../include/renderer_hip.h"
#include "rh_abi.h"
#include "rh_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_last_error = buf;
    return 1;
}

#define RH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

uint32_t bytes_per_texel(uint32_t format) {
    switch (format) {
    case RH_FORMAT_R_U8: return 1;
    case RH_FORMAT_BGRA_U8: case RH_FORMAT_RGBA_U8: return 4;
    case RH_FORMAT_RGBA_F16: return 8;
    case RH_FORMAT_RGBA_F32: return 16;
    default: return 0;
    }
}

constexpr size_t STATE_HEADER_BYTES = (sizeof(rh_bin_state) + 255) & ~(size_t)255;

// Growable device scratch area.
struct Scratch {
    void* ptr = nullptr; size_t cap = 0;
    bool fixed = false;             // a slice of the library's static device array: never grown, never freed
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (fixed) return fail("the pass needs a scratch area of %zu bytes; the entry point that allocates nothing holds %zu", bytes, cap);
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        size_t want = bytes + bytes / 2 + 4096;
        RH_HIP(hipMalloc(&ptr, want));
        cap = want;
        return 0;
    }
    void release() { if (ptr && !fixed) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    template <typename T> T* as() { return reinterpret_cast<T*>(ptr); }
};

} // namespace

struct rh_device {
    std::atomic<int> refs{1};        // the creator's reference plus one per live object created on the device
    int ordinal = 0;
    hipStream_t stream = nullptr;
    uint32_t width = 0, height = 0;
    rh_texture* screen = nullptr;
    int clip_to_viewport = 1;
    uint32_t strip_row0 = 0, strip_row1 = 0;       // sort-first strip restriction; row1 == 0: none
    int general_kernels = 0;                       // RH_OPTION_GENERAL_KERNELS
    int binning_tickets = 0;                       // RH_OPTION_BINNING_TICKETS
    std::string name;
    // scratch of a render pass: binning arrays and the set-up records of the sorted (tile, cluster) pairs.  `grown` is
    // allocated on demand; `held` is carved out of the library's static device array (rh_submit_static)
    struct PassScratch { Scratch ranges, block_sums, chunk_tmp, keys_a, vals_a, keys_b, vals_b, hist, tile_start, rec, ext; };
    PassScratch grown, grown2, held;
    // Binning and set-up of a pass run on a stream of their own (RH_OPTION_EARLY_BINNING): they only read what the vertex
    // stage reads, so they may run while the tile kernel of the pass before is still at work -- that kernel's last
    // workgroups leave most of the device idle.  The two streams meet again before the tile kernel, which waits for the
    // set-up; consecutive passes alternate between two sets of scratch areas.  A pass takes the binning stream only when
    // nothing but the previous pass's tile kernel has been queued on the main stream since that kernel's start (the
    // binning then waits for everything before that kernel); otherwise it runs on the main stream like everything else.
    hipStream_t bin_stream = nullptr;           // created when the first pass takes it (a process has few hardware queues)
    int compressors = 0;                        // rh_compressor objects alive on this device (each brings two streams of its own)
    int bin_priority = 0;
    int early_binning = 1;           // 0 never, 1 where it pays (passes with room for early_min_pairs pairs or more), 2 wherever it is valid
    uint64_t early_min_pairs = 4000000;
    int eager_count = 1;                        // RH_OPTION_EAGER_COUNT
    int setup_launch = 0;                       // RH_OPTION_SETUP_LAUNCH
    // The tiles of a pass's last, partly filled round of workgroups are cut into bands of rows (rh_tile_args::split2_first):
    // 0 = never, 1 = halves, quarters or eighths (one kind per pass), 2 = where that kind is halves, quarters for as many of them as there is room left.
    int tail_split = 2;
    // tile waves set their issue priority by the work they have left (rh_tile_args::prio_quantum)
    int wave_priority = 1;
    struct Room { hipFunction_t fn; uint32_t groups; };            // per tile kernel: workgroups the whole GPU holds at once
    Room room[32] = {};
    int n_room = 0;
    uint32_t room_override = 0;      // tests: the number of workgroups the GPU is taken to hold (RH_TILE_ROOM)
    int scratch_flip = 0;                       // which of grown / grown2 the next pass takes
    hipEvent_t tile_done[2] = { nullptr, nullptr };   // recorded behind the tile kernel that read the set
    bool tile_done_valid[2] = { false, false };
    hipEvent_t before_tile = nullptr;           // recorded on the main stream just before the last tile kernel
    bool before_tile_valid = false;
    hipEvent_t setup_done = nullptr;
    // `before_tile` has been recorded main_epoch times.  A buffer remembers the count at its last write queued on the main
    // stream (rh_buffer::main_write_epoch): the binning stream waits for the latest record, so it is ordered behind every
    // main-stream write stamped below the current count, and a pass may bin early only if that covers every buffer its
    // vertex stage can read.
    uint64_t main_epoch = 0;
    // rh_buffer_write goes to a block of its own on this stream (see cow_write): an upload never waits for the pass at work.
    hipStream_t copy_stream = nullptr;
    int cow_uploads = 1;                        // (RH_COW_UPLOADS=0: every rh_buffer_write in place on the main stream; for measurements)
    Scratch png_slots, png_rows, png_out;          // GPU PNG encoder scratch
    void* png_host = nullptr; size_t png_host_cap = 0;   // pinned
    unsigned long long* d_total = nullptr;
    unsigned long long* h_total = nullptr;      // pinned: [0] pair count of the last pass, [1] set by a binning kernel that gave up, [2] rh_bin_state::wants_clip
    // A pass is queued whole, sized by `pair_guess`, before its pair count has come back; the count is looked at
    // before the next command runs (resolve_pending) and the pass runs again, waiting for the count, if it did not fit.
    uint32_t compute_units = 0, resident_sort_parts = 0;   // (workgroups of a chained sorting pass the device holds at once)
    uint64_t pair_guess = 0;                    // room given to the pair arrays of a queued pass; 0: no pass seen yet
    bool pending_clip_kernels = false;          // the pass awaiting its count ran the clipping kernels
    hipEvent_t counted = nullptr;               // recorded once the count is in h_total and the pairs are sorted
    const void* pending_pass = nullptr;         // (a PassCmd of the command buffer being submitted, or of `pending_cb`)
    const void* pending_fused = nullptr;        // (the CopyCmd performed by that pass, or null)
    rh_cmdbuf* pending_cb = nullptr;            // the last command buffer of a submit, kept while its last pass awaits its count
    std::vector<rh_cmdbuf*> kept_cbs;           // command buffers whose release was put off by rh_submit_static (releasing may free memory)
    uint32_t pending_capacity = 0;
    // allocation caches (exact-size reuse; capture buffers come and go every frame)
    std::multimap<size_t, void*> free_device, free_pinned;
    size_t free_device_bytes = 0, free_pinned_bytes = 0;
    // Uploads in flight (rh_buffer_write, rh_texture_write): the caller's bytes are copied into pinned memory first, so the
    // copy to the device is queued without the host waiting for the stream (a copy from pageable memory waits for
    // everything queued before it: the host could never prepare a frame while the previous one renders).
    // (`old_dev`: the device block the upload replaced, handed back once the upload AND everything that had been queued on
    // the main stream when it was replaced -- `main_mark` -- have finished)
    struct StagedUpload { void* h; size_t size; hipEvent_t done; void* old_dev; size_t old_size; hipEvent_t main_mark; };
    std::deque<StagedUpload> staged;
    size_t staged_bytes = 0;
    std::vector<hipEvent_t> stage_events;
    // freed device blocks kept for reuse: up to half of the device's memory (a recording run with a thousand streams has
    // two thousand capture buffers of 33 MB coming and going; hipFree waits for every kernel on the device)
    size_t device_cache_max_bytes = (size_t)48 << 30;
    std::vector<rh_buffer*> pending_maps;
    rh_pass_stats stats{};
    // optional per-pass HIP-event timing
    int timing = 0;
    std::vector<hipEvent_t> event_pool;            // unused events
    std::vector<hipEvent_t> timing_events;         // 5 per recorded pass: start, before set-up, after set-up, before tile kernel, end
};

struct rh_buffer {
    std::atomic<int> refs{1};
    rh_device* dev = nullptr;
    size_t size = 0; uint32_t usage = 0;
    void* dptr = nullptr;
    uint64_t main_write_epoch = 0;  // rh_device::main_epoch when a write to the buffer was last queued on the main stream
    bool external = false;          // the device pointer has been handed out: writes the library cannot see
    hipEvent_t upload_done = nullptr;   // behind the last upload on the copy stream
    bool upload_pending = false;
    // mapping state
    void* hptr = nullptr;          // pinned host copy
    hipEvent_t map_event = nullptr;
    int map_state = 0;             // 0 unmapped, 1 pending, 2 mapped
    rh_map_callback cb = nullptr; void* cb_user = nullptr;
};

struct rh_texture {
    std::atomic<int> refs{1};
    rh_device* dev = nullptr;
    uint32_t width = 0, height = 0, layers = 0, format = 0, samples = 1, usage = 0;
    uint32_t bpt = 0;
    void* dptr = nullptr;          // layer-major, tightly packed rows
    size_t bytes = 0;
    // phase-regrouped copies for minified sampling (rh_slot_dev::poly): made when a pass first samples the texture with
    // the specialised bilinear body, and again whenever the texels may have changed since
    void* poly = nullptr;
    uint32_t poly_item_bytes = 0;  // the form of the copies `poly` holds (rh_module_info_t::poly_item_bytes); 0: none yet
    bool poly_valid = false;
    bool poly_always = false;      // the texels can change behind the library's back (render target, device pointer handed out)
};

struct rh_sampler { rh_device* dev; uint32_t filter; };

struct rh_module {
    std::atomic<int> refs{1};
    rh_device* dev = nullptr;
    hipModule_t module = nullptr;
    rh_module_info_t info{};
    hipFunction_t bin_count = nullptr, bin_pairs = nullptr;
    hipFunction_t bin_count_clip = nullptr, bin_pairs_clip = nullptr;   // the kernels that clip (see rh_pipeline::clips)
    hipFunction_t setup_clip[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t resident_pair_parts = 0;           // workgroups of rh_bin_pairs the device holds at once
    hipFunction_t setup[4] = {nullptr, nullptr, nullptr, nullptr};   // 1, 2, 4, 8 samples
    hipFunction_t tile[RH_MAX_COLOR_TARGETS] = {nullptr, nullptr, nullptr, nullptr};
    hipFunction_t tile_ms[3] = {nullptr, nullptr, nullptr};        // 2, 4, 8 samples
    hipFunction_t tile_ms_wide[3] = {nullptr, nullptr, nullptr};   // the same for RGBA16F / RGBA32F targets
    hipFunction_t tile_wide[RH_MAX_COLOR_TARGETS] = {nullptr, nullptr, nullptr, nullptr};   // passes with a float attachment
};

struct SlotRef { uint32_t kind = 0, visibility = 0; rh_buffer* buf = nullptr; rh_texture* tex = nullptr; uint32_t filter = 0; };

struct rh_pipeline {
    std::atomic<int> refs{1};
    rh_device* dev = nullptr;
    rh_module* vs = nullptr; rh_module* fs = nullptr;
    std::vector<rh_attribute_desc> attributes;
    std::vector<SlotRef> slots;
    uint32_t topology = 0, samples = 1;
    std::vector<rh_color_target_desc> targets;
    // Set when a pass of the pipeline met a triangle that crosses a clip plane: from then on its passes run the kernels
    // that clip (their polygons live in private memory, which makes them several times slower than the usual ones).
    bool clips = false;
    uint32_t passes_without_clipping = 0;       // consecutive passes run with the clipping kernels that clipped nothing
};

namespace {

struct PassCmd {
    rh_pipeline* pipeline = nullptr;
    std::vector<rh_attachment_desc> attachments;     // textures retained
    std::vector<rh_buffer*> vertex_buffers;          // retained
    bool has_viewport = false; float viewport[6] = {0, 0, 0, 0, 0, 1};
    uint32_t vertex_count = 0, instance_count = 0;
};

struct CopyCmd {
    rh_texture* tex = nullptr; rh_buffer* buf = nullptr;
    uint32_t ox = 0, oy = 0, bytes_per_row = 0, rows_per_image = 0, width = 0, height = 0;
};

struct Cmd { int kind = 0; PassCmd pass; CopyCmd copy; };   // kind 0 = pass, 1 = copy

} // namespace

struct rh_cmd { rh_device* dev = nullptr; std::vector<Cmd> cmds; };
struct rh_cmdbuf { rh_device* dev = nullptr; std::vector<Cmd> cmds; };

// ---- allocation caches ------------------------------------------------------------------------------------------------
namespace {

// Diagnostic: RH_DEBUG_WAITS=1 prints every wait of the host for the device (what it waits for, and when it is over).
bool trace_waits() { static const bool on = getenv("RH_DEBUG_WAITS") != nullptr; return on; }
#define RH_TRACE(...) do { if (trace_waits()) { fprintf(stderr, "[rh] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

// Diagnostic: RH_DEBUG_SYNC=1 waits for the stream after every step of a pass and says which one (a step that never ends
// is then the last one named).
int dbg_sync(hipStream_t s, const char* what) {
    static const bool on = getenv("RH_DEBUG_SYNC") != nullptr;
    if (!on) return 0;
    fprintf(stderr, "[rh] sync %s ...", what); fflush(stderr);
    RH_HIP(hipStreamSynchronize(s));
    fprintf(stderr, " ok\n"); fflush(stderr);
    return 0;
}

constexpr size_t RH_CACHE_MAX_BLOCKS = 4096;
constexpr size_t RH_PINNED_CACHE_MAX_BYTES = (size_t)8 << 30;

void device_retain(rh_device* dev) { dev->refs.fetch_add(1); }
void device_release(rh_device* dev);

int device_alloc(rh_device* dev, size_t size, void** out) {
    auto it = dev->free_device.find(size);
    if (it != dev->free_device.end()) { *out = it->second; dev->free_device.erase(it); dev->free_device_bytes -= size; return 0; }
    // (64 spare bytes: kernels that read whole dwords or look a few bytes ahead -- the LZ4 match finder -- may touch them)
    RH_TRACE("hipMalloc %zu", size);
    RH_HIP(hipMalloc(out, size + 64));
    RH_TRACE("hipMalloc done");
    return 0;
}
void device_free(rh_device* dev, size_t size, void* p) {
    if (!p) return;
    // Freed blocks wait for the next request of the same size (hipFree waits for the whole device and hipMalloc of a 4K
    // frame takes about a millisecond: a recording run with hundreds of capture buffers in flight would pay both per
    // frame).  The cache is bounded in blocks and in bytes.
    if (dev->free_device.size() < RH_CACHE_MAX_BLOCKS && dev->free_device_bytes + size <= dev->device_cache_max_bytes) {
        dev->free_device.emplace(size, p); dev->free_device_bytes += size;
    } else { RH_TRACE("hipFree %zu", size); (void)hipFree(p); RH_TRACE("hipFree done"); }
}
int pinned_alloc(rh_device* dev, size_t size, void** out) {
    auto it = dev->free_pinned.find(size);
    if (it != dev->free_pinned.end()) { *out = it->second; dev->free_pinned.erase(it); dev->free_pinned_bytes -= size; return 0; }
    RH_HIP(hipHostMalloc(out, size ? size : 1, hipHostMallocDefault));
    return 0;
}
void pinned_free(rh_device* dev, size_t size, void* p) {
    if (!p) return;
    if (dev->free_pinned.size() < RH_CACHE_MAX_BLOCKS && dev->free_pinned_bytes + size <= RH_PINNED_CACHE_MAX_BYTES) {
        dev->free_pinned.emplace(size, p); dev->free_pinned_bytes += size;
    } else (void)hipHostFree(p);
}

// Queues a copy of host bytes to device memory on the device's stream without waiting for the stream (see
// rh_device::staged).  At most RH_STAGED_MAX_BYTES wait in pinned memory: beyond that the oldest upload is waited for.
constexpr size_t RH_STAGED_MAX_BYTES = (size_t)512 << 20;
int pooled_event(rh_device* dev, hipEvent_t* e) {
    if (!dev->stage_events.empty()) { *e = dev->stage_events.back(); dev->stage_events.pop_back(); return 0; }
    RH_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return 0;
}
// `stream`: the device's main stream, or its copy stream with `old_dev` the block the upload replaces (cow_write).
int staged_upload(rh_device* dev, void* dst, const void* src, size_t len, hipStream_t stream, void* old_dev = nullptr, size_t old_size = 0) {
    while (!dev->staged.empty()) {
        rh_device::StagedUpload& u = dev->staged.front();
        const bool over = dev->staged_bytes + len > RH_STAGED_MAX_BYTES;
        if (!over && (hipEventQuery(u.done) != hipSuccess || (u.main_mark && hipEventQuery(u.main_mark) != hipSuccess))) break;
        if (over) {
            RH_TRACE("staged_upload: %zu bytes staged in %zu uploads, waiting for the oldest (done %d, mark %d)", dev->staged_bytes, dev->staged.size(),
                     (int)hipEventQuery(u.done), u.main_mark ? (int)hipEventQuery(u.main_mark) : -1);
            RH_HIP(hipEventSynchronize(u.done)); if (u.main_mark) RH_HIP(hipEventSynchronize(u.main_mark));
            RH_TRACE("staged_upload: oldest done");
        }
        pinned_free(dev, u.size, u.h); dev->stage_events.push_back(u.done); dev->staged_bytes -= u.size;
        if (u.main_mark) dev->stage_events.push_back(u.main_mark);
        if (u.old_dev) device_free(dev, u.old_size, u.old_dev);
        dev->staged.pop_front();
    }
    rh_device::StagedUpload u; u.size = len; u.h = nullptr; u.done = nullptr; u.old_dev = old_dev; u.old_size = old_size; u.main_mark = nullptr;
    if (len > RH_STAGED_MAX_BYTES) {
        // (too large to stage: a copy from the caller's memory, which waits for everything queued on the stream before it)
        RH_HIP(hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, stream));
        if (!old_dev) return 0;
        u.size = 0;
    } else {
        if (pinned_alloc(dev, len, &u.h)) return 1;
        memcpy(u.h, src, len);
        RH_HIP(hipMemcpyAsync(dst, u.h, len, hipMemcpyHostToDevice, stream));
    }
    if (pooled_event(dev, &u.done)) return 1;
    RH_HIP(hipEventRecord(u.done, stream));
    if (old_dev) { if (pooled_event(dev, &u.main_mark)) return 1; RH_HIP(hipEventRecord(u.main_mark, dev->stream)); }
    dev->staged.push_back(u); dev->staged_bytes += u.size;
    return 0;
}

int set_device(rh_device* dev) { RH_HIP(hipSetDevice(dev->ordinal)); return 0; }
int resolve_pending(rh_device* dev, bool blocking = true, bool may_rerun = true);
void release_kept(rh_device* dev);
// What every entry point that queues work on the device's stream, reads results or hands out device memory starts with.
int enter(rh_device* dev) { if (set_device(dev)) return 1; release_kept(dev); return resolve_pending(dev); }

// rh_submit_static: every rh_device of a GPU works in that GPU's one static scratch array, each on a stream of its own,
// and a call returns with its kernels still queued.  One event per GPU, recorded behind the last kernel of every call and
// waited for (on the device) by the next call on that GPU from whichever rh_device, keeps returned calls from
// overlapping on the GPU.  Created with the first rh_device of the GPU, kept until the process ends.
constexpr int RH_MAX_ORDINALS = 64;
std::mutex g_static_mutex;
hipEvent_t g_static_tail[RH_MAX_ORDINALS] = {};

uint32_t unorm8(double c) {
    float o = (float)c;
    o = fminf(fmaxf(o, 0.0f), 1.0f);
    return (uint32_t)rintf(o * 255.0f);
}

uint32_t bits_for(uint32_t n) { uint32_t b = 1; while (b < 32 && (1ull << b) < n) b++; return b; }

// rh_buffer_write.  wgpu orders a write_buffer before the command buffers of the next submit (reference src/buffer.rs:42);
// on one HIP stream that would be behind the pass still at work, whose tile kernel reads the vertex stage's inputs (it
// sets lean quads up itself), so neither the upload nor the binning of the pass that needs it could start before that
// pass has finished.  Instead the buffer moves to a block of its own: the bytes are copied into a fresh block on the
// copy stream (what the write leaves untouched comes over from the old block), passes recorded from now on see the new
// block and wait for the copy, passes already queued keep reading the old one, which is handed back when they and the
// copy have finished.  Falls back to a write in place on the main stream when the old block still has a write queued
// there that the copy stream cannot be ordered behind, or when the block must stay where it is.
int cow_write(rh_buffer* buf, size_t offset, const void* data, size_t len) {
    rh_device* dev = buf->dev;
    const bool whole = offset == 0 && len == buf->size;
    // (not beside a compressor: its passes are not binned early, which is what the move is for, and a process has few
    // hardware queues -- a fifth stream shares one, and with the upload stream behind the compression stream's queue a
    // recording run was a sixth slower)
    static const bool beside_compressor = getenv("RH_EXP_EARLY_WITH_COMPRESSOR") != nullptr;       // (diagnostic: see execute_pass)
    const bool movable = dev->early_binning && dev->cow_uploads && (dev->compressors == 0 || beside_compressor) && !buf->external && buf->map_state == 0 &&
                         !(buf->usage & RH_BUFFER_MAP_READ);
    const bool old_settled = dev->before_tile_valid && buf->main_write_epoch < dev->main_epoch;
    if (!movable || (!whole && !old_settled)) {
        if (buf->upload_pending) { RH_HIP(hipStreamWaitEvent(dev->stream, buf->upload_done, 0)); buf->upload_pending = false; }
        buf->main_write_epoch = dev->main_epoch;
        return staged_upload(dev, (uint8_t*)buf->dptr + offset, data, len, dev->stream);
    }
    if (!dev->copy_stream) RH_HIP(hipStreamCreateWithFlags(&dev->copy_stream, hipStreamNonBlocking));
    if (!buf->upload_done) RH_HIP(hipEventCreateWithFlags(&buf->upload_done, hipEventDisableTiming));
    void* fresh = nullptr;
    if (device_alloc(dev, buf->size, &fresh)) return 1;
    if (!whole) {
        // (earlier uploads to the old block ran on this same stream; its main-stream writes lie before the latest `before_tile`)
        RH_HIP(hipStreamWaitEvent(dev->copy_stream, dev->before_tile, 0));
        if (offset) RH_HIP(hipMemcpyAsync(fresh, buf->dptr, offset, hipMemcpyDeviceToDevice, dev->copy_stream));
        if (offset + len < buf->size)
            RH_HIP(hipMemcpyAsync((uint8_t*)fresh + offset + len, (uint8_t*)buf->dptr + offset + len, buf->size - offset - len, hipMemcpyDeviceToDevice, dev->copy_stream));
    }
    if (staged_upload(dev, (uint8_t*)fresh + offset, data, len, dev->copy_stream, buf->dptr, buf->size)) return 1;
    RH_HIP(hipEventRecord(buf->upload_done, dev->copy_stream));
    buf->upload_pending = true;
    buf->dptr = fresh;
    return 0;
}

} // namespace

extern "C" {

const char* rh_last_error(void) { return g_last_error.c_str(); }
int rh_abi_version(void) { return (int)RH_DEVICE_ABI_VERSION; }

// ---- device -----------------------------------------------------------------------------------------------------------
int rh_device_create(int hip_ordinal, uint32_t width, uint32_t height, rh_device** out) {
    if (!out) return fail("rh_device_create: out is null");
    if (width == 0 || height == 0) return fail("rh_device_create: zero-sized screen");
    int count = 0;
    RH_HIP(hipGetDeviceCount(&count));
    if (hip_ordinal < 0 || hip_ordinal >= count) return fail("rh_device_create: HIP ordinal %d out of range (%d devices)", hip_ordinal, count);
    RH_HIP(hipSetDevice(hip_ordinal));
    hipDeviceProp_t prop;
    RH_HIP(hipGetDeviceProperties(&prop, hip_ordinal));
    rh_device* dev = new rh_device();
    dev->ordinal = hip_ordinal; dev->width = width; dev->height = height;
    dev->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete dev; return fail("hipStreamCreate failed: %s", hipGetErrorString(e)); }
    if (hipMalloc((void**)&dev->d_total, 8) != hipSuccess || hipHostMalloc((void**)&dev->h_total, 32, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&dev->counted, hipEventDisableTiming) != hipSuccess) {
        delete dev; return fail("rh_device_create: allocation failed");
    }
    dev->h_total[0] = 0; dev->h_total[1] = 0; dev->h_total[2] = 0; dev->h_total[3] = 0;
    // (the lowest priority: the binning of the next pass takes the places the tile kernel's last workgroups leave free, not
    // the places of workgroups it has yet to start)
    dev->bin_priority = prio_least;
    if (hipEventCreateWithFlags(&dev->tile_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&dev->tile_done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&dev->before_tile, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&dev->setup_done, hipEventDisableTiming) != hipSuccess) {
        rh_device_destroy(dev); return fail("rh_device_create: creating the binning stream failed");
    }
    if (const char* v = getenv("RH_EARLY_BINNING")) dev->early_binning = std::max(0, std::min(2, atoi(v)));
    if (const char* v = getenv("RH_EARLY_MIN_PAIRS")) dev->early_min_pairs = (uint64_t)std::max(0ll, atoll(v));
    if (const char* v = getenv("RH_COW_UPLOADS")) dev->cow_uploads = atoi(v) ? 1 : 0;
    if (const char* v = getenv("RH_TAIL_SPLIT")) dev->tail_split = std::max(0, std::min(3, atoi(v)));
    if (const char* v = getenv("RH_WAVE_PRIORITY")) dev->wave_priority = atoi(v) ? 1 : 0;
    if (const char* v = getenv("RH_TILE_ROOM")) dev->room_override = (uint32_t)std::max(0, atoi(v));
    dev->compute_units = (uint32_t)prop.multiProcessorCount;
    dev->device_cache_max_bytes = (size_t)prop.totalGlobalMem / 2;
    dev->resident_sort_parts = rhk::sort_pass_resident_parts(dev->compute_units);
    if (rh_texture_create(dev, width, height, 1, RH_FORMAT_BGRA_U8, 1,
                          RH_TEXTURE_RENDER_ATTACHMENT | RH_TEXTURE_COPY_SRC | RH_TEXTURE_BINDING, &dev->screen)) { delete dev; return 1; }
    // rh_submit_static's scratch: this device's copy of the library's static device array, cut into the areas of a pass
    {   void* base = nullptr; size_t bytes = 0;
        hipError_t se = rhk::static_scratch(&base, &bytes);
        if (se != hipSuccess) { rh_device_destroy(dev); return fail("rh_device_create: hipGetSymbolAddress of the static scratch array failed: %s", hipGetErrorString(se)); }
        uint8_t* at = (uint8_t*)base;
        auto carve = [&](Scratch& sc, size_t n_bytes) { sc.ptr = at; sc.cap = n_bytes; sc.fixed = true; at += (n_bytes + 255) & ~(size_t)255; };
        rh_device::PassScratch& h = dev->held;
        carve(h.ranges, (size_t)RH_STATIC_MAX_CLUSTERS * sizeof(rh_tile_range));
        carve(h.block_sums, (size_t)(RH_STATIC_MAX_CLUSTERS / 256 + 1) * 4);
        carve(h.chunk_tmp, 64 * 1024);
        carve(h.keys_a, (size_t)RH_STATIC_MAX_PAIRS * 4); carve(h.vals_a, (size_t)RH_STATIC_MAX_PAIRS * 4);
        carve(h.keys_b, (size_t)RH_STATIC_MAX_PAIRS * 4); carve(h.vals_b, (size_t)RH_STATIC_MAX_PAIRS * 4);
        carve(h.hist, (size_t)(RH_STATIC_MAX_PAIRS / 4096 + 1) * 256 * 4);
        carve(h.tile_start, STATE_HEADER_BYTES + (size_t)RH_STATIC_MAX_TILES * 8);
        carve(h.rec, (size_t)RH_STATIC_RECORD_BYTES); carve(h.ext, (size_t)RH_STATIC_RECORD_BYTES);
        if ((size_t)(at - (uint8_t*)base) > bytes) { rh_device_destroy(dev); return fail("rh_device_create: the static scratch array is smaller than its areas"); }
        if (hip_ordinal < RH_MAX_ORDINALS) {
            std::lock_guard<std::mutex> lock(g_static_mutex);
            if (!g_static_tail[hip_ordinal] && hipEventCreateWithFlags(&g_static_tail[hip_ordinal], hipEventDisableTiming) != hipSuccess) {
                g_static_tail[hip_ordinal] = nullptr;
                rh_device_destroy(dev); return fail("rh_device_create: creating the event of the static scratch array failed");
            }
        }
    }
    *out = dev;
    return 0;
}

// Objects created on a device keep it alive: the device is torn down when the creator has called
// rh_device_destroy AND every buffer/texture/module/pipeline/sampler has been released.
void rh_device_destroy(rh_device* dev) {
    if (!dev) return;
    (void)hipSetDevice(dev->ordinal);
    (void)hipStreamSynchronize(dev->stream);
    dev->pending_pass = nullptr;                // (nobody will look at the results of a pass still awaiting its count)
    if (dev->pending_cb) { rh_cmdbuf* cb = dev->pending_cb; dev->pending_cb = nullptr; rh_cmdbuf_release(cb); }
    { std::vector<rh_cmdbuf*> kept; kept.swap(dev->kept_cbs); for (rh_cmdbuf* cb : kept) rh_cmdbuf_release(cb); }
    std::vector<rh_buffer*> pending; pending.swap(dev->pending_maps);
    for (rh_buffer* b : pending) rh_buffer_release(b);
    rh_texture* screen = dev->screen; dev->screen = nullptr;
    if (screen) rh_texture_release(screen);
    device_release(dev);
}

} // extern "C"

namespace {
void device_release(rh_device* dev) {
    if (dev->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(dev->ordinal);
    (void)hipStreamSynchronize(dev->stream);
    if (dev->bin_stream) (void)hipStreamSynchronize(dev->bin_stream);
    for (rh_device::PassScratch* gp : { &dev->grown, &dev->grown2 }) {
        rh_device::PassScratch& g = *gp;
        Scratch* all[] = { &g.ranges, &g.block_sums, &g.chunk_tmp, &g.keys_a, &g.vals_a, &g.keys_b, &g.vals_b, &g.hist, &g.tile_start, &g.rec, &g.ext };
        for (Scratch* s : all) s->release();
    }
    Scratch* png[] = { &dev->png_slots, &dev->png_rows, &dev->png_out };
    for (Scratch* s : png) s->release();
    for (hipEvent_t e : { dev->tile_done[0], dev->tile_done[1], dev->before_tile, dev->setup_done }) if (e) (void)hipEventDestroy(e);
    if (dev->bin_stream) (void)hipStreamDestroy(dev->bin_stream);
    for (hipEvent_t e : dev->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : dev->timing_events) if (e) (void)hipEventDestroy(e);
    if (dev->copy_stream) (void)hipStreamSynchronize(dev->copy_stream);
    for (rh_device::StagedUpload& u : dev->staged) {
        if (u.h) (void)hipHostFree(u.h);
        (void)hipEventDestroy(u.done);
        if (u.main_mark) (void)hipEventDestroy(u.main_mark);
        if (u.old_dev) (void)hipFree(u.old_dev);
    }
    if (dev->copy_stream) (void)hipStreamDestroy(dev->copy_stream);
    for (hipEvent_t e : dev->stage_events) (void)hipEventDestroy(e);
    for (auto& kv : dev->free_device) (void)hipFree(kv.second);
    for (auto& kv : dev->free_pinned) (void)hipHostFree(kv.second);
    if (dev->counted) (void)hipEventDestroy(dev->counted);
    if (dev->d_total) (void)hipFree(dev->d_total);
    if (dev->h_total) (void)hipHostFree(dev->h_total);
    if (dev->png_host) (void)hipHostFree(dev->png_host);
    (void)hipStreamDestroy(dev->stream);
    delete dev;
}
} // namespace

extern "C" {

int rh_device_info(rh_device* dev, char* name, size_t cap) {
    if (!dev || !name || cap == 0) return fail("rh_device_info: bad argument");
    snprintf(name, cap, "%s", dev->name.c_str());
    return 0;
}

int rh_device_set_option(rh_device* dev, int option, int value) {
    if (!dev) return fail("rh_device_set_option: null device");
    if (enter(dev)) return 1;                   // (a pass that has to run again runs under the options it was submitted under)
    if (option == RH_OPTION_CLIP_TO_VIEWPORT) { dev->clip_to_viewport = value ? 1 : 0; return 0; }
    if (option == RH_OPTION_GENERAL_KERNELS) { dev->general_kernels = value ? 1 : 0; return 0; }
    if (option == RH_OPTION_BINNING_TICKETS) { dev->binning_tickets = value ? 1 : 0; return 0; }
    if (option == RH_OPTION_TIMING) {
        if (value < 0 || value > 2) return fail("rh_device_set_option: RH_OPTION_TIMING takes 0, 1 or 2");
        // (the marks of a pass are counted in fives: a change of mode starts from a clean slate)
        if ((int)value != dev->timing) { for (hipEvent_t e : dev->timing_events) if (e) dev->event_pool.push_back(e); dev->timing_events.clear(); }
        dev->timing = (int)value; return 0;
    }
    if (option == RH_OPTION_EARLY_BINNING) { if (value < 0 || value > 2) return fail("rh_device_set_option: RH_OPTION_EARLY_BINNING takes 0, 1 or 2"); dev->early_binning = (int)value; return 0; }
    if (option == RH_OPTION_EAGER_COUNT) { dev->eager_count = value ? 1 : 0; return 0; }
    if (option == RH_OPTION_SETUP_LAUNCH) { dev->setup_launch = value ? 1 : 0; return 0; }
    if (option == RH_OPTION_WAVE_PRIORITY) { dev->wave_priority = value ? 1 : 0; return 0; }
    if (option == RH_OPTION_TAIL_SPLIT) { if (value < 0 || value > 3) return fail("rh_device_set_option: RH_OPTION_TAIL_SPLIT takes 0, 1, 2 or 3"); dev->tail_split = (int)value; return 0; }
    if (option == RH_OPTION_STRIP_ROW0 || option == RH_OPTION_STRIP_ROW1) {
        if (value < 0) return fail("rh_device_set_option: negative strip row");
        (option == RH_OPTION_STRIP_ROW0 ? dev->strip_row0 : dev->strip_row1) = (uint32_t)value;
        return 0;
    }
    return fail("rh_device_set_option: unknown option %d", option);
}

int rh_device_synchronize(rh_device* dev) {
    if (!dev) return fail("rh_device_synchronize: null device");
    if (enter(dev)) return 1;
    if (dev->bin_stream) RH_HIP(hipStreamSynchronize(dev->bin_stream));
    RH_HIP(hipStreamSynchronize(dev->stream));
    return 0;
}

void* rh_device_stream(rh_device* dev) { return dev && !enter(dev) ? (void*)dev->stream : nullptr; }

int rh_screen_texture(rh_device* dev, rh_texture** out) {
    if (!dev || !out) return fail("rh_screen_texture: bad argument");
    *out = dev->screen;
    return 0;
}

int rh_screen_resize(rh_device* dev, uint32_t width, uint32_t height) {
    if (!dev) return fail("rh_screen_resize: null device");
    if (width == 0 || height == 0) return 0;           // reference src/renderer.rs:57 ignores zero sizes
    if (width == dev->width && height == dev->height) return 0;
    if (enter(dev)) return 1;
    rh_texture* t = nullptr;
    if (rh_texture_create(dev, width, height, 1, RH_FORMAT_BGRA_U8, 1,
                          RH_TEXTURE_RENDER_ATTACHMENT | RH_TEXTURE_COPY_SRC | RH_TEXTURE_BINDING, &t)) return 1;
    rh_texture_release(dev->screen);
    dev->screen = t; dev->width = width; dev->height = height;
    return 0;
}

// ---- buffers ------------------------------------------------------------------------------------------------------------
int rh_buffer_create(rh_device* dev, size_t size, uint32_t usage, const void* init, size_t init_len, rh_buffer** out) {
    if (!dev || !out) return fail("rh_buffer_create: bad argument");
    if (init_len > size) return fail("rh_buffer_create: initial data (%zu bytes) larger than the buffer (%zu bytes)", init_len, size);
    if (set_device(dev)) return 1;
    rh_buffer* b = new rh_buffer();
    b->dev = dev; b->size = size; b->usage = usage;
    if (device_alloc(dev, size, &b->dptr)) { delete b; return 1; }
    device_retain(dev);
    // wgpu buffers start zero-initialised, and the allocation may be a recycled one: everything the initial data does
    // not cover is cleared (a never-set uniform, the headroom of a grown Buffer, capture rows the copy does not reach)
    hipError_t e = hipSuccess;
    const size_t filled = (init && init_len) ? init_len : 0;
    // (queued on the main stream: the binning of the next pass may read this buffer, so it must not run ahead of these)
    b->main_write_epoch = dev->main_epoch;
    if (filled) e = hipMemcpyAsync(b->dptr, init, filled, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess && filled < size) e = hipMemsetAsync((uint8_t*)b->dptr + filled, 0, size - filled, dev->stream);
    if (e != hipSuccess) { device_free(dev, size, b->dptr); delete b; device_release(dev); return fail("rh_buffer_create: initialisation failed: %s", hipGetErrorString(e)); }
    *out = b;
    return 0;
}

int rh_buffer_write(rh_buffer* buf, size_t offset, const void* data, size_t len) {
    if (!buf || (!data && len)) return fail("rh_buffer_write: bad argument");
    if (offset > buf->size || len > buf->size - offset) return fail("rh_buffer_write: range %zu+%zu exceeds buffer size %zu", offset, len, buf->size);
    if (len == 0) return 0;
    if (enter(buf->dev)) return 1;
    return cow_write(buf, offset, data, len);
}

size_t rh_buffer_size(rh_buffer* buf) { return buf ? buf->size : 0; }
void* rh_buffer_device_ptr(rh_buffer* buf) {
    if (!buf || enter(buf->dev)) return nullptr;
    // (from now on the block stays where it is, and passes that bind the buffer never bin early)
    buf->external = true;
    if (buf->upload_pending) { (void)hipStreamWaitEvent(buf->dev->stream, buf->upload_done, 0); buf->upload_pending = false; }
    return buf->dptr;
}
void rh_buffer_retain(rh_buffer* buf) { if (buf) buf->refs.fetch_add(1); }

void rh_buffer_release(rh_buffer* buf) {
    if (!buf) return;
    if (buf->refs.fetch_sub(1) != 1) return;
    rh_device* dev = buf->dev;
    (void)hipSetDevice(dev->ordinal);
    // stream-ordered reuse: everything that used this memory was submitted on dev->stream before this point (an upload
    // still on its way on the copy stream joins the main stream first)
    if (buf->upload_pending) (void)hipStreamWaitEvent(dev->stream, buf->upload_done, 0);
    if (buf->upload_done) (void)hipEventDestroy(buf->upload_done);
    device_free(dev, buf->size, buf->dptr);
    if (buf->hptr) pinned_free(dev, buf->size, buf->hptr);
    if (buf->map_event) (void)hipEventDestroy(buf->map_event);
    delete buf;
    device_release(dev);
}

int rh_buffer_map_async(rh_buffer* buf, rh_map_callback cb, void* user) {
    if (!buf) return fail("rh_buffer_map_async: null buffer");
    if (!(buf->usage & RH_BUFFER_MAP_READ)) return fail("rh_buffer_map_async: buffer was not created with MAP_READ");
    if (buf->map_state != 0) return fail("rh_buffer_map_async: buffer is already mapped or mapping");
    rh_device* dev = buf->dev;
    if (enter(dev)) return 1;
    if (!buf->hptr && pinned_alloc(dev, buf->size, &buf->hptr)) return 1;
    if (!buf->map_event) RH_HIP(hipEventCreateWithFlags(&buf->map_event, hipEventDisableTiming));
    if (buf->upload_pending) { RH_HIP(hipStreamWaitEvent(dev->stream, buf->upload_done, 0)); buf->upload_pending = false; }
    RH_HIP(hipMemcpyAsync(buf->hptr, buf->dptr, buf->size, hipMemcpyDeviceToHost, dev->stream));
    RH_HIP(hipEventRecord(buf->map_event, dev->stream));
    buf->map_state = 1; buf->cb = cb; buf->cb_user = user;
    rh_buffer_retain(buf);
    dev->pending_maps.push_back(buf);
    return 0;
}

int rh_buffer_mapped_ptr(rh_buffer* buf, const uint8_t** ptr, size_t* len) {
    if (!buf || !ptr || !len) return fail("rh_buffer_mapped_ptr: bad argument");
    if (buf->map_state != 2) return fail("rh_buffer_mapped_ptr: buffer is not mapped");
    *ptr = (const uint8_t*)buf->hptr; *len = buf->size;
    return 0;
}

int rh_buffer_unmap(rh_buffer* buf) {
    if (!buf) return fail("rh_buffer_unmap: null buffer");
    if (buf->map_state == 1) return fail("rh_buffer_unmap: mapping still pending");
    buf->map_state = 0;
    return 0;
}

static int poll_maps(rh_device* dev, int wait);
int rh_poll(rh_device* dev, int wait) {
    if (!dev) return fail("rh_poll: null device");
    if (set_device(dev) || resolve_pending(dev, wait != 0)) return 1;
    return poll_maps(dev, wait);
}
static int poll_maps(rh_device* dev, int wait) {
    size_t done = 0;
    for (; done < dev->pending_maps.size(); done++) {
        rh_buffer* b = dev->pending_maps[done];
        hipError_t e = wait ? hipEventSynchronize(b->map_event) : hipEventQuery(b->map_event);
        if (e == hipErrorNotReady) break;
        b->map_state = (e == hipSuccess) ? 2 : 0;
        if (b->cb) b->cb(b->cb_user, e == hipSuccess ? 1 : 0);
        b->cb = nullptr;
        rh_buffer_release(b);
    }
    dev->pending_maps.erase(dev->pending_maps.begin(), dev->pending_maps.begin() + done);
    return 0;
}

// ---- textures -----------------------------------------------------------------------------------------------------------
int rh_texture_create(rh_device* dev, uint32_t width, uint32_t height, uint32_t layers, uint32_t format,
                      uint32_t samples, uint32_t usage, rh_texture** out) {
    if (!dev || !out) return fail("rh_texture_create: bad argument");
    if (width == 0 || height == 0 || layers == 0) return fail("rh_texture_create: zero-sized texture");
    const uint32_t bpt = bytes_per_texel(format);
    if (bpt == 0) return fail("rh_texture_create: unknown format %u", format);
    if (samples != 1 && samples != 2 && samples != 4 && samples != 8) return fail("rh_texture_create: %u samples requested; 1, 2, 4 and 8 are implemented", samples);
    if (samples > 1) {
        if (layers != 1) return fail("rh_texture_create: a multisampled texture has one layer");
    }
    if (set_device(dev)) return 1;
    rh_texture* t = new rh_texture();
    t->dev = dev; t->width = width; t->height = height; t->layers = layers; t->format = format; t->samples = samples; t->usage = usage;
    t->bpt = bpt; t->bytes = (size_t)width * height * layers * bpt * samples;      // the samples of a pixel are adjacent
    t->poly_always = (usage & RH_TEXTURE_RENDER_ATTACHMENT) != 0;
    if (device_alloc(dev, t->bytes, &t->dptr)) { delete t; return 1; }
    device_retain(dev);
    // wgpu textures start zero-initialised
    hipError_t e = hipMemsetAsync(t->dptr, 0, t->bytes, dev->stream);
    if (e != hipSuccess) { device_free(dev, t->bytes, t->dptr); delete t; device_release(dev); return fail("rh_texture_create: clear failed: %s", hipGetErrorString(e)); }
    *out = t;
    return 0;
}

int rh_texture_write(rh_texture* tex, uint32_t x, uint32_t y, uint32_t layer, uint32_t width, uint32_t height,
                     const void* data, size_t data_len, uint32_t bytes_per_row) {
    if (!tex || !data) return fail("rh_texture_write: bad argument");
    if (tex->samples != 1) return fail("rh_texture_write: the texture is multisampled");
    if (layer >= tex->layers || x > tex->width || y > tex->height || width > tex->width - x || height > tex->height - y)
        return fail("rh_texture_write: region (%u,%u,%u)+(%u,%u) outside the %ux%ux%u texture", x, y, layer, width, height, tex->width, tex->height, tex->layers);
    if (width == 0 || height == 0) return 0;
    const size_t row_bytes = (size_t)width * tex->bpt;
    if (bytes_per_row < row_bytes) return fail("rh_texture_write: bytes_per_row %u smaller than a row (%zu bytes)", bytes_per_row, row_bytes);
    if (data_len < (size_t)bytes_per_row * (height - 1) + row_bytes) return fail("rh_texture_write: %zu bytes of data are too few for %u rows", data_len, height);
    if (enter(tex->dev)) return 1;
    uint8_t* dst = (uint8_t*)tex->dptr + (((size_t)layer * tex->height + y) * tex->width + x) * tex->bpt;
    tex->poly_valid = false;
    // whole rows of the texture, tightly packed: one contiguous upload through pinned memory
    if (width == tex->width && bytes_per_row == row_bytes) return staged_upload(tex->dev, dst, data, row_bytes * height, tex->dev->stream);
    RH_HIP(hipMemcpy2DAsync(dst, (size_t)tex->width * tex->bpt, data, bytes_per_row, row_bytes, height, hipMemcpyHostToDevice, tex->dev->stream));
    return 0;
}

int rh_texture_read(rh_texture* tex, uint32_t layer, void* out, size_t out_len) {
    if (!tex || !out) return fail("rh_texture_read: bad argument");
    if (layer >= tex->layers) return fail("rh_texture_read: layer %u out of range", layer);
    // a multisampled texture reads back as width*height*samples texels, the samples of a pixel adjacent (for tests)
    const size_t layer_bytes = (size_t)tex->width * tex->height * tex->bpt * tex->samples;
    if (out_len < layer_bytes) return fail("rh_texture_read: output too small (%zu < %zu)", out_len, layer_bytes);
    if (enter(tex->dev)) return 1;
    RH_HIP(hipMemcpyAsync(out, (uint8_t*)tex->dptr + layer * layer_bytes, layer_bytes, hipMemcpyDeviceToHost, tex->dev->stream));
    RH_HIP(hipStreamSynchronize(tex->dev->stream));
    return 0;
}

int rh_texture_size(rh_texture* tex, uint32_t* width, uint32_t* height, uint32_t* layers, uint32_t* format) {
    if (!tex) return fail("rh_texture_size: null texture");
    if (width) *width = tex->width;
    if (height) *height = tex->height;
    if (layers) *layers = tex->layers;
    if (format) *format = tex->format;
    return 0;
}

void* rh_texture_device_ptr(rh_texture* tex) { if (tex) tex->poly_always = true; return tex && !enter(tex->dev) ? tex->dptr : nullptr; }
void rh_texture_retain(rh_texture* tex) { if (tex) tex->refs.fetch_add(1); }
void rh_texture_release(rh_texture* tex) {
    if (!tex) return;
    if (tex->refs.fetch_sub(1) != 1) return;
    rh_device* dev = tex->dev;
    (void)hipSetDevice(dev->ordinal);
    device_free(dev, tex->bytes, tex->dptr);
    if (tex->poly) device_free(dev, tex->bytes / 4 * tex->poly_item_bytes * RH_POLY_COPIES, tex->poly);
    delete tex;
    device_release(dev);
}

int rh_sampler_create(rh_device* dev, uint32_t filter, rh_sampler** out) {
    if (!dev || !out) return fail("rh_sampler_create: bad argument");
    if (filter != RH_FILTER_LINEAR && filter != RH_FILTER_NEAREST) return fail("rh_sampler_create: unknown filter %u", filter);
    *out = new rh_sampler{dev, filter};
    device_retain(dev);
    return 0;
}
void rh_sampler_release(rh_sampler* s) { if (!s) return; rh_device* dev = s->dev; delete s; device_release(dev); }

// ---- modules ------------------------------------------------------------------------------------------------------------
int rh_module_load(rh_device* dev, const void* code_object, size_t size, rh_module** out) {
    if (!dev || !code_object || size < 64 || !out) return fail("rh_module_load: bad argument");
    if (set_device(dev)) return 1;
    rh_module* m = new rh_module();
    m->dev = dev;
    hipError_t e = hipModuleLoadData(&m->module, code_object);
    if (e != hipSuccess) { delete m; return fail("rh_module_load: hipModuleLoadData failed: %s (is this a gfx950 code object?)", hipGetErrorString(e)); }
    hipDeviceptr_t info_ptr = nullptr; size_t info_bytes = 0;
    e = hipModuleGetGlobal(&info_ptr, &info_bytes, m->module, "rh_module_info");
    if (e != hipSuccess || info_bytes != sizeof(rh_module_info_t)) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: code object has no rh_module_info record"); }
    e = hipMemcpy(&m->info, info_ptr, sizeof m->info, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: reading rh_module_info failed: %s", hipGetErrorString(e)); }
    if (m->info.magic != RH_MODULE_MAGIC || m->info.abi_version != RH_DEVICE_ABI_VERSION) {
        uint32_t v = m->info.abi_version;
        (void)hipModuleUnload(m->module); delete m;
        return fail("rh_module_load: code object built for device ABI %u, library expects %u; re-run the shader build", v, RH_DEVICE_ABI_VERSION);
    }
    if (m->info.stage == RH_MODULE_STAGE_VERTEX) {
        if (hipModuleGetFunction(&m->bin_count, m->module, "rh_bin_count") != hipSuccess || hipModuleGetFunction(&m->bin_pairs, m->module, "rh_bin_pairs") != hipSuccess) {
            (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: vertex code object lacks rh_bin_count / rh_bin_pairs");
        }
        if (hipModuleGetFunction(&m->bin_count_clip, m->module, "rh_bin_count_clip") != hipSuccess || hipModuleGetFunction(&m->bin_pairs_clip, m->module, "rh_bin_pairs_clip") != hipSuccess) {
            (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: vertex code object lacks rh_bin_count_clip / rh_bin_pairs_clip");
        }
        int per_cu = 0;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, m->bin_pairs, RH_PAIR_THREADS, 0) == hipSuccess && per_cu > 0)
            m->resident_pair_parts = (uint32_t)per_cu * dev->compute_units;
    } else if (m->info.stage == RH_MODULE_STAGE_FRAGMENT) {
        const char* names[RH_MAX_COLOR_TARGETS] = { "rh_tile_1", "rh_tile_2", "rh_tile_3", "rh_tile_4" };
        for (int i = 0; i < RH_MAX_COLOR_TARGETS; i++)
            if (hipModuleGetFunction(&m->tile[i], m->module, names[i]) != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object lacks %s", names[i]); }
        const char* wide_names[RH_MAX_COLOR_TARGETS] = { "rh_tile_w1", "rh_tile_w2", "rh_tile_w3", "rh_tile_w4" };
        for (int i = 0; i < RH_MAX_COLOR_TARGETS; i++)
            if (hipModuleGetFunction(&m->tile_wide[i], m->module, wide_names[i]) != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object lacks %s", wide_names[i]); }
        const char* setup_names[4] = { "rh_setup", "rh_setup_ms2", "rh_setup_ms4", "rh_setup_ms8" };
        for (int i = 0; i < 4; i++)
            if (hipModuleGetFunction(&m->setup[i], m->module, setup_names[i]) != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object lacks %s", setup_names[i]); }
        const char* clip_names[4] = { "rh_setup_clip", "rh_setup_clip_ms2", "rh_setup_clip_ms4", "rh_setup_clip_ms8" };
        for (int i = 0; i < 4; i++)
            if (hipModuleGetFunction(&m->setup_clip[i], m->module, clip_names[i]) != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object lacks %s", clip_names[i]); }
        if (m->info.rec_bytes == 0 || m->info.rec_bytes % 16 || m->info.ext_bytes % 16) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object declares a malformed record size"); }
        const char* ms_names[3] = { "rh_tile_ms2", "rh_tile_ms4", "rh_tile_ms8" };
        for (int i = 0; i < 3; i++)
            if (hipModuleGetFunction(&m->tile_ms[i], m->module, ms_names[i]) != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object lacks %s", ms_names[i]); }
        const char* msw_names[3] = { "rh_tile_msw2", "rh_tile_msw4", "rh_tile_msw8" };
        for (int i = 0; i < 3; i++)
            if (hipModuleGetFunction(&m->tile_ms_wide[i], m->module, msw_names[i]) != hipSuccess) { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: fragment code object lacks %s", msw_names[i]); }
    } else { (void)hipModuleUnload(m->module); delete m; return fail("rh_module_load: unknown stage"); }
    device_retain(dev);
    *out = m;
    return 0;
}
void rh_module_retain(rh_module* m) { if (m) m->refs.fetch_add(1); }
void rh_module_release(rh_module* m) {
    if (!m) return;
    if (m->refs.fetch_sub(1) != 1) return;
    rh_device* dev = m->dev;
    (void)hipSetDevice(dev->ordinal);
    (void)hipModuleUnload(m->module);
    delete m;
    device_release(dev);
}

// ---- pipelines -----------------------------------------------------------------------------------------------------------
int rh_pipeline_create(rh_device* dev, const rh_pipeline_desc* d, rh_pipeline** out) {
    if (!dev || !d || !out) return fail("rh_pipeline_create: bad argument");
    if (!d->vertex_module || d->vertex_module->info.stage != RH_MODULE_STAGE_VERTEX) return fail("rh_pipeline_create: vertex_module is not a vertex-stage code object");
    if (!d->fragment_module || d->fragment_module->info.stage != RH_MODULE_STAGE_FRAGMENT) return fail("rh_pipeline_create: fragment_module is not a fragment-stage code object");
    if (d->vertex_module->info.vertex_tag != d->fragment_module->info.vertex_tag || d->vertex_module->info.n_varyings != d->fragment_module->info.n_varyings)
        return fail("rh_pipeline_create: the fragment code object was not built against this vertex code object");
    if (d->n_attributes > RH_MAX_ATTRIBUTES) return fail("rh_pipeline_create: %u attributes (max %d)", d->n_attributes, RH_MAX_ATTRIBUTES);
    if (d->n_slots > RH_DEV_MAX_SLOTS) return fail("rh_pipeline_create: %u binding slots (max %d implemented)", d->n_slots, RH_DEV_MAX_SLOTS);
    if (d->n_targets > RH_MAX_COLOR_TARGETS) return fail("rh_pipeline_create: %u colour targets (max %d)", d->n_targets, RH_MAX_COLOR_TARGETS);
    if (d->samples != 1 && d->samples != 2 && d->samples != 4 && d->samples != 8) return fail("rh_pipeline_create: %u samples requested; 1, 2, 4 and 8 are implemented", d->samples);
    if (d->topology != RH_TOPOLOGY_TRIANGLE_LIST && d->topology != RH_TOPOLOGY_TRIANGLE_STRIP) return fail("rh_pipeline_create: unknown topology %u", d->topology);
    rh_pipeline* p = new rh_pipeline();
    p->dev = dev; p->topology = d->topology; p->samples = d->samples;
    for (uint32_t i = 0; i < d->n_attributes; i++) {
        const rh_attribute_desc& a = d->attributes[i];
        if (a.location >= RH_MAX_ATTRIBUTES || a.size < 1 || a.size > 4) { delete p; return fail("rh_pipeline_create: unsupported attribute (location %u, size %u)", a.location, a.size); }
        p->attributes.push_back(a);
    }
    for (uint32_t i = 0; i < d->n_targets; i++) {
        const rh_color_target_desc& t = d->targets[i];
        if (t.format > RH_FORMAT_RGBA_F32) { delete p; return fail("rh_pipeline_create: unknown colour target format %u", t.format); }
        if (t.src_factor > RH_BLEND_ONE_MINUS_SRC_ALPHA || t.dst_factor > RH_BLEND_ONE_MINUS_SRC_ALPHA) { delete p; return fail("rh_pipeline_create: unknown blend factor"); }
        if (t.src_factor != d->targets[0].src_factor || t.dst_factor != d->targets[0].dst_factor) { delete p; return fail("rh_pipeline_create: all colour targets must share one blend state"); }
        p->targets.push_back(t);
    }
    for (uint32_t i = 0; i < d->n_slots; i++) {
        const rh_slot_desc& s = d->slots[i];
        SlotRef r; r.kind = s.kind; r.visibility = s.visibility;
        if (!s.resource) { delete p; return fail("rh_pipeline_create: slot %u has no resource", i); }
        switch (s.kind) {
        case RH_SLOT_STORAGE: case RH_SLOT_UNIFORM: r.buf = (rh_buffer*)s.resource; break;
        case RH_SLOT_TEXTURE: {
            r.tex = (rh_texture*)s.resource;
            if (r.tex->format > RH_FORMAT_RGBA_F32) { delete p; return fail("rh_pipeline_create: sampling format %u is not implemented", r.tex->format); }
            if (r.tex->samples != 1) { delete p; return fail("rh_pipeline_create: slot %u binds a multisampled texture; sampling those is not implemented", i); }
            break;
        }
        case RH_SLOT_SAMPLER: r.filter = ((rh_sampler*)s.resource)->filter; break;
        default: delete p; return fail("rh_pipeline_create: unknown slot kind %u", s.kind);
        }
        p->slots.push_back(r);
    }
    for (SlotRef& r : p->slots) { if (r.buf) rh_buffer_retain(r.buf); if (r.tex) rh_texture_retain(r.tex); }
    p->vs = d->vertex_module; p->fs = d->fragment_module;
    rh_module_retain(p->vs); rh_module_retain(p->fs);
    device_retain(dev);
    *out = p;
    return 0;
}
void rh_pipeline_retain(rh_pipeline* p) { if (p) p->refs.fetch_add(1); }
void rh_pipeline_release(rh_pipeline* p) {
    if (!p) return;
    if (p->refs.fetch_sub(1) != 1) return;
    for (SlotRef& r : p->slots) { if (r.buf) rh_buffer_release(r.buf); if (r.tex) rh_texture_release(r.tex); }
    rh_module_release(p->vs); rh_module_release(p->fs);
    rh_device* dev = p->dev;
    delete p;
    device_release(dev);
}

// ---- command encoding ------------------------------------------------------------------------------------------------------
int rh_cmd_begin(rh_device* dev, rh_cmd** out) {
    if (!dev || !out) return fail("rh_cmd_begin: bad argument");
    rh_cmd* c = new rh_cmd(); c->dev = dev; *out = c;
    return 0;
}

static void release_cmds(std::vector<Cmd>& cmds) {
    for (Cmd& c : cmds) {
        if (c.kind == 0) {
            rh_pipeline_release(c.pass.pipeline);
            for (auto& a : c.pass.attachments) { rh_texture_release(a.texture); rh_texture_release(a.resolve_target); }
            for (rh_buffer* b : c.pass.vertex_buffers) rh_buffer_release(b);
        } else { rh_texture_release(c.copy.tex); rh_buffer_release(c.copy.buf); }
    }
    cmds.clear();
}

int rh_cmd_render_pass(rh_cmd* cmd, const rh_pass_desc* d) {
    if (!cmd || !d || !d->pipeline) return fail("rh_cmd_render_pass: bad argument");
    rh_pipeline* p = d->pipeline;
    if (d->n_attachments == 0) return 0;                        // reference src/render_pass.rs:26
    if (d->n_attachments != p->targets.size()) return fail("rh_cmd_render_pass: %u attachments for a pipeline with %zu colour targets", d->n_attachments, p->targets.size());
    if (d->n_vertex_buffers != p->attributes.size()) return fail("rh_cmd_render_pass: %u vertex buffers for a pipeline with %zu attributes", d->n_vertex_buffers, p->attributes.size());
    const rh_texture* t0 = d->attachments[0].texture;
    for (uint32_t i = 0; i < d->n_attachments; i++) {
        const rh_texture* t = d->attachments[i].texture;
        if (!t) return fail("rh_cmd_render_pass: attachment %u has no texture", i);
        if (t->format != p->targets[i].format) return fail("rh_cmd_render_pass: attachment %u has format %u but the pipeline expects %u", i, t->format, p->targets[i].format);
        if (t->width != t0->width || t->height != t0->height) return fail("rh_cmd_render_pass: attachment %u is %ux%u, attachment 0 is %ux%u", i, t->width, t->height, t0->width, t0->height);
        if (!(t->usage & RH_TEXTURE_RENDER_ATTACHMENT)) return fail("rh_cmd_render_pass: attachment %u was not created renderable", i);
        if (t->samples != p->samples) return fail("rh_cmd_render_pass: attachment %u has %u samples but the pipeline was created for %u", i, t->samples, p->samples);
        const rh_texture* rt = d->attachments[i].resolve_target;
        if (rt) {
            if (t->samples == 1) return fail("rh_cmd_render_pass: attachment %u has a resolve target but is not multisampled", i);
            if (rt->samples != 1 || rt->format != t->format || rt->width != t->width || rt->height != t->height || rt->layers != 1)
                return fail("rh_cmd_render_pass: the resolve target of attachment %u must be a single-sampled %ux%u texture of format %u", i, t->width, t->height, t->format);
            if (!(rt->usage & RH_TEXTURE_RENDER_ATTACHMENT)) return fail("rh_cmd_render_pass: the resolve target of attachment %u was not created renderable", i);
        }
    }
    // The reference attaches its one multisampled texture to every target of the pipeline (src/render_pass.rs:76-79),
    // which is only valid for a single attachment; the recording attachment is single-sampled and cannot join either.
    if (p->samples > 1 && d->n_attachments != 1) return fail("rh_cmd_render_pass: a multisampled pass takes exactly one colour attachment, got %u", d->n_attachments);
    for (uint32_t i = 0; i < d->n_vertex_buffers; i++) {
        rh_buffer* b = d->vertex_buffers[i];
        if (!b) return fail("rh_cmd_render_pass: vertex buffer %u is null", i);
        const size_t need = (size_t)d->vertex_count * p->attributes[i].size * sizeof(float);
        if (b->size < need) return fail("rh_cmd_render_pass: vertex buffer %u holds %zu bytes, %u vertices need %zu", i, b->size, d->vertex_count, need);
    }
    Cmd c; c.kind = 0;
    c.pass.pipeline = p; rh_pipeline_retain(p);
    for (uint32_t i = 0; i < d->n_attachments; i++) { c.pass.attachments.push_back(d->attachments[i]); rh_texture_retain(d->attachments[i].texture); rh_texture_retain(d->attachments[i].resolve_target); }
    for (uint32_t i = 0; i < d->n_vertex_buffers; i++) { c.pass.vertex_buffers.push_back(d->vertex_buffers[i]); rh_buffer_retain(d->vertex_buffers[i]); }
    c.pass.has_viewport = d->has_viewport != 0;
    if (c.pass.has_viewport) memcpy(c.pass.viewport, d->viewport, sizeof c.pass.viewport);
    c.pass.vertex_count = d->vertex_count; c.pass.instance_count = d->instance_count;
    cmd->cmds.push_back(std::move(c));
    return 0;
}

int rh_cmd_copy_texture_to_buffer(rh_cmd* cmd, rh_texture* tex, uint32_t origin_x, uint32_t origin_y,
                                  rh_buffer* buf, uint32_t bytes_per_row, uint32_t rows_per_image, uint32_t width, uint32_t height) {
    if (!cmd || !tex || !buf) return fail("rh_cmd_copy_texture_to_buffer: bad argument");
    if (!(tex->usage & RH_TEXTURE_COPY_SRC)) return fail("rh_cmd_copy_texture_to_buffer: texture was not created copyable");
    if (bytes_per_row % RH_COPY_BYTES_PER_ROW_ALIGNMENT) return fail("rh_cmd_copy_texture_to_buffer: bytes_per_row %u is not a multiple of %d", bytes_per_row, RH_COPY_BYTES_PER_ROW_ALIGNMENT);
    if (origin_x > tex->width || origin_y > tex->height || width > tex->width - origin_x || height > tex->height - origin_y)
        return fail("rh_cmd_copy_texture_to_buffer: region (%u,%u)+(%u,%u) outside the %ux%u texture", origin_x, origin_y, width, height, tex->width, tex->height);
    if ((size_t)width * tex->bpt > bytes_per_row) return fail("rh_cmd_copy_texture_to_buffer: bytes_per_row %u smaller than a row", bytes_per_row);
    if (height > 0 && (size_t)bytes_per_row * height > buf->size) return fail("rh_cmd_copy_texture_to_buffer: buffer of %zu bytes too small for %u rows of %u bytes", buf->size, height, bytes_per_row);
    Cmd c; c.kind = 1;
    c.copy.tex = tex; c.copy.buf = buf; rh_texture_retain(tex); rh_buffer_retain(buf);
    c.copy.ox = origin_x; c.copy.oy = origin_y; c.copy.bytes_per_row = bytes_per_row; c.copy.rows_per_image = rows_per_image;
    c.copy.width = width; c.copy.height = height;
    cmd->cmds.push_back(std::move(c));
    return 0;
}

int rh_cmd_finish(rh_cmd* cmd, rh_cmdbuf** out) {
    if (!cmd || !out) return fail("rh_cmd_finish: bad argument");
    rh_cmdbuf* cb = new rh_cmdbuf(); cb->dev = cmd->dev; cb->cmds = std::move(cmd->cmds);
    delete cmd;
    *out = cb;
    return 0;
}

void rh_cmdbuf_release(rh_cmdbuf* cb) { if (!cb) return; release_cmds(cb->cmds); delete cb; }

} // extern "C"

// ---- execution ---------------------------------------------------------------------------------------------------------------
namespace {

int timing_mark(rh_device* dev, hipStream_t on = nullptr) {
    if (!dev->timing) return 0;
    // (a pass makes five marks, in order; mode 2 keeps the two around the tile kernel only -- every mark is a bubble of
    // some 7 us in the stream)
    if (dev->timing == 2 && dev->timing_events.size() % 5 < 3) { dev->timing_events.push_back(nullptr); return 0; }
    hipEvent_t e = nullptr;
    if (!dev->event_pool.empty()) { e = dev->event_pool.back(); dev->event_pool.pop_back(); }
    else RH_HIP(hipEventCreate(&e));
    RH_HIP(hipEventRecord(e, on ? on : dev->stream));
    dev->timing_events.push_back(e);
    return 0;
}

int launch_module_kernel(hipFunction_t fn, uint32_t grid, uint32_t block, void* args, size_t args_size, hipStream_t s) {
    void* config[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_size, HIP_LAUNCH_PARAM_END };
    RH_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, s, nullptr, config));
    return 0;
}

// The copy of a colour attachment into a buffer right behind the pass that wrote it (the capture of a recorded frame,
// reference src/video_recorder.rs:107-128) needs no launch and no second trip of the texels through memory when the
// tile kernel can write the buffer as it writes the tiles back: single-sampled pass over the whole target, 4-byte
// UNORM attachments only.
bool fusable_copy(const rh_device* dev, const PassCmd& c, const CopyCmd& cp) {
    if (dev->strip_row1 != 0 || c.pipeline->samples != 1 || cp.width == 0 || cp.height == 0) return false;
    bool found = false;
    for (const rh_attachment_desc& a : c.attachments) {
        const uint32_t f = a.texture->format;
        if (f == RH_FORMAT_RGBA_F16 || f == RH_FORMAT_RGBA_F32) return false;
        if (a.texture == cp.tex) found = a.texture->bpt == 4;
    }
    return found;
}

// A pipeline that clips goes back to the usual kernels after RH_CLIP_IDLE_PASSES passes in a row that clipped nothing.
constexpr uint32_t RH_CLIP_IDLE_PASSES = 16;
void note_clipping(rh_pipeline* pl, bool clipped) {
    if (clipped) { pl->passes_without_clipping = 0; return; }
    if (++pl->passes_without_clipping >= RH_CLIP_IDLE_PASSES) { pl->clips = false; pl->passes_without_clipping = 0; }
}

// `ss`: the scratch areas the pass works in; `no_alloc`: called from rh_submit_static, which must not allocate -- timing
// marks (they create events) and footprint copies that do not exist yet are left out.
// `fused`: a texture-to-buffer copy that follows the pass in its command buffer and that the tile kernel performs as it
// writes its tiles back (see fusable_copy), or null.
int execute_pass(rh_device* dev, const PassCmd& c, bool may_defer, rh_device::PassScratch& ss_in, bool no_alloc, const CopyCmd* fused) {
    rh_pipeline* pl = c.pipeline;
    const rh_texture* t0 = c.attachments[0].texture;
    const uint32_t W = t0->width, H = t0->height;

    rh_draw_params p;
    memset(&p, 0, sizeof p);
    for (size_t i = 0; i < pl->attributes.size(); i++) {
        p.attr[pl->attributes[i].location] = (const float*)c.vertex_buffers[i]->dptr;
        p.attr_size[pl->attributes[i].location] = pl->attributes[i].size;
    }
    for (size_t i = 0; i < pl->slots.size(); i++) {
        const SlotRef& s = pl->slots[i];
        rh_slot_dev& d = p.slots[i];
        if (s.buf) { d.ptr = s.buf->dptr; d.size_bytes = (uint32_t)std::min<size_t>(s.buf->size, 0xffffffffu); }
        if (s.tex) {
            d.ptr = s.tex->dptr; d.width = s.tex->width; d.height = s.tex->height; d.layers = s.tex->layers; d.format = s.tex->format;
            // a texture's sampler, when it has one, occupies the next slot (reference src/pipeline.rs:105-112)
            d.filter = (i + 1 < pl->slots.size() && pl->slots[i + 1].kind == RH_SLOT_SAMPLER) ? pl->slots[i + 1].filter : RH_FILTER_NEAREST;
        }
        if (s.kind == RH_SLOT_SAMPLER) d.filter = s.filter;
    }
    float vx = 0.0f, vy = 0.0f, vw = (float)W, vh = (float)H;
    if (c.has_viewport) { vx = c.viewport[0]; vy = c.viewport[1]; vw = c.viewport[2]; vh = c.viewport[3]; }
    p.vp_hw = vw * 0.5f; p.vp_hh = vh * 0.5f; p.vp_ox = vx + p.vp_hw; p.vp_oy = vy + p.vp_hh;
    p.sc_x0 = 0; p.sc_y0 = 0; p.sc_x1 = (int32_t)W; p.sc_y1 = (int32_t)H;
    if (dev->clip_to_viewport) {
        // pixel centre (i + 0.5) inside [vx, vx + vw)
        p.sc_x0 = std::max(0, (int)std::ceil((double)vx - 0.5));
        p.sc_y0 = std::max(0, (int)std::ceil((double)vy - 0.5));
        p.sc_x1 = std::min((int)W, (int)std::ceil((double)vx + (double)vw - 0.5));
        p.sc_y1 = std::min((int)H, (int)std::ceil((double)vy + (double)vh - 0.5));
    }
    p.width = W; p.height = H;
    p.tiles_x = (W + RH_TILE_W - 1) / RH_TILE_W; p.tiles_y = (H + RH_TILE_H - 1) / RH_TILE_H;
    if (p.tiles_x > RH_RANGE_MAX_TILES_X) return fail("render pass: a target %u pixels wide exceeds %u tile columns", W, RH_RANGE_MAX_TILES_X);
    const uint32_t n_tiles = p.tiles_x * p.tiles_y;
    // sort-first strip: only tile rows [row0, row1) are rendered (and cleared) by this device
    uint32_t tile_row0 = 0, tile_row1 = p.tiles_y;
    if (dev->strip_row1 != 0) {
        const uint32_t r0 = dev->strip_row0, r1 = std::min(dev->strip_row1, H);
        if (r0 % RH_TILE_H || (r1 % RH_TILE_H && r1 != H) || r0 > r1) return fail("render pass: strip rows [%u, %u) are not multiples of %d", r0, r1, RH_TILE_H);
        tile_row0 = r0 / RH_TILE_H; tile_row1 = (r1 + RH_TILE_H - 1) / RH_TILE_H;
        p.sc_y0 = std::max(p.sc_y0, (int32_t)r0); p.sc_y1 = std::min(p.sc_y1, (int32_t)r1);
    }
    p.tile_first = tile_row0 * p.tiles_x;
    const uint32_t n_launch_tiles = (tile_row1 - tile_row0) * p.tiles_x;
    p.topology = pl->topology;
    p.vertex_count = c.vertex_count; p.instance_count = c.instance_count;
    p.tpi = pl->topology == RH_TOPOLOGY_TRIANGLE_STRIP ? (c.vertex_count >= 3 ? c.vertex_count - 2 : 0) : c.vertex_count / 3;
    p.cpi = (p.tpi + 1) / 2;
    const uint64_t n_clusters64 = (uint64_t)p.cpi * c.instance_count;
    // (the top three bits of a (tile, cluster) pair's value number the sub-records of a clipped cluster, the next says
    // whether the cluster's records come from the set-up launch)
    if (n_clusters64 >= (1ull << 28)) return fail("render pass: %llu triangle pairs exceed the 28-bit cluster index", (unsigned long long)n_clusters64);
    p.n_clusters = (uint32_t)n_clusters64;
    p.samples = pl->samples; p.bbox_pad = pl->samples > 1 ? 128 : 0;
    p.no_lean = dev->setup_launch ? 1u : 0u;
    const bool empty_scissor = p.sc_x0 >= p.sc_x1 || p.sc_y0 >= p.sc_y1;
    const uint32_t n_clusters = empty_scissor ? 0u : p.n_clusters;

    if (no_alloc && pl->clips) return fail("rh_submit_static: the pipeline has triangles that cross a clip plane; the kernels that clip them need private memory, which the runtime allocates");
    hipStream_t const sm = dev->stream;         // the main stream: the tile kernel and everything else
    uint64_t total = 0;                          // pairs, or the room for them when the count has not come back yet
    const unsigned long long* pair_total = nullptr;
    const int ms_index = pl->samples == 1 ? 0 : (pl->samples == 2 ? 1 : (pl->samples == 4 ? 2 : 3));
    const uint32_t rec_bytes = pl->fs->info.rec_bytes;
    const uint32_t ext_bytes = pl->fs->info.ext_bytes + (pl->samples > 1 ? 4u * RH_MS_EXTRA_WORDS(pl->samples) : 0u);
    // Once a pass has been seen the next ones are queued whole, without waiting for the pair count, with room for
    // `pair_guess` pairs: the kernels read the count on the device.  (The option that forces the general kernels also
    // forces the waiting form, so that tests cover both.)
    const uint32_t key_bits = bits_for(n_tiles), sort_passes = (key_bits + 7) / 8;
    const bool deferred = may_defer && n_clusters > 0 && dev->pair_guess > 0 && dev->pair_guess < (1ull << 30) && !dev->general_kernels &&
                          sort_passes <= RH_SORT_MAX_PASSES;
    // The growing scratch areas come in two sets that consecutive passes alternate between; binning and set-up of a pass
    // queued whole run on the binning stream (see rh_device::bin_stream) unless the vertex stage reads a texture, which a
    // pass still at work could be rendering.
    const bool alternating = &ss_in == &dev->grown;
    const int set = alternating ? dev->scratch_flip : 0;
    rh_device::PassScratch& ss = (alternating && set) ? dev->grown2 : ss_in;
    bool vertex_reads_texture = false;
    for (const SlotRef& sl : pl->slots) if (sl.tex && (sl.visibility & RH_STAGE_VERTEX)) vertex_reads_texture = true;
    // (only when nothing but the tile kernel of the pass before has been queued on the main stream since that kernel's
    // start: anything else -- an upload, a copy -- may be what this pass's vertex stage reads, the binning would have to
    // wait for it and there would be nothing left to run beside)
    // (and every buffer the vertex stage can read is settled: no write to it queued on the main stream since the latest
    // `before_tile`, which is what the binning stream waits for; an upload by rh_buffer_write runs on the copy stream and
    // is waited for by itself, below)
    bool buffers_settled = true;
    auto unsettled = [&](const rh_buffer* b) { return b && (b->external || b->main_write_epoch >= dev->main_epoch); };
    for (const rh_buffer* b : c.vertex_buffers) if (unsettled(b)) buffers_settled = false;
    for (const SlotRef& sl : pl->slots) if (unsettled(sl.buf)) buffers_settled = false;
    // (and no compressor on the device: with its two streams beside the three of a pass binned early, passes were seen
    // to stop for good at the first one binned early, under some assignments of the streams to hardware queues and not
    // under others; a recording run renders on the main stream alone, as it always has for a scene that moves)
    // (and, in the default mode, only a pass with a long binning chain: beside a tile kernel the chain gets the places
    // that kernel's workgroups leave, slows it down and still ends behind it -- measured at 4K, a frame with 2 M pairs
    // gains or loses 2 .. 3 % by it depending on what else is in the streams, one with 3.6 M or more gains 1 .. 4 %)
    const bool early_pays = dev->early_binning >= 2 || dev->pair_guess >= dev->early_min_pairs;
    const bool early = deferred && alternating && dev->early_binning && early_pays && !vertex_reads_texture && buffers_settled && dev->before_tile_valid &&
                       (dev->compressors == 0 || getenv("RH_EXP_EARLY_WITH_COMPRESSOR"));
    if (early && !dev->bin_stream) RH_HIP(hipStreamCreateWithPriority(&dev->bin_stream, hipStreamNonBlocking, dev->bin_priority));
    if (early) dev->stats.early++;
    hipStream_t const s = early ? dev->bin_stream : sm;     // binning and set-up
    if (early) {
        RH_HIP(hipStreamWaitEvent(s, dev->before_tile, 0));
        if (dev->tile_done_valid[set]) RH_HIP(hipStreamWaitEvent(s, dev->tile_done[set], 0));
    }
    {   // uploads still on their way on the copy stream
        auto join = [&](rh_buffer* b) -> int {
            if (!b || !b->upload_pending) return 0;
            if (hipEventQuery(b->upload_done) == hipSuccess) { b->upload_pending = false; return 0; }
            RH_HIP(hipStreamWaitEvent(s, b->upload_done, 0));       // (the tile kernel waits for the set-up on `s`)
            return 0;
        };
        for (rh_buffer* b : c.vertex_buffers) if (join(b)) return 1;
        for (const SlotRef& sl : pl->slots) if (join(sl.buf)) return 1;
    }
    const uint32_t n_parts = (n_clusters + 255) / 256, n_pair_parts = (n_clusters + RH_PAIR_PART - 1) / RH_PAIR_PART;
    // one area, cleared by one fill: [binning state | start of every tile | end of every tile | look-back status words]
    const size_t cluster_status_at = STATE_HEADER_BYTES + (size_t)n_tiles * 8;
    const size_t sort_status_at = cluster_status_at + (deferred ? (size_t)n_pair_parts * 8 : 0);
    const size_t state_bytes = (sort_status_at + (deferred ? (size_t)sort_passes * rhk::sort_status_words((uint32_t)dev->pair_guess) * 4 : 0) + 15) & ~(size_t)15;
    if (ss.tile_start.ensure(state_bytes)) return 1;
    uint8_t* const state_base = ss.tile_start.as<uint8_t>();
    rh_bin_state* const state = reinterpret_cast<rh_bin_state*>(state_base);
    uint32_t* const tile_start = reinterpret_cast<uint32_t*>(state_base + STATE_HEADER_BYTES); uint32_t* const tile_end = tile_start + n_tiles;
    if (!no_alloc && timing_mark(dev, s)) return 1;
    if (dbg_sync(s, "waits before the pass")) return 1;
    rhk::clear(state_base, state_bytes, s);
    RH_HIP(hipGetLastError());
    if (dbg_sync(s, "clear")) return 1;
    uint32_t* sorted_keys = nullptr; uint32_t* sorted_vals = nullptr;
    if (deferred) {
        const uint32_t n = (uint32_t)dev->pair_guess;
        if (ss.keys_a.ensure((size_t)n * 4) || ss.vals_a.ensure((size_t)n * 4) || ss.keys_b.ensure((size_t)n * 4) || ss.vals_b.ensure((size_t)n * 4)) return 1;
        rh_bin_pairs_args ba;
        memset(&ba, 0, sizeof ba);
        ba.p = p; ba.p.n_clusters = n_clusters;
        ba.state = state; ba.cluster_status = reinterpret_cast<unsigned long long*>(state_base + cluster_status_at);
        ba.keys = ss.keys_a.as<uint32_t>(); ba.vals = ss.vals_a.as<uint32_t>();
        ba.capacity = n; ba.sort_passes = sort_passes; ba.host_total = dev->h_total;
        // (beside another kernel not every workgroup of the chain may be on the device at once: tickets)
        ba.ticketed = (early || dev->binning_tickets || n_pair_parts > pl->vs->resident_pair_parts) ? 1u : 0u;
        if (launch_module_kernel(pl->clips ? pl->vs->bin_pairs_clip : pl->vs->bin_pairs, n_pair_parts, RH_PAIR_THREADS, &ba, sizeof ba, s)) return 1;
        if (dbg_sync(s, ba.ticketed ? "bin_pairs (ticketed)" : "bin_pairs")) return 1;
        const int where = rhk::sort_passes_chained(ss.keys_a.as<uint32_t>(), ss.vals_a.as<uint32_t>(), ss.keys_b.as<uint32_t>(), ss.vals_b.as<uint32_t>(),
                                                   n, key_bits, state, reinterpret_cast<uint32_t*>(state_base + sort_status_at), dev->h_total, (early || dev->binning_tickets) ? 0u : dev->resident_sort_parts, s);
        RH_HIP(hipGetLastError());
        if (dbg_sync(s, "chained sort")) return 1;
        RH_HIP(hipEventRecord(dev->counted, s));
        sorted_keys = where ? ss.keys_b.as<uint32_t>() : ss.keys_a.as<uint32_t>();
        sorted_vals = where ? ss.vals_b.as<uint32_t>() : ss.vals_a.as<uint32_t>();
        total = n;
        pair_total = &state->total;
    } else if (n_clusters > 0) {
        if (ss.ranges.ensure((size_t)n_clusters * sizeof(rh_tile_range)) || ss.block_sums.ensure((size_t)n_parts * 4) ||
            ss.chunk_tmp.ensure((size_t)rhk::scan_chunks_for(n_parts) * 4)) return 1;
        rh_bin_count_args ba;
        ba.p = p; ba.p.n_clusters = n_clusters;
        ba.ranges = ss.ranges.as<rh_tile_range>(); ba.block_sums = ss.block_sums.as<uint32_t>(); ba.wants_clip = &state->wants_clip;
        if (launch_module_kernel(pl->clips ? pl->vs->bin_count_clip : pl->vs->bin_count, n_parts, 256, &ba, sizeof ba, s)) return 1;
        rhk::exclusive_scan(ss.block_sums.as<uint32_t>(), n_parts, ss.chunk_tmp.as<uint32_t>(), dev->d_total, s, dev->general_kernels != 0);
        RH_HIP(hipMemcpyAsync(dev->h_total, dev->d_total, 8, hipMemcpyDeviceToHost, s));
        RH_HIP(hipMemcpyAsync(dev->h_total + 2, &state->wants_clip, 4, hipMemcpyDeviceToHost, s));
        RH_HIP(hipStreamSynchronize(s));
        if ((uint32_t)dev->h_total[2] != 0u && !pl->clips) {
            // a triangle crosses a clip plane: the pass starts again with the kernels that clip
            if (no_alloc) return fail("rh_submit_static: the pass has triangles that cross a clip plane; the kernels that clip them need private memory, which the runtime allocates");
            pl->clips = true;
            if (dev->timing && !dev->timing_events.empty()) { if (dev->timing_events.back()) dev->event_pool.push_back(dev->timing_events.back()); dev->timing_events.pop_back(); }
            return execute_pass(dev, c, false, ss_in, no_alloc, fused);
        }
        if (pl->clips) note_clipping(pl, (uint32_t)dev->h_total[2] != 0u);
        total = dev->h_total[0];
        if (total >= 0x7fffffffull) return fail("render pass: %llu (tile, cluster) pairs exceed the 31-bit pair index", (unsigned long long)total);
        dev->pair_guess = std::max<uint64_t>(dev->pair_guess, std::min<uint64_t>(total + total / 4 + 4096, 0x7ffffffeull));
        if (total > 0) {
            const uint32_t n = (uint32_t)total;
            const uint32_t sort_blocks = rhk::sort_blocks_for(n);
            if (ss.keys_a.ensure((size_t)n * 4) || ss.vals_a.ensure((size_t)n * 4) || ss.keys_b.ensure((size_t)n * 4) || ss.vals_b.ensure((size_t)n * 4) ||
                ss.hist.ensure((size_t)sort_blocks * 256 * 4) || ss.chunk_tmp.ensure((size_t)rhk::scan_chunks_for(sort_blocks * 256) * 4)) return 1;
            rhk::bin_emit(ss.ranges.as<rh_tile_range>(), n_clusters, p.tiles_x, ss.block_sums.as<uint32_t>(),
                          ss.keys_a.as<uint32_t>(), ss.vals_a.as<uint32_t>(), s);
            const int where = rhk::radix_sort_pairs(ss.keys_a.as<uint32_t>(), ss.vals_a.as<uint32_t>(), ss.keys_b.as<uint32_t>(), ss.vals_b.as<uint32_t>(),
                                                    n, key_bits, ss.hist.as<uint32_t>(), ss.chunk_tmp.as<uint32_t>(), s, dev->general_kernels != 0);
            sorted_keys = where ? ss.keys_b.as<uint32_t>() : ss.keys_a.as<uint32_t>();
            sorted_vals = where ? ss.vals_b.as<uint32_t>() : ss.vals_a.as<uint32_t>();
        }
    }
    if (total > 0) {
        // triangle set-up: one record per pair, in list order; also finds where the list of every tile starts and ends
        const uint32_t n = (uint32_t)total;
        if (ss.rec.ensure((size_t)n * rec_bytes) || ss.ext.ensure((size_t)n * ext_bytes)) return 1;
        rh_setup_args sa;
        memset(&sa, 0, sizeof sa);
        sa.p = p; sa.p.n_clusters = n_clusters;
        sa.clusters = sorted_vals; sa.tiles = sorted_keys;
        sa.rec = ss.rec.as<uint8_t>(); sa.ext = ss.ext.as<uint8_t>(); sa.ext_bytes = ext_bytes;
        sa.n_pairs = n; sa.pair_total = pair_total;
        sa.tile_start = tile_start; sa.tile_end = tile_end;
        if (!no_alloc && timing_mark(dev, s)) return 1;
        if (launch_module_kernel(pl->clips ? pl->fs->setup_clip[ms_index] : pl->fs->setup[ms_index], (n + 255) / 256, 256, &sa, sizeof sa, s)) return 1;
    } else if (!no_alloc && timing_mark(dev, s)) return 1;
    if (dbg_sync(s, "set-up")) return 1;
    if (!no_alloc && timing_mark(dev, s)) return 1;          // the end of the set-up
    // the streams meet: the tile kernel waits for the set-up
    if (early) { RH_HIP(hipEventRecord(dev->setup_done, s)); RH_HIP(hipStreamWaitEvent(sm, dev->setup_done, 0)); }

    rh_tile_args ta;
    memset(&ta, 0, sizeof ta);
    ta.p = p; ta.p.n_clusters = n_clusters;
    ta.rec = ss.rec.as<uint8_t>(); ta.vals = sorted_vals; ta.ext = ss.ext.as<uint8_t>(); ta.ext_bytes = ext_bytes;
    ta.tile_start = tile_start; ta.tile_end = tile_end;
    ta.pair_total = pair_total; ta.pair_capacity = (uint32_t)total;
    // (a sixteenth of the average tile's pairs x rows; `total` is the room given to the pairs when their number is not known yet)
    ta.prio_quantum = dev->wave_priority && n_launch_tiles ? (uint32_t)std::max<unsigned long long>(1ull, (unsigned long long)total * 2ull / n_launch_tiles) : 0u;
    ta.n_att = (uint32_t)c.attachments.size();
    bool any_clear = false, capturing = false;
    for (size_t i = 0; i < c.attachments.size(); i++) {
        const rh_attachment_desc& a = c.attachments[i];
        rh_attach_dev& d = ta.att[i];
        d.ptr = (uint8_t*)a.texture->dptr; d.pitch_px = a.texture->width; d.format = a.texture->format; d.load_op = a.load_op;
        d.clear_rgba8 = unorm8(a.clear[0]) | (unorm8(a.clear[1]) << 8) | (unorm8(a.clear[2]) << 16) | (unorm8(a.clear[3]) << 24);
        for (int k = 0; k < 4; k++) d.clear_f[k] = (float)a.clear[k];
        if (a.resolve_target) { d.resolve_ptr = (uint8_t*)a.resolve_target->dptr; d.resolve_pitch_px = a.resolve_target->width; d.resolve_format = a.resolve_target->format; }
        if (fused && a.texture == fused->tex && !capturing) {
            capturing = true;
            d.cap_ptr = (uint8_t*)fused->buf->dptr; d.cap_pitch = fused->bytes_per_row;
            d.cap_x0 = fused->ox; d.cap_y0 = fused->oy; d.cap_w = fused->width; d.cap_h = fused->height;
        }
        any_clear = any_clear || a.load_op == RH_LOAD_OP_CLEAR;
    }
    ta.src_factor = pl->targets[0].src_factor; ta.dst_factor = pl->targets[0].dst_factor;
    // pick the specialised kernel body when the pipeline state allows it (the results are identical either way)
    {   uint32_t blend_class = 0, tex_class = 0;
        if (ta.src_factor == RH_BLEND_ONE && ta.dst_factor == RH_BLEND_ONE_MINUS_SRC_ALPHA) blend_class = 1;
        else if (ta.src_factor == RH_BLEND_ONE && ta.dst_factor == RH_BLEND_ONE) blend_class = 2;
        bool all_rgba8 = true, any_linear = false, any_nearest = false;
        for (const SlotRef& sl : pl->slots) {
            // (the specialised bodies compute texel offsets in f32: exact up to 16 MiB of texels)
            if (sl.kind == RH_SLOT_TEXTURE && (sl.tex->format != RH_FORMAT_RGBA_U8 || sl.tex->width < 2 || sl.tex->height < 2 || sl.tex->bytes > ((size_t)1 << 24))) all_rgba8 = false;
            if (sl.kind == RH_SLOT_SAMPLER) { if (sl.filter == RH_FILTER_LINEAR) any_linear = true; else any_nearest = true; }
        }
        if (all_rgba8 && !(any_linear && any_nearest)) tex_class = any_nearest ? 2 : 1;
        // The specialised bilinear body can read its textures through footprint copies regrouped by phase
        // (rh_slot_dev::poly): made here, on the pass's stream, when the texels may have changed since they were made.
        // (Item offsets are computed in f32, exact below 2^24 bytes: 2^20 texels of 16-byte items, 2^19 of 32-byte ones.)
        const uint32_t item_bytes = pl->fs->info.poly_item_bytes;
        // (48: both forms, the 32-byte copies first and the 16-byte ones behind them in one block; the kernel picks per primitive)
        const size_t poly_max_bytes = item_bytes >= 32u ? ((size_t)2 << 20) : ((size_t)4 << 20);
        bool all_poly = blend_class && tex_class == 1 && !dev->general_kernels && pl->samples == 1 && total > 0 && (item_bytes == 16u || item_bytes == 32u || item_bytes == 48u);
        for (const SlotRef& sl : pl->slots)
            if (sl.kind == RH_SLOT_TEXTURE && (sl.tex->width % 32u != 0 || sl.tex->bytes > poly_max_bytes ||
                                               (no_alloc && !(sl.tex->poly && sl.tex->poly_item_bytes == item_bytes)))) all_poly = false;
        if (all_poly) {
            for (size_t i = 0; i < pl->slots.size(); i++) {
                rh_texture* t = pl->slots[i].tex;
                if (!t || pl->slots[i].kind != RH_SLOT_TEXTURE) continue;
                if (t->poly && t->poly_item_bytes != item_bytes) {
                    // (copies made for a code object that reads the other form: freed behind everything queued that reads them)
                    device_free(dev, t->bytes / 4 * t->poly_item_bytes * RH_POLY_COPIES, t->poly); t->poly = nullptr;
                }
                if (!t->poly) {         // (no_alloc: every copy exists, see all_poly)
                    if (device_alloc(dev, t->bytes / 4 * item_bytes * RH_POLY_COPIES, &t->poly)) return 1;
                    t->poly_item_bytes = item_bytes; t->poly_valid = false;
                }
                if (!t->poly_valid || t->poly_always) {
                    rhk::polyphase_build(t->dptr, t->poly, t->width, t->height, t->layers, item_bytes == 48u ? 32u : item_bytes, sm);
                    if (item_bytes == 48u)
                        rhk::polyphase_build(t->dptr, (uint8_t*)t->poly + t->bytes / 4 * 32u * RH_POLY_COPIES, t->width, t->height, t->layers, 16u, sm);
                    RH_HIP(hipGetLastError());
                    t->poly_valid = true;
                }
                if (!ta.poly_log2w) { ta.poly_log2w = bits_for(t->width + 1) - 1; ta.poly_pow2 = (t->width & (t->width - 1)) == 0 ? 1u : 0u; }
                ta.p.slots[i].poly = t->poly;
                ta.p.slots[i].size_bytes = (uint32_t)(t->bytes / 4 * (item_bytes == 48u ? 32u : item_bytes));
            }
            tex_class = 3;
        }
        ta.variant = (blend_class && tex_class && !dev->general_kernels) ? 1 + (blend_class - 1) * 3 + (tex_class - 1) : 0;
    }
    if (!no_alloc && timing_mark(dev)) return 1;
    // (what the binning of the next pass may run beside: everything from here to the next thing queued on the main stream)
    if (dbg_sync(sm, "main stream before the tile kernel")) return 1;
    RH_HIP(hipEventRecord(dev->before_tile, sm));
    dev->before_tile_valid = true;
    dev->main_epoch++;
    if (fused) fused->buf->main_write_epoch = dev->main_epoch;          // (the tile kernel writes it)
    // How many workgroups the tile kernel `fn` is launched with, and which tiles they take whole, in halves or in quarters.
    auto workgroups_for = [&](hipFunction_t fn) -> uint32_t {
        uint32_t n_groups = n_launch_tiles;
        ta.split2_first = ta.split3_first = ta.split4_first = ta.split8_first = n_launch_tiles;
        if (dev->tail_split) {
            // The GPU holds `room` workgroups of this kernel at once.  The tiles beyond the last full round of them would
            // run a few to a SIMD, each no faster than one wave's dependent chain: cut into bands they fill the SIMDs
            // and end sooner (identical results: a pixel belongs to one band, which walks the same records in order).
            uint32_t room = dev->room_override;
            for (int i = 0; i < dev->n_room && !room; i++) if (dev->room[i].fn == fn) room = dev->room[i].groups;
            if (!room) {
                int per_cu = 0;
                if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, RH_TILE_THREADS, 0) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); per_cu = 16; }
                room = (uint32_t)per_cu * std::max(1u, dev->compute_units);
                if (dev->n_room < 32) dev->room[dev->n_room++] = {fn, room};
            }
            const uint32_t tail = n_launch_tiles % room, whole = n_launch_tiles - tail;
            // the finest bands of which the whole tail fits the round (a last round more than half full is left
            // alone); in the mixed mode, where those are halves, quarters for as many of its last tiles as there is
            // room left (measured: quarters mixed with eighths are slower than quarters alone)
            const uint32_t cut = tail * 8u <= room ? 8u : tail * 4u <= room ? 4u : tail * 2u <= room ? 2u : 1u;
            if (dev->tail_split >= 3 && cut == 2u && tail * 3u <= room) {
                // thirds for the whole tail: every place of the last round then gets about the same work
                ta.split2_first = ta.split3_first = whole;
                n_groups = whole + tail * 3u;
            } else if (cut > 1u) {
                const uint32_t finer = (dev->tail_split >= 2 && cut == 2u) ? (room - tail * cut) / cut : 0u;    // (< tail)
                const uint32_t coarse_first = whole, finer_first = n_launch_tiles - finer;
                ta.split2_first = cut == 2u ? coarse_first : whole;
                ta.split4_first = cut == 2u ? finer_first : (cut == 4u ? coarse_first : whole);
                ta.split8_first = cut == 2u ? n_launch_tiles : (cut == 4u ? finer_first : coarse_first);
                ta.split3_first = ta.split4_first;
                n_groups = whole + (tail - finer) * cut + finer * cut * 2u;
            }
        }
        return n_groups;
    };
    if (pl->samples > 1) {
        // every tile runs: the resolve rewrites the whole target from the multisampled texture (Load included)
        const uint32_t f0 = c.attachments[0].texture->format;
        hipFunction_t fn = (f0 == RH_FORMAT_RGBA_F16 || f0 == RH_FORMAT_RGBA_F32) ? pl->fs->tile_ms_wide[ms_index - 1] : pl->fs->tile_ms[ms_index - 1];
        if (n_launch_tiles > 0 && launch_module_kernel(fn, workgroups_for(fn), RH_TILE_THREADS, &ta, sizeof ta, sm)) return 1;
    } else if ((total > 0 || any_clear || fused) && n_launch_tiles > 0) {
        bool any_float = false;
        for (const rh_attachment_desc& a : c.attachments) any_float = any_float || a.texture->format == RH_FORMAT_RGBA_F16 || a.texture->format == RH_FORMAT_RGBA_F32;
        hipFunction_t fn = any_float ? pl->fs->tile_wide[c.attachments.size() - 1] : pl->fs->tile[c.attachments.size() - 1];
        const uint32_t n_groups = workgroups_for(fn);
        if (launch_module_kernel(fn, n_groups, RH_TILE_THREADS, &ta, sizeof ta, sm)) return 1;
    }
    if (!no_alloc && timing_mark(dev)) return 1;
    if (alternating) {
        RH_HIP(hipEventRecord(dev->tile_done[set], sm));
        dev->tile_done_valid[set] = true;
        dev->scratch_flip = set ^ 1;
    }

    if (deferred) { dev->pending_clip_kernels = pl->clips; dev->pending_pass = &c; dev->pending_capacity = (uint32_t)total; dev->pending_fused = fused; }
    dev->stats.clusters = n_clusters; dev->stats.bin_entries = deferred ? 0 : total;
    dev->stats.tiles_x = p.tiles_x; dev->stats.tiles_y = p.tiles_y; dev->stats.tile_w = RH_TILE_W; dev->stats.tile_h = RH_TILE_H;
    return 0;
}

// Looks at the pair count of the pass that was queued without waiting for it (before anything else is queued behind
// it): normally it fitted and there is nothing to do; otherwise its kernels have written nothing, and the pass runs
// again in the waiting form.
void release_kept(rh_device* dev) {
    if (dev->kept_cbs.empty()) return;
    std::vector<rh_cmdbuf*> kept; kept.swap(dev->kept_cbs);
    for (rh_cmdbuf* cb : kept) rh_cmdbuf_release(cb);
}

// `may_rerun` false (rh_submit_static, which must not allocate): a pass that has to run again is an error, and the
// release of its command buffer -- which may hand memory back -- is left to the next call that may.
int resolve_pending(rh_device* dev, bool blocking, bool may_rerun) {
    if (!dev->pending_pass) return 0;
    if (!blocking && hipEventQuery(dev->counted) == hipErrorNotReady) return 0;
    const PassCmd& c = *static_cast<const PassCmd*>(dev->pending_pass);
    const CopyCmd* const fused = static_cast<const CopyCmd*>(dev->pending_fused);
    const uint64_t capacity = dev->pending_capacity;
    rh_cmdbuf* const kept = dev->pending_cb;
    dev->pending_pass = nullptr; dev->pending_cb = nullptr;
    int rc = 0;
    RH_TRACE("resolve_pending: waiting for the pair count");
    if (trace_waits()) {
        // (diagnostic: after five seconds without the count, say what every stream and event of the device is doing)
        const auto t0 = std::chrono::steady_clock::now();
        while (hipEventQuery(dev->counted) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                RH_TRACE("stuck: main stream %d, bin stream %d, copy stream %d; before_tile %d (valid %d), tile_done %d/%d (valid %d/%d), setup_done %d, h_total %llu %llu",
                         (int)hipStreamQuery(dev->stream), dev->bin_stream ? (int)hipStreamQuery(dev->bin_stream) : -1, dev->copy_stream ? (int)hipStreamQuery(dev->copy_stream) : -1,
                         (int)hipEventQuery(dev->before_tile), (int)dev->before_tile_valid, (int)hipEventQuery(dev->tile_done[0]), (int)hipEventQuery(dev->tile_done[1]),
                         (int)dev->tile_done_valid[0], (int)dev->tile_done_valid[1], (int)hipEventQuery(dev->setup_done),
                         (unsigned long long)dev->h_total[0], (unsigned long long)dev->h_total[1]);
                // (with RH_OPTION_TIMING on: the marks of the last passes -- start, before set-up, after set-up, before
                // the tile kernel, its end -- say which launch has not finished)
                const size_t n_ev = dev->timing_events.size();
                for (size_t i = n_ev >= 15 ? n_ev - 15 : 0; i < n_ev; i++)
                    RH_TRACE("  pass %zu mark %zu: %s", i / 5, i % 5, !dev->timing_events[i] ? "not recorded" : hipEventQuery(dev->timing_events[i]) == hipSuccess ? "reached" : "not reached");
                break;
            }
        }
    }
    hipError_t e = hipEventSynchronize(dev->counted);
    RH_TRACE("resolve_pending: count there");
    const uint64_t total = dev->h_total[0];
    if (e != hipSuccess) rc = fail("render pass: waiting for the pair count failed: %s", hipGetErrorString(e));
    else if (dev->h_total[1]) { dev->h_total[1] = 0; rc = fail("render pass: a binning workgroup gave up waiting for its predecessors; the results of the pass are undefined"); }
    else if (!may_rerun && ((total >= RH_WANTS_CLIP && !c.pipeline->clips) || (total < 0x7fffffffull && total > capacity))) {
        // (left pending: the next call that may allocate runs it again)
        dev->pending_pass = &c; dev->pending_cb = kept;
        return fail("rh_submit_static: a render pass of an earlier rh_submit has to run again (its (tile, triangle pair) array was too small, "
                    "or it met its pipeline's first clipped triangle), which allocates; call rh_device_synchronize first");
    }
    else if (total >= RH_WANTS_CLIP && !c.pipeline->clips) {
        // a triangle crosses a clip plane (nothing was written): the pass runs again, now and from now on with the kernels that clip
        c.pipeline->clips = true;
        for (int k = 0; k < 5 && dev->timing && !dev->timing_events.empty(); k++) { if (dev->timing_events.back()) dev->event_pool.push_back(dev->timing_events.back()); dev->timing_events.pop_back(); }
        dev->stats.reruns++;
        rc = execute_pass(dev, c, false, dev->grown, false, fused);
    }
    else if (total >= 0x7fffffffull) rc = fail("render pass: %llu (tile, cluster) pairs exceed the 31-bit pair index", (unsigned long long)total);
    else {
        dev->stats.bin_entries = total;
        if (c.pipeline->clips && dev->pending_clip_kernels) note_clipping(c.pipeline, (uint32_t)dev->h_total[2] != 0u);
        if (total <= capacity) dev->pair_guess = std::max<uint64_t>(dev->pair_guess, std::min<uint64_t>(total + total / 4 + 4096, 0x7ffffffeull));
        else {
            // the attempt that wrote nothing is not a pass of its own: its timing marks go back to the pool
            for (int k = 0; k < 5 && dev->timing && !dev->timing_events.empty(); k++) { if (dev->timing_events.back()) dev->event_pool.push_back(dev->timing_events.back()); dev->timing_events.pop_back(); }
            dev->stats.reruns++;
            rc = execute_pass(dev, c, false, dev->grown, false, fused);
        }
    }
    if (kept) { if (may_rerun) rh_cmdbuf_release(kept); else dev->kept_cbs.push_back(kept); }
    return rc;
}

int execute_copy(rh_device* dev, const CopyCmd& c) {
    if (c.buf->upload_pending) { RH_HIP(hipStreamWaitEvent(dev->stream, c.buf->upload_done, 0)); c.buf->upload_pending = false; }
    c.buf->main_write_epoch = dev->main_epoch;
    rhk::copy_texture_to_buffer((const uint8_t*)c.tex->dptr, c.tex->width * c.tex->bpt, c.ox, c.oy, c.tex->bpt, c.width, c.height,
                                (uint8_t*)c.buf->dptr, c.bytes_per_row, dev->stream);
    RH_HIP(hipGetLastError());
    return 0;
}

} // namespace

extern "C" {

int rh_submit(rh_device* dev, rh_cmdbuf* const* list, size_t n) {
    if (!dev || (!list && n)) return fail("rh_submit: bad argument");
    int rc = enter(dev);
    size_t last = n;                            // the last command buffer of the list
    for (size_t i = 0; i < n; i++) if (list[i]) last = i;
    for (size_t i = 0; i < n; i++) {
        rh_cmdbuf* cb = list[i];
        if (!cb) continue;
        if (!rc) {
            for (size_t k = 0; k < cb->cmds.size() && !rc; k++) {
                const Cmd& c = cb->cmds[k];
                rc = resolve_pending(dev);
                if (rc) break;
                if (c.kind != 0) { rc = execute_copy(dev, c.copy); continue; }
                const CopyCmd* fused = (k + 1 < cb->cmds.size() && cb->cmds[k + 1].kind == 1 && fusable_copy(dev, c.pass, cb->cmds[k + 1].copy)) ? &cb->cmds[k + 1].copy : nullptr;
                rc = execute_pass(dev, c.pass, true, dev->grown, false, fused);
                if (fused) k++;
            }
            // A pass still awaiting its count at the end of the submit stays that way: the host does not wait here.  Its
            // command buffer is kept until the count has been looked at, by the next call that queues work or reads results.
            if (!rc && dev->pending_pass && i == last) { dev->pending_cb = cb; continue; }
            if (!rc) rc = resolve_pending(dev);
            dev->pending_pass = nullptr;
        }
        rh_cmdbuf_release(cb);
    }
    if (!rc) rc = dev->eager_count ? rh_poll(dev, 0) : poll_maps(dev, 0);
    return rc;
}

int rh_submit_static(rh_device* dev, rh_cmdbuf* const* list, size_t n) {
    if (!dev || (!list && n)) return fail("rh_submit_static: bad argument");
    if (!dev->held.rec.ptr) return fail("rh_submit_static: the static scratch array of this device is not available");
    if (set_device(dev) || resolve_pending(dev, true, false)) return 1;
    hipEvent_t const tail = dev->ordinal < RH_MAX_ORDINALS ? g_static_tail[dev->ordinal] : nullptr;
    if (!tail) return fail("rh_submit_static: GPU %d has no event for its static scratch array", dev->ordinal);
    // the kernels of the previous call on this GPU (from any rh_device) may still be reading the array
    RH_HIP(hipStreamWaitEvent(dev->stream, tail, 0));
    struct RecordTail { hipEvent_t e; hipStream_t s; ~RecordTail() { (void)hipEventRecord(e, s); } } record_tail{tail, dev->stream};
    for (size_t i = 0; i < n; i++) {
        const rh_cmdbuf* cb = list[i];
        if (!cb) continue;
        for (size_t k = 0; k < cb->cmds.size(); k++) {
            const Cmd& c = cb->cmds[k];
            // every pass waits for its pair count (that is a synchronisation, not an allocation) and works in the held areas
            if (c.kind != 0) { if (execute_copy(dev, c.copy)) return 1; continue; }
            const CopyCmd* fused = (k + 1 < cb->cmds.size() && cb->cmds[k + 1].kind == 1 && fusable_copy(dev, c.pass, cb->cmds[k + 1].copy)) ? &cb->cmds[k + 1].copy : nullptr;
            if (execute_pass(dev, c.pass, false, dev->held, true, fused)) return 1;
            if (fused) k++;
        }
    }
    return 0;
}

int rh_device_drain_timings(rh_device* dev, rh_timings* out) {
    if (!dev || !out) return fail("rh_device_drain_timings: bad argument");
    if (enter(dev)) return 1;
    out->passes = 0; out->bin_ms = 0.0; out->setup_ms = 0.0; out->tile_ms = 0.0;
    const size_t n = dev->timing_events.size() / 5;
    for (size_t i = 0; i < n; i++) {
        // start, before set-up, after set-up (the three on the stream that did the binning), before and after the tile kernel
        hipEvent_t e0 = dev->timing_events[5 * i], e1 = dev->timing_events[5 * i + 1], e2 = dev->timing_events[5 * i + 2], e3 = dev->timing_events[5 * i + 3], e4 = dev->timing_events[5 * i + 4];
        RH_HIP(hipEventSynchronize(e4));
        float a = 0.0f, b = 0.0f, c = 0.0f;
        if (e0 && e1 && e2) {
            RH_HIP(hipEventElapsedTime(&a, e0, e1));
            RH_HIP(hipEventElapsedTime(&b, e1, e2));
        }
        RH_HIP(hipEventElapsedTime(&c, e3, e4));
        out->passes++; out->bin_ms += a; out->setup_ms += b; out->tile_ms += c;
    }
    for (hipEvent_t e : dev->timing_events) if (e) dev->event_pool.push_back(e);
    dev->timing_events.clear();
    return 0;
}

int rh_device_last_pass_stats(rh_device* dev, rh_pass_stats* out) {
    if (!dev || !out) return fail("rh_device_last_pass_stats: bad argument");
    if (enter(dev)) return 1;                   // (the pair count of a pass still awaiting it)
    *out = dev->stats;
    return 0;
}

} // extern "C"

// ---- capture codec --------------------------------------------------------------------------------------------------------
#include "rh_codec.h"
#include "rh_png.h"
#include <deque>

// Compressed bytes wait for the reader in segments: either a small owned vector (frame header, end mark) or a slice
// of a pinned round buffer that the device-to-host copies filled.  A round buffer is recycled when its last slice has
// been consumed.
struct rh_compressor {
    rh_device* dev = nullptr;
    uint32_t n_streams = 0;
    hipStream_t stream = nullptr;                    // compression runs beside rendering
    hipStream_t copy_stream = nullptr;               // brings a finished round's bytes back while the next round is compressed
    hipEvent_t frames_ready = nullptr;               // recorded on the device's stream when a round is submitted
    rh_lz4_stream_state* d_states = nullptr;
    struct Packet { std::vector<uint8_t> meta; rh_buffer* image = nullptr; };
    std::vector<std::deque<Packet>> queue;           // per stream: not yet submitted
    // The copies that fill a buffer are recorded in pieces of COPY_PIECE streams, an event behind each piece, so that
    // the reader writes the first streams to their files while the last ones are still crossing PCIe (`pending`:
    // nobody has waited for the piece yet).
    static constexpr uint32_t COPY_PIECE = 64;
    struct RoundBuf { void* h = nullptr; size_t cap = 0; size_t live = 0; std::vector<hipEvent_t> ready; std::vector<char> pending; };
    std::vector<RoundBuf> bufs;
    hipEvent_t out_copied[2] = { nullptr, nullptr };    // the last copy out of each output area (a round that writes the area waits for it)
    struct Segment { const uint8_t* ptr = nullptr; size_t len = 0; int buf = -1; uint32_t piece = 0; std::vector<uint8_t> own; };
    std::vector<std::deque<Segment>> ready;          // per stream: compressed bytes not yet consumed
    // the round in flight
    struct Round {
        bool active = false;                         // its kernel has been launched and its bytes not yet collected
        bool waited = false;                         // its kernel has finished: `lens` is valid, the images are released
        int out_buf = 0;                             // which of the two output areas it writes
        std::vector<uint32_t> job_stream;
        std::vector<size_t> out_off, out_cap;
        std::vector<unsigned long long> lens;
        std::vector<rh_buffer*> images;
    } round;
    Scratch d_jobs, d_stage, d_out[2], d_lens;       // two output areas: round k+1 is compressed while round k's bytes are copied out
    int out_sel = 0;
    unsigned long long* h_lens = nullptr; size_t h_lens_cap = 0;
    hipEvent_t k0 = nullptr, k1 = nullptr;           // around the kernel of the round in flight
    rh_compressor_stats stats{};                     // since the last rh_compressor_drain_stats
    uint64_t round_bytes_in = 0;
};

namespace {

size_t lz4f_bound(size_t n) { return n + 4 * (n / 65536 + 1); }      // every block stored + one header each
size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }

void push_owned(rh_compressor* c, uint32_t stream, const uint8_t* p, size_t n) {
    rh_compressor::Segment sg; sg.own.assign(p, p + n);
    c->ready[stream].push_back(std::move(sg));
    rh_compressor::Segment& back = c->ready[stream].back();
    back.ptr = back.own.data(); back.len = n;
}

// Starts compressing the first queued packet of every stream (one workgroup per stream).  Returns at once; the result
// is collected by compressor_complete_round.
int compressor_submit_round(rh_compressor* c) {
    rh_device* dev = c->dev;
    rh_compressor::Round& R = c->round;
    std::vector<rh_lz4_job> jobs;
    std::vector<size_t> stage_off;
    std::vector<uint8_t> stage;
    size_t out_total = 0;
    R.job_stream.clear(); R.out_off.clear(); R.out_cap.clear(); R.images.clear();
    for (uint32_t st = 0; st < c->n_streams; st++) {
        if (c->queue[st].empty()) continue;
        const rh_compressor::Packet& pk = c->queue[st].front();
        const uint64_t image_len = pk.image ? pk.image->size : 0;
        const uint64_t packet_len = 16 + pk.meta.size() + image_len;
        const uint64_t meta_len = pk.meta.size();
        // staged small writes: [packet_len BE][pad to 16][meta_len BE][pad to 16][metadata][pad]
        const size_t base = stage.size();
        stage.resize(base + 32 + align16(pk.meta.size()), 0);
        for (int i = 0; i < 8; i++) { stage[base + i] = (uint8_t)(packet_len >> (56 - 8 * i)); stage[base + 16 + i] = (uint8_t)(meta_len >> (56 - 8 * i)); }
        memcpy(stage.data() + base + 32, pk.meta.data(), pk.meta.size());
        stage_off.push_back(base);
        const size_t cap = align16(lz4f_bound(8) * 2 + lz4f_bound(pk.meta.size()) + lz4f_bound(image_len));
        R.out_off.push_back(out_total); R.out_cap.push_back(cap);
        out_total += cap;
        R.job_stream.push_back(st);
    }
    const uint32_t n = (uint32_t)R.job_stream.size();
    if (n == 0) return 0;
    R.out_buf = c->out_sel; c->out_sel ^= 1;
    Scratch& d_out = c->d_out[R.out_buf];
    if (c->d_jobs.ensure(n * sizeof(rh_lz4_job)) || c->d_stage.ensure(stage.size()) || d_out.ensure(out_total) || c->d_lens.ensure(n * 8)) return 1;
    if (n > c->h_lens_cap) {
        if (c->h_lens) (void)hipHostFree(c->h_lens);
        c->h_lens = nullptr; c->h_lens_cap = 0;
        RH_HIP(hipHostMalloc((void**)&c->h_lens, (size_t)n * 2 * 8, hipHostMallocDefault));
        c->h_lens_cap = (size_t)n * 2;
    }
    jobs.resize(n);
    for (uint32_t j = 0; j < n; j++) {
        rh_compressor::Packet& pk = c->queue[R.job_stream[j]].front();
        rh_lz4_job& jb = jobs[j];
        memset(&jb, 0, sizeof jb);
        jb.stream = R.job_stream[j];
        const uint8_t* st = c->d_stage.as<uint8_t>() + stage_off[j];
        jb.write_ptr[0] = st; jb.write_len[0] = 8;
        jb.write_ptr[1] = st + 16; jb.write_len[1] = 8;
        jb.write_ptr[2] = st + 32; jb.write_len[2] = pk.meta.size();
        jb.n_writes = 3;
        if (pk.image) { jb.write_ptr[3] = (const uint8_t*)pk.image->dptr; jb.write_len[3] = pk.image->size; jb.n_writes = 4; }
        jb.out = d_out.as<uint8_t>() + R.out_off[j];
        jb.out_len = c->d_lens.as<unsigned long long>() + j;
        R.images.push_back(pk.image);                 // stays retained until the round completes
        c->queue[R.job_stream[j]].pop_front();
    }
    // the captured frames are produced on the device's stream
    RH_HIP(hipEventRecord(c->frames_ready, dev->stream));
    RH_HIP(hipStreamWaitEvent(c->stream, c->frames_ready, 0));
    RH_HIP(hipMemcpyAsync(c->d_stage.ptr, stage.data(), stage.size(), hipMemcpyHostToDevice, c->stream));
    RH_HIP(hipMemcpyAsync(c->d_jobs.ptr, jobs.data(), n * sizeof(rh_lz4_job), hipMemcpyHostToDevice, c->stream));
    RH_TRACE("compressor: waiting for the round's frames and job records");
    RH_HIP(hipStreamSynchronize(c->stream));        // the pageable staging vectors go out of scope below
    RH_TRACE("compressor: job records there");
    c->round_bytes_in = 0;
    for (const rh_lz4_job& jb : jobs) for (uint32_t w = 0; w < jb.n_writes; w++) c->round_bytes_in += jb.write_len[w];
    if (c->out_copied[R.out_buf]) RH_HIP(hipStreamWaitEvent(c->stream, c->out_copied[R.out_buf], 0));     // the area's previous bytes have left
    RH_HIP(hipEventRecord(c->k0, c->stream));
    int rc = rhk::lz4f_compress_launch(c->d_jobs.as<rh_lz4_job>(), n, c->d_states, c->stream);
    if (rc) return fail("rh_compressor: kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    RH_HIP(hipEventRecord(c->k1, c->stream));
    RH_HIP(hipMemcpyAsync(c->h_lens, c->d_lens.ptr, n * 8, hipMemcpyDeviceToHost, c->stream));
    R.active = true; R.waited = false;
    return 0;
}

// Waits for the kernel of the round in flight: its compressed sizes are known and its capture buffers are free again.
int compressor_wait_round(rh_compressor* c) {
    rh_compressor::Round& R = c->round;
    if (!R.active || R.waited) return 0;
    const uint32_t n = (uint32_t)R.job_stream.size();
    RH_TRACE("compressor: waiting for the round in flight");
    RH_HIP(hipStreamSynchronize(c->stream));
    RH_TRACE("compressor: round done");
    R.waited = true;
    R.lens.assign(c->h_lens, c->h_lens + n);
    {   float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->k0, c->k1) == hipSuccess) c->stats.kernel_ms += ms;
        c->stats.rounds++; c->stats.jobs += n; c->stats.bytes_in += c->round_bytes_in;
        for (uint32_t j = 0; j < n; j++) c->stats.bytes_out += R.lens[j];
    }
    for (uint32_t j = 0; j < n; j++) {
        if (R.images[j]) rh_buffer_release(R.images[j]);
        if (R.lens[j] > R.out_cap[j]) return fail("rh_compressor: stream %u overflowed its output area", R.job_stream[j]);
    }
    R.images.clear();
    return 0;
}

// Brings exactly the compressed bytes of a finished round to the host (only they cross PCIe), on the copy stream, so
// that a round launched meanwhile keeps compressing.
int compressor_collect_round(rh_compressor* c, rh_compressor::Round& R) {
    if (!R.active) return 0;
    R.active = false;
    const uint32_t n = (uint32_t)R.job_stream.size();
    if (n == 0) return 0;
    size_t total = 0;
    for (uint32_t j = 0; j < n; j++) total += (size_t)R.lens[j];
    int bi = -1;
    for (size_t i = 0; i < c->bufs.size(); i++) if (c->bufs[i].live == 0 && c->bufs[i].cap >= total) { bi = (int)i; break; }
    if (bi < 0) {
        for (size_t i = 0; i < c->bufs.size(); i++) if (c->bufs[i].live == 0) { bi = (int)i; break; }
        if (bi < 0) { c->bufs.emplace_back(); bi = (int)c->bufs.size() - 1; }
        rh_compressor::RoundBuf& b = c->bufs[bi];
        if (b.h) (void)hipHostFree(b.h);
        b.h = nullptr; b.cap = 0;
        const size_t want = total + total / 16 + 4096;        // (rounds of one scene differ by a few per cent)
        RH_HIP(hipHostMalloc(&b.h, want, hipHostMallocDefault));
        b.cap = want;
    }
    rh_compressor::RoundBuf& b = c->bufs[bi];
    const uint32_t pieces = (n + rh_compressor::COPY_PIECE - 1) / rh_compressor::COPY_PIECE;
    while (b.ready.size() < pieces) {
        hipEvent_t e = nullptr;
        RH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b.ready.push_back(e);
    }
    b.pending.assign(b.ready.size(), 0);
    // The copies run beside whatever the caller does next; whoever reads the bytes waits for them (compressor_wait_bytes).
    size_t off = 0;
    for (uint32_t j = 0; j < n; j++) {
        const size_t len = (size_t)R.lens[j];
        if (len) RH_HIP(hipMemcpyAsync((uint8_t*)b.h + off, c->d_out[R.out_buf].as<uint8_t>() + R.out_off[j], len, hipMemcpyDeviceToHost, c->copy_stream));
        off += len;
        if ((j + 1) % rh_compressor::COPY_PIECE == 0 || j + 1 == n) {
            const uint32_t piece = j / rh_compressor::COPY_PIECE;
            RH_HIP(hipEventRecord(b.ready[piece], c->copy_stream));
            b.pending[piece] = 1;
        }
    }
    c->out_copied[R.out_buf] = b.ready[pieces - 1];
    off = 0;
    for (uint32_t j = 0; j < n; j++) {
        const size_t len = (size_t)R.lens[j];
        if (len) {
            rh_compressor::Segment sg; sg.ptr = (const uint8_t*)b.h + off; sg.len = len; sg.buf = bi; sg.piece = j / rh_compressor::COPY_PIECE;
            c->ready[R.job_stream[j]].push_back(std::move(sg));
            b.live++;
        }
        off += len;
    }
    return 0;
}

// Before the bytes of a segment are read on the host: the copies that fill its buffer have finished.
int compressor_wait_bytes(rh_compressor* c, const rh_compressor::Segment& sg) {
    if (sg.buf < 0 || !c->bufs[sg.buf].pending[sg.piece]) return 0;
    RH_HIP(hipSetDevice(c->dev->ordinal));
    RH_HIP(hipEventSynchronize(c->bufs[sg.buf].ready[sg.piece]));
    // (the pieces are recorded in order on one stream: the earlier ones have arrived as well)
    for (uint32_t i = 0; i <= sg.piece; i++) c->bufs[sg.buf].pending[i] = 0;
    return 0;
}

int compressor_complete_round(rh_compressor* c) {
    if (compressor_wait_round(c)) return 1;
    return compressor_collect_round(c, c->round);
}

bool compressor_queue_empty(const rh_compressor* c) {
    for (const auto& q : c->queue) if (!q.empty()) return false;
    return true;
}

} // namespace

extern "C" {

int rh_compressor_create(rh_device* dev, uint32_t n_streams, int level, rh_compressor** out) {
    if (!dev || !out || n_streams == 0) return fail("rh_compressor_create: bad argument");
    if (level < 0 || level > 2) return fail("rh_compressor_create: compression level %d is not implemented on the GPU (0..2 = LZ4 fast)", level);
    if (set_device(dev)) return 1;
    rh_compressor* c = new rh_compressor();
    c->dev = dev; c->n_streams = n_streams;
    hipError_t e = hipMalloc((void**)&c->d_states, (size_t)n_streams * sizeof(rh_lz4_stream_state));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_states, 0, (size_t)n_streams * sizeof(rh_lz4_stream_state), dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    // A stream of another priority gets a hardware queue of its own: with the few queues a process has, a second
    // stream of the same priority can land on the queue of the compression stream, behind a kernel that runs a second.
    if (e == hipSuccess) {
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, greatest);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->frames_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->k0);
    if (e == hipSuccess) e = hipEventCreate(&c->k1);
    if (e != hipSuccess) {
        if (c->k0) (void)hipEventDestroy(c->k0);
        if (c->k1) (void)hipEventDestroy(c->k1);
        if (c->frames_ready) (void)hipEventDestroy(c->frames_ready);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        if (c->d_states) (void)hipFree(c->d_states);
        delete c;
        return fail("rh_compressor_create: %s", hipGetErrorString(e));
    }
    c->queue.resize(n_streams); c->ready.resize(n_streams);
    // LZ4F_compressBegin: magic, FLG (version 01, block-linked), BD (64 KiB blocks), header checksum
    static const uint8_t header[7] = { 0x04, 0x22, 0x4d, 0x18, 0x40, 0x40, 0xc0 };
    for (uint32_t i = 0; i < n_streams; i++) push_owned(c, i, header, 7);
    device_retain(dev);
    dev->compressors++;
    *out = c;
    return 0;
}

void rh_compressor_destroy(rh_compressor* c) {
    if (!c) return;
    rh_device* dev = c->dev;
    (void)hipSetDevice(dev->ordinal);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(dev->stream);
    for (auto& q : c->queue) for (auto& pk : q) if (pk.image) rh_buffer_release(pk.image);
    for (rh_buffer* b : c->round.images) if (b) rh_buffer_release(b);
    (void)hipStreamSynchronize(c->copy_stream);
    Scratch* all[] = { &c->d_jobs, &c->d_stage, &c->d_out[0], &c->d_out[1], &c->d_lens };
    for (Scratch* s : all) s->release();
    for (auto& b : c->bufs) { if (b.h) (void)hipHostFree(b.h); for (hipEvent_t e : b.ready) (void)hipEventDestroy(e); }
    if (c->h_lens) (void)hipHostFree(c->h_lens);
    if (c->d_states) (void)hipFree(c->d_states);
    (void)hipEventDestroy(c->frames_ready);
    (void)hipEventDestroy(c->k0); (void)hipEventDestroy(c->k1);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->copy_stream);
    delete c;
    dev->compressors--;
    device_release(dev);
}

int rh_compressor_write_packet(rh_compressor* c, uint32_t stream, const void* metadata, size_t metadata_len, rh_buffer* image) {
    if (!c || (!metadata && metadata_len)) return fail("rh_compressor_write_packet: bad argument");
    if (stream >= c->n_streams) return fail("rh_compressor_write_packet: stream %u out of range (%u streams)", stream, c->n_streams);
    if (metadata_len == 0) return fail("rh_compressor_write_packet: empty metadata");
    rh_compressor::Packet pk;
    pk.meta.assign((const uint8_t*)metadata, (const uint8_t*)metadata + metadata_len);
    pk.image = image;
    if (image) rh_buffer_retain(image);
    c->queue[stream].push_back(std::move(pk));
    return 0;
}

int rh_compressor_submit(rh_compressor* c) {
    if (!c) return fail("rh_compressor_submit: null compressor");
    if (enter(c->dev)) return 1;
    for (;;) {
        if (compressor_wait_round(c)) return 1;
        if (compressor_queue_empty(c)) return compressor_collect_round(c, c->round);
        // the next round starts before the finished one's bytes are copied out (it writes the other output area)
        rh_compressor::Round prev = std::move(c->round);
        c->round = rh_compressor::Round();
        if (compressor_submit_round(c)) return 1;
        if (compressor_collect_round(c, prev)) return 1;
        if (compressor_queue_empty(c)) return 0;      // the last round stays in flight
    }
}

int rh_compressor_wait(rh_compressor* c) {
    if (!c) return fail("rh_compressor_wait: null compressor");
    if (set_device(c->dev)) return 1;
    return compressor_complete_round(c);
}

int rh_compressor_flush(rh_compressor* c) {
    if (!c) return fail("rh_compressor_flush: null compressor");
    if (rh_compressor_submit(c)) return 1;
    return compressor_complete_round(c);
}

size_t rh_compressor_pending(rh_compressor* c, uint32_t stream) {
    if (!c || stream >= c->n_streams) return 0;
    size_t n = 0;
    for (const auto& sg : c->ready[stream]) n += sg.len;
    return n;
}

int rh_compressor_peek(rh_compressor* c, uint32_t stream, const uint8_t** ptr, size_t* len) {
    if (!c || !ptr || !len) return fail("rh_compressor_peek: bad argument");
    if (stream >= c->n_streams) return fail("rh_compressor_peek: stream %u out of range", stream);
    if (c->ready[stream].empty()) { *ptr = nullptr; *len = 0; return 0; }
    if (compressor_wait_bytes(c, c->ready[stream].front())) return 1;
    *ptr = c->ready[stream].front().ptr; *len = c->ready[stream].front().len;
    return 0;
}

int rh_compressor_consume(rh_compressor* c, uint32_t stream, size_t n) {
    if (!c) return fail("rh_compressor_consume: null compressor");
    if (stream >= c->n_streams) return fail("rh_compressor_consume: stream %u out of range", stream);
    while (n) {
        if (c->ready[stream].empty()) return fail("rh_compressor_consume: more bytes than are ready");
        rh_compressor::Segment& sg = c->ready[stream].front();
        const size_t k = std::min(n, sg.len);
        sg.ptr += k; sg.len -= k; n -= k;
        if (sg.len == 0) {
            if (sg.buf >= 0) c->bufs[sg.buf].live--;
            c->ready[stream].pop_front();
        }
    }
    return 0;
}

int rh_compressor_read(rh_compressor* c, uint32_t stream, void* dst, size_t cap, size_t* len) {
    if (!c || !len || (!dst && cap)) return fail("rh_compressor_read: bad argument");
    if (stream >= c->n_streams) return fail("rh_compressor_read: stream %u out of range", stream);
    size_t done = 0;
    while (done < cap && !c->ready[stream].empty()) {
        const rh_compressor::Segment& sg = c->ready[stream].front();
        if (compressor_wait_bytes(c, sg)) return 1;
        const size_t k = std::min(cap - done, sg.len);
        memcpy((uint8_t*)dst + done, sg.ptr, k);
        done += k;
        if (rh_compressor_consume(c, stream, k)) return 1;
    }
    *len = done;
    return 0;
}

int rh_compressor_drain_stats(rh_compressor* c, rh_compressor_stats* out) {
    if (!c || !out) return fail("rh_compressor_drain_stats: bad argument");
    *out = c->stats;
    c->stats = rh_compressor_stats{};
    return 0;
}

int rh_compressor_finish_stream(rh_compressor* c, uint32_t stream) {
    if (!c) return fail("rh_compressor_finish_stream: null compressor");
    if (stream >= c->n_streams) return fail("rh_compressor_finish_stream: stream %u out of range", stream);
    if (rh_compressor_flush(c)) return 1;
    static const uint8_t end_mark[4] = { 0, 0, 0, 0 };
    push_owned(c, stream, end_mark, 4);
    return 0;
}

// ---- GPU PNG encoder ------------------------------------------------------------------------------------------------------
namespace {
uint32_t png_slot_dwords(uint32_t width) { return (width * 36u + 160u) / 32u + 2u; }      // worst case: every pixel four 9-bit literals
const uint32_t PNG_HEAD = 8 + 25, PNG_TAIL = 12;                                          // signature + IHDR, IEND

uint32_t host_crc32(const uint8_t* p, size_t n) {
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0xedb88320u : c >> 1; }
    return c ^ 0xffffffffu;
}
} // namespace

size_t rh_png_bound(uint32_t width, uint32_t height) {
    return (size_t)PNG_HEAD + (size_t)height * (12u + (size_t)png_slot_dwords(width) * 4u) + 16u + PNG_TAIL;
}

int rh_png_encode(rh_device* dev, rh_buffer* image, uint32_t width, uint32_t height, uint32_t padded_bytes_per_row,
                  void* out, size_t cap, size_t* len) {
    if (!dev || !image || !out || !len) return fail("rh_png_encode: bad argument");
    if (width == 0 || height == 0) return fail("rh_png_encode: empty image");
    if ((uint64_t)width * 4u > padded_bytes_per_row || (padded_bytes_per_row & 3u)) return fail("rh_png_encode: row pitch %u does not hold %u RGBA8 pixels", padded_bytes_per_row, width);
    if ((uint64_t)padded_bytes_per_row * height > image->size) return fail("rh_png_encode: the buffer holds %zu bytes, %u rows of %u are needed", image->size, height, padded_bytes_per_row);
    if (width > (1u << 24)) return fail("rh_png_encode: image too wide");
    if (enter(dev)) return 1;
    const uint32_t slot = png_slot_dwords(width);
    const size_t out_dev = (size_t)height * (12u + (size_t)slot * 4u) + 16u;
    const size_t rows_bytes = (size_t)height * (3 * 4 + 8) + 8;
    if (dev->png_slots.ensure((size_t)height * slot * 4u) || dev->png_rows.ensure(rows_bytes) || dev->png_out.ensure(out_dev)) return 1;
    if (dev->png_host_cap < out_dev) {
        if (dev->png_host) (void)hipHostFree(dev->png_host);
        dev->png_host = nullptr; dev->png_host_cap = 0;
        RH_HIP(hipHostMalloc(&dev->png_host, out_dev + out_dev / 4, hipHostMallocDefault));
        dev->png_host_cap = out_dev + out_dev / 4;
    }
    rh_png_args a;
    memset(&a, 0, sizeof a);
    a.image = (const uint8_t*)image->dptr; a.width = width; a.height = height; a.padded_bytes_per_row = padded_bytes_per_row;
    const uint32_t stride = width * 4u + 1u;
    a.up_ok = stride <= 32768u ? 1u : 0u;
    if (a.up_ok) {
        // fixed Huffman distance code (5 bits, sent most significant bit first) followed by its extra bits
        const uint32_t d = stride - 1u;
        uint32_t code, e;
        if (d < 4u) { code = d; e = 0; }
        else { e = (31u - (uint32_t)__builtin_clz(d)) - 1u; code = 2u * e + 2u + ((d >> e) & 1u); }
        uint32_t rev = 0;
        for (int k = 0; k < 5; k++) rev |= ((code >> k) & 1u) << (4 - k);
        a.up_code = rev | ((d & ((1u << e) - 1u)) << 5);
        a.up_code_bits = 5u + e;
    }
    a.slot_dwords = slot;
    a.slots = dev->png_slots.as<uint32_t>();
    uint8_t* rows = dev->png_rows.as<uint8_t>();
    a.row_offset = (unsigned long long*)rows;
    a.total_bytes = (unsigned long long*)(rows + (size_t)height * 8);
    a.row_bytes = (uint32_t*)(rows + (size_t)height * 8 + 8);
    a.row_s1 = a.row_bytes + height;
    a.row_s2 = a.row_s1 + height;
    a.out = dev->png_out.as<uint8_t>();
    hipStream_t s = dev->stream;
    rhk::png_encode_launch(a, s);
    RH_HIP(hipGetLastError());
    RH_HIP(hipMemcpyAsync(dev->h_total, a.total_bytes, 8, hipMemcpyDeviceToHost, s));
    RH_HIP(hipStreamSynchronize(s));
    const size_t idat = (size_t)*dev->h_total;
    if (idat > out_dev) return fail("rh_png_encode: the encoder overran its output area");
    const size_t total = (size_t)PNG_HEAD + idat + PNG_TAIL;
    *len = total;
    if (total > cap) return fail("rh_png_encode: %zu bytes are needed, the output holds %zu", total, cap);
    RH_HIP(hipMemcpyAsync(dev->png_host, a.out, idat, hipMemcpyDeviceToHost, s));
    RH_HIP(hipStreamSynchronize(s));
    uint8_t* o = (uint8_t*)out;
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    memcpy(o, sig, 8);
    uint8_t ihdr[25] = { 0, 0, 0, 13, 'I', 'H', 'D', 'R' };
    for (int i = 0; i < 4; i++) { ihdr[8 + i] = (uint8_t)(width >> (24 - 8 * i)); ihdr[12 + i] = (uint8_t)(height >> (24 - 8 * i)); }
    ihdr[16] = 8; ihdr[17] = 6; ihdr[18] = 0; ihdr[19] = 0; ihdr[20] = 0;       // 8 bit, RGBA, deflate, adaptive filtering, no interlace
    const uint32_t c1 = host_crc32(ihdr + 4, 17);
    for (int i = 0; i < 4; i++) ihdr[21 + i] = (uint8_t)(c1 >> (24 - 8 * i));
    memcpy(o + 8, ihdr, 25);
    memcpy(o + PNG_HEAD, dev->png_host, idat);
    static const uint8_t iend[12] = { 0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82 };
    memcpy(o + PNG_HEAD + idat, iend, 12);
    return 0;
}

// LZ4 frame decoder (host).  Block-linked frames: matches may reach into everything decoded so far.
int rh_lz4f_decompress(const void* src_, size_t n, void* dst_, size_t cap, size_t* produced, size_t* consumed, int* complete) {
    const uint8_t* src = (const uint8_t*)src_; uint8_t* dst = (uint8_t*)dst_;
    if (!src || (!dst && cap) || !produced || !consumed || !complete) return fail("rh_lz4f_decompress: bad argument");
    *produced = 0; *consumed = 0; *complete = 0;
    auto le32 = [&](size_t at) { return (uint32_t)src[at] | ((uint32_t)src[at + 1] << 8) | ((uint32_t)src[at + 2] << 16) | ((uint32_t)src[at + 3] << 24); };
    if (n < 7 || le32(0) != 0x184D2204u) return fail("rh_lz4f_decompress: not an LZ4 frame");
    const uint8_t flg = src[4];
    if ((flg >> 6) != 1) return fail("rh_lz4f_decompress: unsupported frame version");
    const bool block_checksum = (flg >> 4) & 1, content_size = (flg >> 3) & 1, content_checksum = (flg >> 2) & 1, dict_id = flg & 1;
    size_t ip = 6 + (content_size ? 8 : 0) + (dict_id ? 4 : 0) + 1, pos = 0;
    if (n < ip) return fail("rh_lz4f_decompress: truncated frame header");
    *consumed = ip;
    for (;;) {
        if (n - ip < 4) break;
        const uint32_t bh = le32(ip);
        if (bh == 0) { ip += 4; if (content_checksum) { if (n - ip < 4) break; ip += 4; } *complete = 1; break; }
        const size_t bsz = bh & 0x7fffffffu, trailer = block_checksum ? 4 : 0;
        if (n - ip - 4 < bsz + trailer) break;                              // truncated block: stop at the last complete one
        const uint8_t* b = src + ip + 4; const uint8_t* const bend = b + bsz;
        if (bh & 0x80000000u) {
            if (cap - pos < bsz) return fail("rh_lz4f_decompress: output buffer too small");
            memcpy(dst + pos, b, bsz); pos += bsz;
        } else {
            while (b < bend) {
                const unsigned token = *b++;
                size_t lit = token >> 4;
                if (lit == 15) { unsigned x; do { if (b >= bend) return fail("rh_lz4f_decompress: malformed block"); x = *b++; lit += x; } while (x == 255); }
                if ((size_t)(bend - b) < lit) return fail("rh_lz4f_decompress: malformed block");
                if (cap - pos < lit) return fail("rh_lz4f_decompress: output buffer too small");
                memcpy(dst + pos, b, lit); pos += lit; b += lit;
                if (b >= bend) break;                                       // last sequence: literals only
                if (bend - b < 2) return fail("rh_lz4f_decompress: malformed block");
                const size_t off = (size_t)b[0] | ((size_t)b[1] << 8); b += 2;
                size_t ml = token & 15;
                if (ml == 15) { unsigned x; do { if (b >= bend) return fail("rh_lz4f_decompress: malformed block"); x = *b++; ml += x; } while (x == 255); }
                ml += 4;
                if (off == 0 || off > pos) return fail("rh_lz4f_decompress: match offset outside the decoded data");
                if (cap - pos < ml) return fail("rh_lz4f_decompress: output buffer too small");
                for (size_t i = 0; i < ml; i++) dst[pos + i] = dst[pos + i - off];
                pos += ml;
            }
        }
        ip += 4 + bsz + trailer;
        *consumed = ip; *produced = pos;
    }
    *consumed = ip; *produced = pos;
    return 0;
}

} // extern "C"

// ---- stream decoder ------------------------------------------------------------------------------------------------------------
struct rh_decoder {
    rh_device* dev = nullptr;                        // null: host decoding
    uint32_t n_streams = 0;
    struct Segment { const uint8_t* ptr = nullptr; size_t len = 0; int buf = -1; };
    struct Stream {
        std::vector<uint8_t> pending;                // fed bytes not decoded yet, from `at`
        size_t at = 0;
        bool header_done = false;
        uint32_t trailer = 0;                        // 4 with block checksums
        bool content_checksum = false;
        uint32_t max_block = 0;                      // decoded bytes of a block at most (BD byte)
        int state = RH_STREAM_RUNNING;
        uint64_t decoded = 0;
        std::deque<Segment> ready;
        std::vector<uint8_t> window;                 // host decoding: the last 64 KiB decoded
    };
    std::vector<Stream> st;
    struct OutBuf { void* h = nullptr; size_t cap = 0; size_t live = 0; };
    std::vector<OutBuf> bufs;                        // decoded bytes wait here (pinned with a device, plain memory without)
    rh_lz4_decode_state* d_states = nullptr;
    Scratch d_jobs, d_src, d_out;
    void* h_src = nullptr; size_t h_src_cap = 0;     // pinned staging of the compressed bytes of a run
    rh_lz4_decode_job* h_jobs = nullptr; size_t h_jobs_cap = 0;
};

namespace {

// One block's sequences on the host (`hist` = the decoded bytes so far, ending at out + pos).
bool host_decode_block(const uint8_t* b, const uint8_t* bend, std::vector<uint8_t>& out, size_t block_start, size_t max_block, uint64_t before) {
    while (b < bend) {
        const unsigned token = *b++;
        size_t lit = token >> 4;
        if (lit == 15) { unsigned x; do { if (b >= bend) return false; x = *b++; lit += x; } while (x == 255); }
        if ((size_t)(bend - b) < lit || out.size() - block_start + lit > max_block) return false;
        out.insert(out.end(), b, b + lit); b += lit;
        if (b >= bend) break;
        if (bend - b < 2) return false;
        const size_t off = (size_t)b[0] | ((size_t)b[1] << 8); b += 2;
        size_t ml = token & 15;
        if (ml == 15) { unsigned x; do { if (b >= bend) return false; x = *b++; ml += x; } while (x == 255); }
        ml += 4;
        if (off == 0 || off > before + out.size() || out.size() - block_start + ml > max_block) return false;
        if (off > out.size()) return false;          // (the caller keeps 64 KiB of history in front of the data)
        const size_t from = out.size() - off;
        for (size_t i = 0; i < ml; i++) out.push_back(out[from + i]);
    }
    return b == bend;
}

int decoder_out_buffer(rh_decoder* d, size_t total) {
    int bi = -1;
    for (size_t i = 0; i < d->bufs.size(); i++) if (d->bufs[i].live == 0 && d->bufs[i].cap >= total) { bi = (int)i; break; }
    if (bi >= 0) return bi;
    for (size_t i = 0; i < d->bufs.size(); i++) if (d->bufs[i].live == 0) { bi = (int)i; break; }
    if (bi < 0) { d->bufs.emplace_back(); bi = (int)d->bufs.size() - 1; }
    rh_decoder::OutBuf& b = d->bufs[bi];
    if (b.h) { if (d->dev) (void)hipHostFree(b.h); else free(b.h); }
    b.h = nullptr; b.cap = 0;
    const size_t want = total + total / 4 + 4096;
    if (d->dev) { if (hipHostMalloc(&b.h, want, hipHostMallocDefault) != hipSuccess) { fail("rh_decoder: pinned allocation of %zu bytes failed", want); return -1; } }
    else { b.h = malloc(want); if (!b.h) { fail("rh_decoder: out of memory"); return -1; } }
    b.cap = want;
    return bi;
}

} // namespace

extern "C" {

int rh_decoder_create(rh_device* dev, uint32_t n_streams, rh_decoder** out) {
    if (!out || n_streams == 0) return fail("rh_decoder_create: bad argument");
    rh_decoder* d = new rh_decoder();
    d->dev = dev; d->n_streams = n_streams; d->st.resize(n_streams);
    if (dev) {
        if (enter(dev)) { delete d; return 1; }
        if (hipMalloc((void**)&d->d_states, (size_t)n_streams * sizeof(rh_lz4_decode_state)) != hipSuccess) { delete d; return fail("rh_decoder_create: device allocation failed"); }
        device_retain(dev);
    }
    *out = d;
    return 0;
}

void rh_decoder_destroy(rh_decoder* d) {
    if (!d) return;
    rh_device* dev = d->dev;
    if (dev) {
        (void)hipSetDevice(dev->ordinal);
        (void)hipStreamSynchronize(dev->stream);
        Scratch* all[] = { &d->d_jobs, &d->d_src, &d->d_out };
        for (Scratch* s : all) s->release();
        if (d->d_states) (void)hipFree(d->d_states);
        if (d->h_src) (void)hipHostFree(d->h_src);
        if (d->h_jobs) (void)hipHostFree(d->h_jobs);
    }
    for (auto& b : d->bufs) if (b.h) { if (dev) (void)hipHostFree(b.h); else free(b.h); }
    delete d;
    if (dev) device_release(dev);
}

int rh_decoder_feed(rh_decoder* d, uint32_t stream, const void* src, size_t len) {
    if (!d || (!src && len)) return fail("rh_decoder_feed: bad argument");
    if (stream >= d->n_streams) return fail("rh_decoder_feed: stream %u out of range (%u streams)", stream, d->n_streams);
    rh_decoder::Stream& S = d->st[stream];
    if (S.state != RH_STREAM_RUNNING) return 0;      // bytes behind the end mark, or behind the failure: ignored
    if (S.at > 0 && S.at * 2 > S.pending.size()) { S.pending.erase(S.pending.begin(), S.pending.begin() + (ptrdiff_t)S.at); S.at = 0; }
    S.pending.insert(S.pending.end(), (const uint8_t*)src, (const uint8_t*)src + len);
    return 0;
}

int rh_decoder_run(rh_decoder* d) {
    if (!d) return fail("rh_decoder_run: null decoder");
    if (d->dev && enter(d->dev)) return 1;
    struct Plan { uint32_t stream, n_blocks; size_t src_at, src_len, stage_off; uint64_t out_cap, out_off; bool ends; size_t end_skip; };
    std::vector<Plan> plans;
    size_t stage_total = 0, out_total = 0;
    for (uint32_t si = 0; si < d->n_streams; si++) {
        rh_decoder::Stream& S = d->st[si];
        if (S.state != RH_STREAM_RUNNING) continue;
        const uint8_t* p = S.pending.data(); const size_t n = S.pending.size();
        auto le32 = [&](size_t at) { return (uint32_t)p[at] | ((uint32_t)p[at + 1] << 8) | ((uint32_t)p[at + 2] << 16) | ((uint32_t)p[at + 3] << 24); };
        if (!S.header_done) {
            if (n - S.at < 7) continue;
            if (le32(S.at) != 0x184D2204u || (p[S.at + 4] >> 6) != 1) { S.state = RH_STREAM_FAILED; continue; }
            const uint8_t flg = p[S.at + 4], bd = p[S.at + 5];
            const size_t header = 6 + ((flg >> 3) & 1 ? 8 : 0) + (flg & 1 ? 4 : 0) + 1;
            if (n - S.at < header) continue;
            const uint32_t id = (bd >> 4) & 7u;
            if (id < 4) { S.state = RH_STREAM_FAILED; continue; }
            S.max_block = 1u << (8 + 2 * id);                             // 4: 64 KiB ... 7: 4 MiB
            if (d->dev && S.max_block > 65536u) {
                // only this stream fails (the message stays in rh_last_error); the others decode as usual
                (void)fail("rh_decoder: stream %u uses %u-byte blocks; the device decoder takes 64 KiB blocks (the reference's setting)", si, S.max_block);
                S.state = RH_STREAM_FAILED; continue;
            }
            S.trailer = (flg >> 4) & 1 ? 4 : 0; S.content_checksum = (flg >> 2) & 1;
            S.at += header; S.header_done = true;
        }
        Plan pl{}; pl.stream = si; pl.src_at = S.at;
        size_t pos = S.at;
        while (pl.n_blocks < RH_DECODER_MAX_BLOCKS_PER_RUN) {
            if (n - pos < 4) break;
            const uint32_t bh = le32(pos);
            if (bh == 0) { pl.ends = true; pl.end_skip = 4; break; }
            const size_t bsz = bh & 0x7fffffffu;
            if (bsz > (size_t)S.max_block + 65536) { if (pl.n_blocks == 0) S.state = RH_STREAM_FAILED; break; }      // (decoded after the blocks before it)
            if (n - pos - 4 < bsz + S.trailer) break;                     // incomplete: wait for more bytes
            pos += 4 + bsz + S.trailer; pl.n_blocks++;
        }
        pl.src_len = pos - S.at;
        if (pl.n_blocks == 0) { if (pl.ends) { S.at += pl.end_skip; S.state = RH_STREAM_ENDED; } continue; }
        pl.out_cap = (uint64_t)pl.n_blocks * S.max_block;
        pl.stage_off = stage_total; stage_total += ((pl.src_len + 64 + 4095) & ~(size_t)4095) + 4096;
        pl.out_off = out_total; out_total += pl.out_cap + 32;
        plans.push_back(pl);
    }
    if (plans.empty()) return 0;
    const int bi = decoder_out_buffer(d, out_total);
    if (bi < 0) return 1;
    uint8_t* const host_out = (uint8_t*)d->bufs[bi].h;
    std::vector<uint64_t> produced(plans.size(), 0);
    std::vector<uint32_t> blocks_done(plans.size(), 0), status(plans.size(), 0);
    if (d->dev) {
        rh_device* dev = d->dev;
        if (stage_total > d->h_src_cap) {
            if (d->h_src) (void)hipHostFree(d->h_src);
            d->h_src = nullptr; d->h_src_cap = 0;
            RH_HIP(hipHostMalloc(&d->h_src, stage_total + stage_total / 4, hipHostMallocDefault));
            d->h_src_cap = stage_total + stage_total / 4;
        }
        if (plans.size() > d->h_jobs_cap) {
            if (d->h_jobs) (void)hipHostFree(d->h_jobs);
            d->h_jobs = nullptr; d->h_jobs_cap = 0;
            RH_HIP(hipHostMalloc((void**)&d->h_jobs, plans.size() * 2 * sizeof(rh_lz4_decode_job), hipHostMallocDefault));
            d->h_jobs_cap = plans.size() * 2;
        }
        if (d->d_src.ensure(stage_total) || d->d_out.ensure(out_total) || d->d_jobs.ensure(plans.size() * sizeof(rh_lz4_decode_job))) return 1;
        for (size_t j = 0; j < plans.size(); j++) {
            const Plan& pl = plans[j];
            rh_decoder::Stream& S = d->st[pl.stream];
            memcpy((uint8_t*)d->h_src + pl.stage_off, S.pending.data() + pl.src_at, pl.src_len);
            rh_lz4_decode_job& jb = d->h_jobs[j];
            memset(&jb, 0, sizeof jb);
            jb.stream = pl.stream; jb.n_blocks = pl.n_blocks;
            jb.src = d->d_src.as<uint8_t>() + pl.stage_off; jb.src_len = (uint32_t)pl.src_len; jb.block_trailer = S.trailer;
            // (out_off is a multiple of 16 plus what makes the address congruent to the stream position)
            jb.out = d->d_out.as<uint8_t>() + ((pl.out_off + 15) & ~(size_t)15) + (S.decoded & 15u);
            jb.out_cap = pl.out_cap; jb.decoded_before = S.decoded;
        }
        hipStream_t s = dev->stream;
        RH_HIP(hipMemcpyAsync(d->d_src.ptr, d->h_src, stage_total, hipMemcpyHostToDevice, s));
        RH_HIP(hipMemcpyAsync(d->d_jobs.ptr, d->h_jobs, plans.size() * sizeof(rh_lz4_decode_job), hipMemcpyHostToDevice, s));
        const int rc = rhk::lz4f_decode_launch(d->d_jobs.as<rh_lz4_decode_job>(), (uint32_t)plans.size(), d->d_states, s);
        if (rc) return fail("rh_decoder: kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        RH_HIP(hipMemcpyAsync(d->h_jobs, d->d_jobs.ptr, plans.size() * sizeof(rh_lz4_decode_job), hipMemcpyDeviceToHost, s));
        RH_HIP(hipStreamSynchronize(s));
        size_t off = 0;
        for (size_t j = 0; j < plans.size(); j++) {
            produced[j] = d->h_jobs[j].out_len; blocks_done[j] = d->h_jobs[j].blocks_done; status[j] = d->h_jobs[j].status;
            if (produced[j] > plans[j].out_cap) return fail("rh_decoder: stream %u overflowed its output area", plans[j].stream);
            if (produced[j]) RH_HIP(hipMemcpyAsync(host_out + off, d->h_jobs[j].out, produced[j], hipMemcpyDeviceToHost, s));
            off += produced[j];
        }
        RH_HIP(hipStreamSynchronize(s));
    } else {
        size_t off = 0;
        for (size_t j = 0; j < plans.size(); j++) {
            const Plan& pl = plans[j];
            rh_decoder::Stream& S = d->st[pl.stream];
            // decoded bytes are appended behind the stream's window, so that matches find their history in front
            std::vector<uint8_t> work = S.window;
            const size_t hist = work.size();
            const uint8_t* p = S.pending.data() + pl.src_at;
            size_t pos = 0; uint64_t good = 0;
            for (uint32_t b = 0; b < pl.n_blocks; b++) {
                const uint32_t bh = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
                const size_t bsz = bh & 0x7fffffffu, block_start = work.size();
                bool ok = true;
                if (bh & 0x80000000u) { ok = bsz <= S.max_block; if (ok) work.insert(work.end(), p + pos + 4, p + pos + 4 + bsz); }
                else ok = host_decode_block(p + pos + 4, p + pos + 4 + bsz, work, block_start, S.max_block, S.decoded - hist);
                if (!ok) { status[j] = RH_LZ4_DECODE_MALFORMED; work.resize(block_start); break; }
                pos += 4 + bsz + S.trailer; blocks_done[j]++; good = work.size() - hist;
            }
            produced[j] = good;
            memcpy(host_out + off, work.data() + hist, good);
            off += good;
            const size_t keep = std::min<size_t>(work.size(), 65536);
            S.window.assign(work.end() - (ptrdiff_t)keep, work.end());
        }
    }
    size_t off = 0;
    for (size_t j = 0; j < plans.size(); j++) {
        const Plan& pl = plans[j];
        rh_decoder::Stream& S = d->st[pl.stream];
        if (produced[j]) {
            rh_decoder::Segment sg; sg.ptr = host_out + off; sg.len = produced[j]; sg.buf = bi;
            S.ready.push_back(sg); d->bufs[bi].live++;
        }
        off += produced[j];
        S.decoded += produced[j];
        if (status[j] != RH_LZ4_DECODE_OK) { S.state = RH_STREAM_FAILED; continue; }
        S.at = pl.src_at + pl.src_len;
        if (pl.ends) { S.at += pl.end_skip; S.state = RH_STREAM_ENDED; }
    }
    return 0;
}

int rh_decoder_peek(rh_decoder* d, uint32_t stream, const uint8_t** ptr, size_t* len) {
    if (!d || !ptr || !len) return fail("rh_decoder_peek: bad argument");
    if (stream >= d->n_streams) return fail("rh_decoder_peek: stream %u out of range", stream);
    rh_decoder::Stream& S = d->st[stream];
    if (S.ready.empty()) { *ptr = nullptr; *len = 0; return 0; }
    *ptr = S.ready.front().ptr; *len = S.ready.front().len;
    return 0;
}

int rh_decoder_consume(rh_decoder* d, uint32_t stream, size_t n) {
    if (!d) return fail("rh_decoder_consume: null decoder");
    if (stream >= d->n_streams) return fail("rh_decoder_consume: stream %u out of range", stream);
    rh_decoder::Stream& S = d->st[stream];
    while (n) {
        if (S.ready.empty()) return fail("rh_decoder_consume: more bytes than are ready");
        rh_decoder::Segment& sg = S.ready.front();
        const size_t k = std::min(n, sg.len);
        sg.ptr += k; sg.len -= k; n -= k;
        if (sg.len == 0) { d->bufs[sg.buf].live--; S.ready.pop_front(); }
    }
    return 0;
}

int rh_decoder_stream_info(rh_decoder* d, uint32_t stream, int* state, size_t* buffered, size_t* ready) {
    if (!d) return fail("rh_decoder_stream_info: null decoder");
    if (stream >= d->n_streams) return fail("rh_decoder_stream_info: stream %u out of range", stream);
    const rh_decoder::Stream& S = d->st[stream];
    if (state) *state = S.state;
    if (buffered) *buffered = S.pending.size() - S.at;
    if (ready) { size_t r = 0; for (const auto& sg : S.ready) r += sg.len; *ready = r; }
    return 0;
}

} // extern "C"
