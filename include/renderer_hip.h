/* renderer_hip.h -- C ABI of librenderer_hip.so, the MI355X (gfx950) implementation of the device layer that
 * the `renderer` crate reaches through wgpu.  One export per wgpu call site of the reference; each declaration
 * cites the reference call site(s) it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every call returns 0 on success, non-zero on failure; rh_last_error() gives a thread-local message.
 *     The reference panics on these conditions (unwrap/panic!), so a host shim should panic on non-zero.
 *   - handles are opaque and reference counted: *_create returns a handle with one reference,
 *     *_retain / *_release adjust it (the reference clones Buffer/Texture/Program freely: src/buffer.rs:3-6,
 *     src/texture.rs:3-6, src/program.rs:3-6, and un-submitted command buffers keep old resources alive).
 *   - single-thread affinity per device, like the reference (RefCell/Rc everywhere), except
 *     rh_buffer_mapped_ptr memory, which may be read from any thread until the buffer is released
 *     (the reference reads mapped capture buffers on compressor worker threads: src/compressor.rs:118).
 */
#ifndef RENDERER_HIP_H
#define RENDERER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rh_device rh_device;
typedef struct rh_buffer rh_buffer;
typedef struct rh_texture rh_texture;
typedef struct rh_sampler rh_sampler;
typedef struct rh_module rh_module;
typedef struct rh_pipeline rh_pipeline;
typedef struct rh_cmd rh_cmd;          /* command encoder */
typedef struct rh_cmdbuf rh_cmdbuf;    /* finished command buffer */
typedef struct rh_compressor rh_compressor;

/* src/format.rs:3-9 (declaration order = bincode discriminant) */
enum { RH_FORMAT_R_U8 = 0, RH_FORMAT_BGRA_U8 = 1, RH_FORMAT_RGBA_U8 = 2, RH_FORMAT_RGBA_F16 = 3, RH_FORMAT_RGBA_F32 = 4 };
/* src/filter_mode.rs:2-5 */
enum { RH_FILTER_LINEAR = 0, RH_FILTER_NEAREST = 1 };
/* src/primitive.rs:2-5 */
enum { RH_TOPOLOGY_TRIANGLE_LIST = 0, RH_TOPOLOGY_TRIANGLE_STRIP = 1 };
/* wgpu::BlendFactor values the reference uses (src/blend_mode.rs:8-14) plus Zero/SrcAlpha */
enum { RH_BLEND_ZERO = 0, RH_BLEND_ONE = 1, RH_BLEND_SRC_ALPHA = 2, RH_BLEND_ONE_MINUS_SRC_ALPHA = 3 };
/* wgpu::LoadOp (src/render_pass.rs:72) */
enum { RH_LOAD_OP_LOAD = 0, RH_LOAD_OP_CLEAR = 1 };
/* kinds of flat binding slot, in the order src/pipeline.rs:95-113 emits them */
enum { RH_SLOT_STORAGE = 0, RH_SLOT_UNIFORM = 1, RH_SLOT_TEXTURE = 2, RH_SLOT_SAMPLER = 3 };
/* src/visibility.rs:2-15 */
enum { RH_STAGE_VERTEX = 1, RH_STAGE_FRAGMENT = 2 };
/* wgpu::BufferUsages bits the reference sets (src/attribute.rs:11, src/instanced.rs:10, src/uniform.rs:10,
 * src/video_recorder.rs:88) */
enum { RH_BUFFER_VERTEX = 1, RH_BUFFER_STORAGE = 2, RH_BUFFER_UNIFORM = 4, RH_BUFFER_COPY_DST = 8, RH_BUFFER_MAP_READ = 16 };
/* wgpu::TextureUsages bits (src/texture.rs:111-114) */
enum { RH_TEXTURE_BINDING = 1, RH_TEXTURE_COPY_DST = 2, RH_TEXTURE_RENDER_ATTACHMENT = 4, RH_TEXTURE_COPY_SRC = 8 };
/* device options */
enum {
    RH_OPTION_CLIP_TO_VIEWPORT = 0,   /* 1 (default): fragments outside the viewport rectangle are clipped, as strict
                                         clip-volume clipping gives; 0: only the attachment bounds restrict them */
    RH_OPTION_TIMING = 1,             /* 1: record HIP events around the binning phase and the tile kernel of every
                                         render pass (read with rh_device_drain_timings); 2: around the tile kernel
                                         only (bin_ms and setup_ms then read 0; every mark is a bubble of some 7 us in
                                         the stream); default 0 */
    RH_OPTION_STRIP_ROW0 = 2,         /* sort-first split of a large target over several devices: render passes touch */
    RH_OPTION_STRIP_ROW1 = 3,         /* only pixel rows [ROW0, ROW1) (multiples of 32, or the target height); clears
                                         included.  ROW1 = 0 (default) disables the restriction. */
    RH_OPTION_GENERAL_KERNELS = 4,    /* 1: always run the kernel body that reads blend factors, texture formats and
                                         filters at run time instead of the body specialised for the pipeline's state,
                                         and the prefix-sum form that serves any length instead of the short forms
                                         (identical results; for testing); default 0 */
    RH_OPTION_BINNING_TICKETS = 5,    /* 1: the workgroups of the binning chain always take their place in the chain from
                                         a ticket counter, as launches too large to fit the device at once do (identical
                                         results; for testing); default 0 */
    RH_OPTION_EARLY_BINNING = 6,      /* 1 (default): binning and set-up of a pass that is queued whole run on a stream of
                                         their own, beside the tile kernel of the pass before (they read only what the
                                         vertex stage reads, and wait for everything queued before the pass that could
                                         change it); 1 does so for passes with room for four million (tile, pair) pairs
                                         or more, whose binning chain is long enough for that to pay, 2 for every pass
                                         it is valid for; 0: everything on one stream (identical results) */
    RH_OPTION_TAIL_SPLIT = 7,         /* the tiles beyond the last full round of workgroups the GPU holds at once are cut
                                         into bands of rows (halves, quarters or eighths of a tile) with a workgroup
                                         each, so that the end of the pass fills the SIMDs instead of running a few long
                                         waves on each.  0: never; 1: the finest bands of which they all fit the round;
                                         2 (default): where those are halves, quarters for as many tiles as there is
                                         room left; 3: thirds (11, 11 and 10 rows) where the tail fits the round in
                                         thirds but not in quarters, else as 2 (identical results) */
    RH_OPTION_EAGER_COUNT = 8,        /* 1 (default): rh_submit, before it returns, looks whether the pair count of its last
                                         pass has already arrived (without waiting for it); 0: it leaves that to the next
                                         call, so that the pass is certainly still awaiting its count when rh_submit
                                         returns (for testing what the next call does with it) */
    RH_OPTION_SETUP_LAUNCH = 9,       /* 0 (default): the tile kernel sets axis-aligned strip quads with affine varyings up
                                         itself, in LDS, and only other primitives get records from the set-up launch; 1:
                                         every record comes from the set-up launch (identical results; for testing the two
                                         set-up paths against each other) */
    RH_OPTION_WAVE_PRIORITY = 10      /* 1 (default): the waves of the tile kernel set their issue priority by the work they
                                         have left, so that the waves of a SIMD end closer together and a pass drains
                                         faster; 0: priorities are left alone (identical results) */
};

#define RH_MAX_ATTRIBUTES 8
#define RH_MAX_SLOTS 32               /* 8 bind groups x 4 bindings (src/pipeline.rs:25-27) */
#define RH_BINDINGS_PER_GROUP 4
#define RH_MAX_COLOR_TARGETS 4
#define RH_COPY_BYTES_PER_ROW_ALIGNMENT 256   /* wgpu::COPY_BYTES_PER_ROW_ALIGNMENT (src/video_recorder.rs:75) */

const char* rh_last_error(void);
int rh_abi_version(void);

/* ---- device: get_instance/get_adapter/get_device + surface configure (src/renderer.rs:301-354) ------------------- */
/* Creates a device on HIP ordinal `hip_ordinal` with an offscreen BGRA8 "screen" of width x height (there is no
 * display: Target::Screen becomes an offscreen texture, src/target.rs:10). */
int rh_device_create(int hip_ordinal, uint32_t width, uint32_t height, rh_device** out);
void rh_device_destroy(rh_device* dev);
/* adapter.get_info() (src/renderer.rs:199-201): device name into `name` (NUL terminated, truncated to cap). */
int rh_device_info(rh_device* dev, char* name, size_t cap);
int rh_device_set_option(rh_device* dev, int option, int value);
/* Blocks until all submitted work has finished (wgpu has no call site for this in the reference; tests and
 * benchmarks need it). */
int rh_device_synchronize(rh_device* dev);
/* The HIP stream all submitted work runs on (a hipStream_t), for callers that time with HIP events. */
void* rh_device_stream(rh_device* dev);

/* surface.configure / get_current_texture / present (src/renderer.rs:301-319,96,115) */
int rh_screen_texture(rh_device* dev, rh_texture** out);       /* borrowed: do not release */
int rh_screen_resize(rh_device* dev, uint32_t width, uint32_t height);

/* ---- buffers: device.create_buffer, queue.write_buffer, mapped-at-creation fill (src/buffer.rs:42,54,61-64;
 *      src/video_recorder.rs:88-91) ------------------------------------------------------------------------------------ */
int rh_buffer_create(rh_device* dev, size_t size, uint32_t usage, const void* init, size_t init_len, rh_buffer** out);
int rh_buffer_write(rh_buffer* buf, size_t offset, const void* data, size_t len);
size_t rh_buffer_size(rh_buffer* buf);
/* Device address of the buffer (for callers that fill it with their own kernels / torch tensors). */
void* rh_buffer_device_ptr(rh_buffer* buf);
void rh_buffer_retain(rh_buffer* buf);
void rh_buffer_release(rh_buffer* buf);
/* buffer.slice(..).map_async(Read, cb) / get_mapped_range (src/video_recorder.rs:141; src/video_frame.rs:64;
 * src/compressor.rs:118).  The callback fires from rh_poll or rh_submit once the copy into pinned host memory is done. */
typedef void (*rh_map_callback)(void* user, int ok);
int rh_buffer_map_async(rh_buffer* buf, rh_map_callback cb, void* user);
int rh_buffer_mapped_ptr(rh_buffer* buf, const uint8_t** ptr, size_t* len);
int rh_buffer_unmap(rh_buffer* buf);
/* Drives map callbacks (wgpu fires them from later submits; the reference never calls device.poll). */
int rh_poll(rh_device* dev, int wait);

/* ---- textures: device.create_texture / create_view / create_sampler / queue.write_texture
 *      (src/texture.rs:110-158,45-56) ---------------------------------------------------------------------------------- */
int rh_texture_create(rh_device* dev, uint32_t width, uint32_t height, uint32_t layers, uint32_t format,
                      uint32_t samples, uint32_t usage, rh_texture** out);
int rh_texture_write(rh_texture* tex, uint32_t x, uint32_t y, uint32_t layer, uint32_t width, uint32_t height,
                     const void* data, size_t data_len, uint32_t bytes_per_row);
/* Synchronous read-back of one layer, tightly packed rows (test/inspection helper: the reference reads textures
 * only through copy_texture_to_buffer + map). */
int rh_texture_read(rh_texture* tex, uint32_t layer, void* out, size_t out_len);
int rh_texture_size(rh_texture* tex, uint32_t* width, uint32_t* height, uint32_t* layers, uint32_t* format);
void* rh_texture_device_ptr(rh_texture* tex);
void rh_texture_retain(rh_texture* tex);
void rh_texture_release(rh_texture* tex);
/* ClampToEdge on all axes, mag = min = filter, mipmap nearest, lod 0..0 (src/texture.rs:141-158) */
int rh_sampler_create(rh_device* dev, uint32_t filter, rh_sampler** out);
void rh_sampler_release(rh_sampler* s);

/* ---- shader modules: device.create_shader_module (src/program.rs:43-48) -------------------------------------------- */
/* `code_object` is a gfx950 code object produced by the offline shader build (src/compiler.rs:22-33 replaced by
 * hipcc --genco); loaded with hipModuleLoadData.  A vertex-stage object exports the binning kernels, a
 * fragment-stage object exports the tile kernels; both carry an `rh_module_info` record. */
int rh_module_load(rh_device* dev, const void* code_object, size_t size, rh_module** out);
void rh_module_retain(rh_module* m);
void rh_module_release(rh_module* m);

/* ---- pipelines: create_bind_group_layout / create_bind_group / create_pipeline_layout / create_render_pipeline
 *      (src/pipeline.rs:90-162,197-237) ------------------------------------------------------------------------------- */
typedef struct rh_attribute_desc {
    uint32_t location;      /* shader location (src/attribute.rs:28) */
    uint32_t size;          /* 1..4 f32 per vertex, tightly packed, step mode Vertex (src/pipeline.rs:219-229) */
} rh_attribute_desc;

typedef struct rh_slot_desc {
    uint32_t kind;          /* RH_SLOT_* */
    uint32_t visibility;    /* RH_STAGE_* mask */
    void* resource;         /* rh_buffer* (storage/uniform), rh_texture* or rh_sampler* */
} rh_slot_desc;

typedef struct rh_color_target_desc {
    uint32_t format;        /* RH_FORMAT_* */
    uint32_t src_factor;    /* same factors for colour and alpha, op Add, write mask ALL (src/blend_mode.rs:23-40) */
    uint32_t dst_factor;
} rh_color_target_desc;

typedef struct rh_pipeline_desc {
    rh_module* vertex_module;
    rh_module* fragment_module;
    uint32_t n_attributes;
    const rh_attribute_desc* attributes;   /* vertex buffer slot i = attributes[i] (src/render_pass.rs:35-37) */
    uint32_t n_slots;
    const rh_slot_desc* slots;             /* flat slot i = (set i/4, binding i%4) (src/pipeline.rs:90-131) */
    uint32_t topology;                     /* RH_TOPOLOGY_*; no culling, no depth/stencil (src/pipeline.rs:150-159,197-207) */
    uint32_t samples;                      /* MultisampleState.count (src/pipeline.rs:209-211): 1, 2, 4 or 8; mask !0,
                                              no alpha-to-coverage; a multisampled pass takes one attachment */
    uint32_t n_targets;
    const rh_color_target_desc* targets;   /* incl. the extra RGBA8 recording target (src/pipeline.rs:133-142) */
} rh_pipeline_desc;

int rh_pipeline_create(rh_device* dev, const rh_pipeline_desc* desc, rh_pipeline** out);
void rh_pipeline_retain(rh_pipeline* p);
void rh_pipeline_release(rh_pipeline* p);

/* ---- command encoding: create_command_encoder / begin_render_pass / set_* / draw / copy_texture_to_buffer / finish
 *      (src/render_pass.rs:25-53; src/video_recorder.rs:107-128) ------------------------------------------------------ */
typedef struct rh_attachment_desc {
    rh_texture* texture;
    uint32_t load_op;       /* RH_LOAD_OP_* ; store op is always Store (src/render_pass.rs:73) */
    double clear[4];        /* wgpu::Color (src/clear_color.rs:7-16) */
    rh_texture* resolve_target;   /* NULL, or for a multisampled `texture` the single-sampled texture that receives the
                                     averaged samples when the pass ends (src/render_pass.rs:76-81: view = the pipeline's
                                     MSAA texture, resolve_target = the target's view).  Load/clear act on `texture`. */
} rh_attachment_desc;

typedef struct rh_pass_desc {
    rh_pipeline* pipeline;
    uint32_t n_attachments;
    const rh_attachment_desc* attachments;
    uint32_t n_vertex_buffers;
    rh_buffer* const* vertex_buffers;      /* slot order = pipeline attribute order */
    int32_t has_viewport;                  /* src/render_pass.rs:39-41 */
    float viewport[6];                     /* x, y, width, height, min_depth, max_depth */
    uint32_t vertex_count;                 /* draw(0..vertex_count, 0..instance_count) (src/render_pass.rs:43) */
    uint32_t instance_count;
} rh_pass_desc;

int rh_cmd_begin(rh_device* dev, rh_cmd** out);
int rh_cmd_render_pass(rh_cmd* cmd, const rh_pass_desc* pass);
/* encoder.copy_texture_to_buffer (src/video_recorder.rs:113-127): copies width x height texels starting at
 * (origin_x, origin_y) of layer 0 into `buf` with `bytes_per_row` pitch (a multiple of 256). */
int rh_cmd_copy_texture_to_buffer(rh_cmd* cmd, rh_texture* tex, uint32_t origin_x, uint32_t origin_y,
                                  rh_buffer* buf, uint32_t bytes_per_row, uint32_t rows_per_image,
                                  uint32_t width, uint32_t height);
int rh_cmd_finish(rh_cmd* cmd, rh_cmdbuf** out);   /* consumes cmd */
void rh_cmdbuf_release(rh_cmdbuf* cb);
/* queue.submit (src/renderer.rs:119-122): executes the command buffers in order on the device's stream and
 * releases them.  Also polls pending map callbacks. */
int rh_submit(rh_device* dev, rh_cmdbuf* const* list, size_t n);

/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): executes the command buffers like rh_submit but calls no hipMalloc,
 * hipFree or other allocating routine.  The scratch of a render pass (binning arrays, set-up records) lives in a static
 * device array of 40 MiB that every GPU the library runs on holds until the process ends; a pass is limited to 65536
 * triangle pairs, 131072 (tile, triangle pair) pairs, 16384 tiles of 32x32 pixels and 16 MiB of set-up records, and is
 * rejected with an error beyond that.  Differences from rh_submit: the command buffers are NOT consumed (release them
 * with rh_cmdbuf_release once the work has finished; they may be submitted again), map callbacks are not driven, no
 * timings are recorded, and textures are sampled through footprint copies only where an earlier rh_submit made them.
 * Calls on one GPU share that GPU's array, from whichever rh_device they come: a call returns with its kernels queued,
 * and the next call on that GPU waits for them on the device (one event per GPU), so calls made one after the other do
 * not overlap; calls made at the same time from several threads on one GPU race.  A pass with triangles that cross a
 * clip plane is rejected: the kernels that clip them use private memory, which the runtime allocates.  A pass that an
 * earlier rh_submit left pending and that has to run again (too many pairs for its array, or its pipeline's first
 * clipped triangle) is reported as an error here, because running it again allocates: call rh_device_synchronize or
 * rh_submit first. */
int rh_submit_static(rh_device* dev, rh_cmdbuf* const* list, size_t n);

/* ---- statistics of the most recent render pass executed (diagnostics for benchmarks) ------------------------------- */
typedef struct rh_pass_stats {
    uint64_t clusters;        /* binning units (pairs of consecutive triangles) */
    uint64_t bin_entries;     /* (cluster, tile) pairs after binning */
    uint32_t tiles_x, tiles_y;
    uint32_t tile_w, tile_h;    /* pixels per tile (one workgroup) */
    uint64_t reruns;          /* passes of this device, so far, that were queued before their pair count was known and ran
                                 again: they did not fit the room they had been given, or met the pipeline's first triangle
                                 that crosses a clip plane (from then on the pipeline runs the kernels that clip) */
    uint64_t early;           /* passes of this device, so far, whose binning ran on the binning stream beside the tile kernel
                                 of the pass before (RH_OPTION_EARLY_BINNING) */
} rh_pass_stats;
int rh_device_last_pass_stats(rh_device* dev, rh_pass_stats* out);

/* Sums of the HIP-event timings recorded since the previous drain (RH_OPTION_TIMING).  Waits for the recorded
 * passes to finish. */
typedef struct rh_timings {
    uint32_t passes;          /* render passes covered */
    double bin_ms;            /* vertex stage and binning: the (tile, primitive pair) array built and sorted (the form that
                               * waits for the pair count includes that host round trip) */
    double tile_ms;           /* tile kernel (coverage + shading + blend + store) */
    double setup_ms;          /* triangle set-up: one record per (tile, primitive pair), tile list bounds */
} rh_timings;
int rh_device_drain_timings(rh_device* dev, rh_timings* out);

/* ---- capture codec: lz4f::WriteCompressor (src/compressor.rs:98,126-129,138-144) and lz4f::BufReadDecompressor
 *      (src/decompressor.rs:198,216-235) -------------------------------------------------------------------------------- */
/* Creates `n_streams` independent LZ4F streams (the reference opens one per worker thread/file,
 * src/compressor.rs:24-26,94-98) with the reference's preferences: compression level `level` (0..2 select the LZ4
 * "fast" compressor and give identical bytes; higher levels are not implemented), favor_dec_speed off, auto_flush on,
 * everything else default (block-linked, 64 KiB blocks, no checksums).  Streams are compressed on the GPU, one
 * workgroup per stream; stream bytes are identical to liblz4's for the same write sequence. */
int rh_compressor_create(rh_device* dev, uint32_t n_streams, int level, rh_compressor** out);
void rh_compressor_destroy(rh_compressor* c);
/* Queues one packet for `stream`, i.e. the four write_all calls of src/compressor.rs:126-129:
 * [packet_len u64 BE][metadata_len u64 BE][metadata][image bytes], where the image is the whole device buffer `image`
 * (NULL for a frame without image data: then only the first three writes happen).  The buffer is retained until the
 * packet has been compressed. */
int rh_compressor_write_packet(rh_compressor* c, uint32_t stream, const void* metadata, size_t metadata_len, rh_buffer* image);
/* Starts compressing everything queued (all streams concurrently, on a HIP stream of the compressor's own that waits
 * for the frames rendered so far) and returns while the last round of packets is still being compressed; the bytes of
 * earlier rounds become ready.  This is the reference's hand-off to its worker threads (src/compressor.rs:64-69): the
 * caller goes on rendering, and writes the ready bytes to the stream files meanwhile. */
int rh_compressor_submit(rh_compressor* c);
/* Compresses everything queued and waits: afterwards every byte is ready. */
/* Waits for the round that rh_compressor_submit started: its compressed bytes are on their way to the host when this
 * returns (what is still queued stays queued).  For callers that would rather have the GPU to themselves while they
 * render the next round's frames than share it with the compression kernel. */
int rh_compressor_wait(rh_compressor* c);
int rh_compressor_flush(rh_compressor* c);
/* Bytes of `stream` ready to be read (the 7-byte frame header precedes the first block). */
size_t rh_compressor_pending(rh_compressor* c, uint32_t stream);
/* Moves up to `cap` ready bytes of `stream` into dst. */
int rh_compressor_read(rh_compressor* c, uint32_t stream, void* dst, size_t cap, size_t* len);
/* Zero-copy form of rh_compressor_read: *ptr / *len describe the first ready run of bytes of `stream` (pinned host
 * memory owned by the compressor, NULL / 0 when nothing is ready); rh_compressor_consume releases the first n ready
 * bytes.  A run stays valid until it is consumed; runs of different streams may be written out from different threads,
 * peek and consume themselves are called from one thread at a time, which may be another thread than the one that
 * queues packets with rh_compressor_write_packet meanwhile (they share no state); every other call on the compressor
 * (submit, flush, finish_stream, destroy) must wait until that thread is done. */
int rh_compressor_peek(rh_compressor* c, uint32_t stream, const uint8_t** ptr, size_t* len);
int rh_compressor_consume(rh_compressor* c, uint32_t stream, size_t n);
/* Flushes and appends the LZ4F end mark to `stream` (LZ4F_compressEnd, when the reference drops the writer). */
int rh_compressor_finish_stream(rh_compressor* c, uint32_t stream);
/* What the compression kernel did since the previous call (the counters start again): for benchmarks; the reference
 * prints comparable figures in its statistics display (src/compressor.rs:150-229). */
typedef struct rh_compressor_stats {
    uint64_t rounds;          /* kernel launches that have finished */
    uint64_t jobs;            /* packets compressed (one per stream and launch) */
    uint64_t bytes_in;        /* uncompressed bytes read */
    uint64_t bytes_out;       /* LZ4F bytes written */
    double kernel_ms;         /* HIP-event time of those launches on the compression stream */
} rh_compressor_stats;
int rh_compressor_drain_stats(rh_compressor* c, rh_compressor_stats* out);
/* PngEncoder::encode_to_bytes (src/png_encoder.rs:6-38) on the device: encodes the captured frame held in `image`
 * (RGBA8 rows `padded_bytes_per_row` apart, `width` x `height`, as src/video_recorder.rs:66-128 lays it out) as an 8-bit
 * RGBA PNG and writes the complete file to `out` (host memory, `cap` bytes; *len = file size).  Only the finished PNG
 * bytes cross PCIe.  The deflate stream differs from the `png` crate's; the decoded pixels are identical.  Fails when
 * `cap` is too small: rh_png_bound gives a sufficient size. */
size_t rh_png_bound(uint32_t width, uint32_t height);
int rh_png_encode(rh_device* dev, rh_buffer* image, uint32_t width, uint32_t height, uint32_t padded_bytes_per_row,
                  void* out, size_t cap, size_t* len);
/* Host-side LZ4 frame decoder.  Tolerates a truncated tail (src/decompressor.rs:216-235 stops at an unclean end):
 * decoding stops after the last complete block; *complete is 1 when the end mark was reached. */
int rh_lz4f_decompress(const void* src, size_t src_len, void* dst, size_t dst_cap, size_t* produced, size_t* consumed, int* complete);

/* lz4f::BufReadDecompressor over many stream files at once (src/decompressor.rs:186-249: one reader per `.sz` file, read
 * incrementally).  The caller feeds each stream the bytes of its file in pieces of any size; rh_decoder_run decodes the
 * complete LZ4F blocks fed so far, every stream in the same kernel launch (one wavefront per stream; at most
 * RH_DECODER_MAX_BLOCKS_PER_RUN blocks per stream and call), and the decoded bytes are handed out in order through
 * peek / consume (pinned host memory owned by the decoder).  A stream whose file ends in the middle of a block simply
 * stops at its last complete block (src/decompressor.rs:216-235 stops at an unclean end); a malformed block ends its
 * stream the same way.  dev = NULL: the same object decoding on the host (no GPU needed). */
typedef struct rh_decoder rh_decoder;
#define RH_DECODER_MAX_BLOCKS_PER_RUN 32
enum { RH_STREAM_RUNNING = 0, RH_STREAM_ENDED = 1 /* end mark reached */, RH_STREAM_FAILED = 2 /* not an LZ4 frame, or a malformed block */ };
int rh_decoder_create(rh_device* dev, uint32_t n_streams, rh_decoder** out);
void rh_decoder_destroy(rh_decoder* d);
int rh_decoder_feed(rh_decoder* d, uint32_t stream, const void* src, size_t len);
int rh_decoder_run(rh_decoder* d);
int rh_decoder_peek(rh_decoder* d, uint32_t stream, const uint8_t** ptr, size_t* len);
int rh_decoder_consume(rh_decoder* d, uint32_t stream, size_t n);
/* state: RH_STREAM_*; buffered: fed bytes not decoded yet; ready: decoded bytes not consumed yet (any pointer may be NULL) */
int rh_decoder_stream_info(rh_decoder* d, uint32_t stream, int* state, size_t* buffered, size_t* ready);

#ifdef __cplusplus
}
#endif
#endif /* RENDERER_HIP_H */
