// Plain-data structures shared between the host library (librenderer_hip.so) and the shader code objects it
// launches.  Everything here is passed by value as kernel arguments; keep it POD and keep the layout stable:
// a code object records RH_DEVICE_ABI_VERSION in its rh_module_info and the host refuses a mismatch.
#pragma once
#include <stdint.h>

#define RH_DEVICE_ABI_VERSION 23u
#define RH_MODULE_MAGIC 0x52484d4fu   // "RHMO"

#define RH_DEV_MAX_ATTRIBUTES 8
#define RH_DEV_MAX_SLOTS 32
#define RH_DEV_MAX_TARGETS 4

// Rasteriser geometry
#define RH_SUBPIXEL_BITS 8
#define RH_TILE_W 32               // screen tile in pixels: one wavefront (a workgroup of its own) shades a tile from its
#define RH_TILE_H 32               // list of set-up records; no two waves share pixels, so a tile needs no barrier
#define RH_TILE_THREADS 64
#ifndef RH_ROUND_HITS
#define RH_ROUND_HITS 32           // set-up records a tile wave brings into LDS at a time
#endif
#define RH_MS_EXTRA_WORDS(S) (12 + 6 * (S))   // dwords a multisampled pass adds to an extended record (S samples)

enum { RH_MODULE_STAGE_VERTEX = 1, RH_MODULE_STAGE_FRAGMENT = 2 };

// Exported by every shader code object as the global `rh_module_info`.
struct rh_module_info_t {
    uint32_t magic;
    uint32_t abi_version;
    uint32_t stage;          // RH_MODULE_STAGE_*
    uint32_t n_varyings;     // floats interpolated per fragment
    uint32_t n_outputs;      // colour outputs the fragment function writes
    uint32_t vertex_tag;     // hash of the vertex function's source name: ties a fragment object to its vertex object
    uint32_t rec_bytes;      // fragment objects: bytes of one set-up record (depends on the number of varyings)
    uint32_t ext_bytes;      // fragment objects: bytes of one extended record of a single-sampled pass
    uint32_t poly_item_bytes;   // fragment objects: the form of footprint copies its bilinear body reads (rh_slot_dev::poly): 16, 32 or 48 (both)
};

struct rh_slot_dev {
    const void* ptr;         // buffer base / texture base (layer 0)
    uint32_t width, height, layers;
    uint32_t format;         // RH_FORMAT_*
    uint32_t filter;         // RH_FILTER_* (for texture slots: the filter of the sampler that follows)
    uint32_t size_bytes;     // buffers: capacity in bytes (whole buffer is bound, reference src/instanced.rs:18);
                             // textures with `poly`: bytes of one footprint copy
    // Textures: null, or RH_POLY_COPIES copies of the texture made for bilinear sampling.  An item of a copy is the
    // whole 2x2 footprint of a texel: t(x, y), t(x+1, y), t(x, y+1), t(x+1, y+1) (clamped to the layer), so a fragment
    // fetches its four taps at ONE address.  Two forms (rh_module_info_t::poly_item_bytes; 48 there = both, RH_POLY_COPIES
    // copies of the 32-byte form followed by as many of the 16-byte form in one block, `size_bytes` = one 32-byte copy):
    //   16 bytes: the four texels as they are;
    //   32 bytes: per channel c two dwords of f16 pairs, { t(x,y).c | (t(x+1,y).c - t(x,y).c) << 16 } and the same for
    //     the row below: 8-bit values and their differences are exact in f16, and an f32 fma that reads f16 operands
    //     (v_fma_mix_f32) does a horizontal lerp fma(a, right - left, left) in one instruction where the byte form takes
    //     two conversions, a subtraction and the fma -- the same f32 result, bit for bit.
    // Copy k regroups the items of every row by their phase modulo
    // S = 2^k: item x sits at column (x mod S) * (width / S) + x / S.  A minified draw samples texels S apart; in the
    // copy for its S they are neighbours, so the lanes of a wave share cache lines instead of fetching one line per
    // fragment (a CU fills its L1 at 64 B per clock: at one 128-byte line per fragment that is the whole frame time).
    // Copy 0 is the plain order.
    const void* poly;
};
#define RH_POLY_COPIES 6

struct rh_draw_params {
    const float* attr[RH_DEV_MAX_ATTRIBUTES];        // indexed by shader location
    uint32_t attr_size[RH_DEV_MAX_ATTRIBUTES];
    rh_slot_dev slots[RH_DEV_MAX_SLOTS];             // flat slot index = set * 4 + binding
    float vp_hw, vp_hh, vp_ox, vp_oy;                // viewport transform: x_w = x*hw + ox ; y_w = y*(-hh) + oy
    int32_t sc_x0, sc_y0, sc_x1, sc_y1;              // scissor rectangle in pixels [x0,x1) x [y0,y1)
    uint32_t width, height;                          // attachment size in pixels
    uint32_t tiles_x, tiles_y;
    uint32_t tile_first;                             // index of the first tile this launch covers (strip rendering)
    uint32_t topology;
    uint32_t vertex_count, instance_count;
    uint32_t tpi;                                    // triangles per instance
    uint32_t cpi;                                    // clusters (pairs of consecutive triangles) per instance
    uint32_t n_clusters;                             // instance_count * cpi
    uint32_t samples;                                // 1, or 2 / 4 / 8 (multisampled pass)
    int32_t bbox_pad;                                // growth of primitive bounding boxes in 1/256 px: 0, or 128 when
                                                     // samples away from the pixel centre can be covered
    uint32_t no_lean;                                // 1: no cluster counts as a lean quad (rh_geometry.h): every record comes
                                                     // from the set-up launch (RH_OPTION_SETUP_LAUNCH; identical results)
};

// Packed tile rectangle of a cluster: lo = x0 | y0 << 16, hi = x1 | y1 << 16 (inclusive); x0 > x1 marks "touches nothing".
// Bit 15 of hi: RH_RANGE_GENERAL.
// Tile columns take ten bits.  Above x0, `lo` carries how many sub-records the cluster's first (bits 10-12) and second
// triangle (bits 13-15) need when a triangle of the cluster is clipped, 0..4 each: the cluster then makes that many
// (tile, cluster) pairs per tile instead of one, the pair's value carrying the number of the sub-record in its top three
// bits (0..3: first triangle, 4..7: second; cluster indices stay below 2^29).  Both zero: the usual cluster, one pair.
struct rh_tile_range { uint32_t lo, hi; };
#define RH_RANGE_X0(lo) ((lo) & 0x3ffu)
#define RH_RANGE_X1(hi) ((hi) & 0x3ffu)
#define RH_RANGE_GENERAL 0x8000u        /* bit 15 of `hi`: the cluster is not a lean quad (its pairs carry rh::PAIR_GENERAL, bit 28) */
#define RH_RANGE_SUBS_A(lo) (((lo) >> 10) & 7u)
#define RH_RANGE_SUBS_B(lo) (((lo) >> 13) & 7u)
#define RH_RANGE_SUBS(sa, sb) (((sa) << 10) | ((sb) << 13))
#define RH_RANGE_MAX_TILES_X 1024u

struct rh_bin_count_args {
    rh_draw_params p;
    rh_tile_range* ranges;      // [n_clusters]
    uint32_t* block_sums;       // [ceil(n_clusters / 256)]
    uint32_t* wants_clip;       // set to 1 when a triangle crosses a clip plane (rh_bin_count leaves it out, rh_bin_count_clip clips it)
};
// Added to the pair count of a pass by rh_bin_pairs when it left out a triangle that needs clipping: the count then exceeds
// every capacity (the kernels behind it write nothing) and tells the host to run the pass with the clipping kernels.
#define RH_WANTS_CLIP (1ull << 40)

// ---- binning queued in three launches ------------------------------------------------------------------------------------
// Once the host has a guess of how many (tile, cluster) pairs a pass makes, the pair array is built and sorted without
// a host round trip and without global scans: every kernel of the chain hands its workgroups their position in the
// output through a chained look-back (a workgroup publishes the count of its own part at once, adds up the counts
// published by its predecessors until it meets one that already knows its prefix, then publishes its own prefix).
// Workgroups take their place in the chain from a ticket counter, so a workgroup only ever waits for workgroups that
// have started before it.  The state below is cleared before every pass.
#define RH_SORT_MAX_PASSES 4            /* 8 bits of the tile number per pass */
#define RH_LOOKBACK_SPINS (1u << 22)    /* polls of one predecessor before a workgroup gives up and reports a fault */
#define RH_PAIR_THREADS 1024            /* rh_bin_pairs: threads of a workgroup, */
#define RH_PAIR_PER_THREAD 2            /* consecutive clusters per thread, */
#define RH_PAIR_PART (RH_PAIR_THREADS * RH_PAIR_PER_THREAD)    /* clusters per workgroup */
#define RH_DIGIT_COPIES 8               /* copies of the digit counts the workgroups of rh_bin_pairs spread their adds over */
struct rh_bin_state {
    unsigned long long total;           // pairs of the pass (all of them, also when they do not fit the arrays)
    uint32_t ticket[1 + RH_SORT_MAX_PASSES];    // [0] pair kernel, [1 + i] sorting pass i
    uint32_t fault;                     // set when a look-back gave up (the results of the pass are then undefined)
    uint32_t wants_clip;                // a triangle of the pass crosses a clip plane (rh_bin_count, rh_bin_count_clip, rh_bin_pairs_clip)
    uint32_t reserved[7];
    uint32_t digit_count[RH_SORT_MAX_PASSES][RH_DIGIT_COPIES][256];    // pairs per value of every 8-bit digit of the tile number (sum of the copies)
};
// status words of the look-back chains: 2 flag bits on top, the count below
#define RH_STATUS_PARTIAL 1u            /* the count of the workgroup's own part */
#define RH_STATUS_PREFIX 2u             /* the count of everything up to and including the workgroup's part */

// rh_bin_pairs (vertex-stage code object): one thread per cluster; runs the vertex function, finds the tiles the
// cluster's bounding box touches, and writes its (tile, cluster) pairs in cluster order at the position the look-back
// gives, while counting the digits of the tile numbers for the sorting passes.  Pairs beyond `capacity` are not
// written; the last workgroup of the chain stores the total in state->total and in *host_total (pinned host memory).
struct rh_bin_pairs_args {
    rh_draw_params p;
    rh_bin_state* state;
    unsigned long long* cluster_status;     // [ceil(n_clusters / RH_PAIR_PART)], cleared
    uint32_t* keys;                         // [capacity] tile of every pair
    uint32_t* vals;                         // [capacity] cluster of every pair
    uint32_t capacity;
    uint32_t sort_passes;
    uint32_t ticketed;                      // 0: all the workgroups fit the device at once and take their index as their place in the chain
    unsigned long long* host_total;         // pinned host memory: [0] receives the total as well, [1] is set on a fault,
                                            // [2] (rh_bin_pairs_clip) receives rh_bin_state::wants_clip
};

// Triangle set-up (its own launch, one thread per (tile, cluster) pair of the sorted pair array): runs the vertex
// function for the cluster's vertices and writes the pair's record: what a fragment needs to find its pixels and its
// varyings, relative to the pair's tile.  Records are in list order, so a tile reads its records as one contiguous run.
struct rh_setup_args {
    rh_draw_params p;
    const uint32_t* clusters;   // [pairs] cluster of every pair, sorted by (tile, cluster)
    const uint32_t* tiles;      // [pairs] tile of every pair
    uint8_t* rec;               // [pairs][rec_bytes]: bounding box, flags, shared diagonal, plane equations
    uint8_t* ext;               // [pairs][ext_bytes]: edge functions and 1/w planes (written for the clusters that need
                                // them) and, in a multisampled pass, the per-sample edge constants
    uint32_t ext_bytes;
    uint32_t n_pairs;           // pairs, or the room for them when `pair_total` is set
    const unsigned long long* pair_total;
    // [tiles] first pair and [tiles] one past the last pair of every tile, cleared beforehand: the threads at the ends of
    // a tile's run write them
    uint32_t* tile_start;
    uint32_t* tile_end;
};

struct rh_attach_dev {
    uint8_t* ptr;               // texel (0,0)
    uint32_t pitch_px;          // row pitch in pixels
    uint32_t format;            // RH_FORMAT_*
    uint32_t load_op;           // RH_LOAD_OP_*
    uint32_t clear_rgba8;       // clear colour already converted to UNORM8, r in byte 0
    float clear_f[4];           // clear colour as f32 (float formats)
    // Multisampled passes: `ptr` is the multisampled texture (the samples of a pixel adjacent, pitch in pixels) and
    // the averaged samples are written to the single-sampled resolve target (null: no resolve).
    uint8_t* resolve_ptr;
    uint32_t resolve_pitch_px;
    uint32_t resolve_format;
    // Capture (4-byte UNORM attachments of single-sampled passes): null, or a buffer that receives the texels of the
    // region [cap_x0, cap_x0 + cap_w) x [cap_y0, cap_y0 + cap_h) as the tile goes back to memory, rows `cap_pitch` bytes
    // apart, the bytes behind a row's texels zeroed -- what a texture-to-buffer copy behind the pass would write.  Every
    // tile then writes its part of the region, also the tiles of a Load attachment that no primitive touches.
    uint8_t* cap_ptr;
    uint32_t cap_pitch;
    uint32_t cap_x0, cap_y0, cap_w, cap_h;
};

struct rh_tile_args {
    rh_draw_params p;
    const uint8_t* rec;         // set-up records of the sorted pairs (rh_setup_args::rec): only those of general pairs are there
    const uint32_t* vals;       // values of the sorted pairs (rh_setup_args::clusters): cluster | rh::PAIR_GENERAL | sub-record << 29
    const uint8_t* ext;
    uint32_t ext_bytes;
    const uint32_t* tile_start; // [tiles]
    const uint32_t* tile_end;   // [tiles]
    rh_attach_dev att[RH_DEV_MAX_TARGETS];
    uint32_t n_att;
    uint32_t src_factor, dst_factor;
    // Which specialisation of the kernel body to run: 0 = every state read at run time; otherwise
    // 1 + (blend_class - 1) * 3 + (texture_class - 1) with blend_class 1 = (One, OneMinusSrcAlpha), 2 = (One, One) and
    // texture_class 1 = every bound texture RGBA8, at least 2x2 and at most 16 MiB, with linear samplers, 2 = the same with
    // nearest samplers, 3 = class 1 and every bound texture has footprint copies (rh_slot_dev::poly).
    uint32_t variant;
    // The pass may be queued before the host knows how many (tile, cluster) pairs it has: the kernels size their work
    // from *pair_total, and when that exceeds `pair_capacity` (the room the host gave the pair arrays) nobody writes
    // anything -- the host notices and runs the pass again with enough room.  Null: the lists are known to fit.
    uint32_t pair_capacity;
    const unsigned long long* pair_total;
    // Workgroups against tiles.  A pass ends with the machine draining: the last tiles run a few to a SIMD, each at the
    // speed of one wave's dependent chain, so the tiles the last workgroups take may be cut into 2, 3, 4 or 8 bands of
    // rows with a workgroup each (same records, the boxes cut to the band; thirds are 11, 11 and 10 rows).  Workgroup
    // j < split2_first takes tile j whole; the next 2 * (split3_first - split2_first) take the halves of tiles
    // split2_first .. split3_first - 1; the next 3 * (split4_first - split3_first) the thirds of tiles split3_first ..
    // split4_first - 1; the next 4 * (split8_first - split4_first) the quarters of tiles split4_first .. split8_first - 1;
    // the rest the eighths of the tiles from split8_first on.  (All relative to rh_draw_params::tile_first; all four =
    // the number of tiles: nothing is cut.)
    uint32_t split2_first, split4_first, split8_first, split3_first;
    // Issue priority of a tile wave (s_setprio): min(3, pairs it has left * rows of its band / prio_quantum), looked at
    // again every round of pairs; 0: priorities are left alone.  The host passes a sixteenth of the pass's average tile
    // (pairs per tile * 32 rows / 16), so a wave runs at the top priority until its last fifth or so.
    uint32_t prio_quantum;
    // Bodies that read footprint copies of both forms choose the form per primitive (PolyCopy::wide_for in rh_shader.h)
    // by the width of the pass's first texture that has copies: floor(log2(width)), and whether it is a power of two.
    uint32_t poly_log2w, poly_pow2;
};
