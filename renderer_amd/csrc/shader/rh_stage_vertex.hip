// Vertex-stage code object: compiled once per vertex source with
//     hipcc --genco --offload-arch=gfx950 -ffp-contract=off -DRH_VERTEX_SOURCE='"<file>.vert.hip"' rh_stage_vertex.hip
// Exports rh_bin_count: runs the vertex function for every cluster (pair of consecutive triangles of one
// instance), and records which screen tiles the cluster's bounding box touches.
#include "rh_shader.h"
#include RH_VERTEX_SOURCE
#include "rh_geometry.h"

extern "C" __device__ const rh_module_info_t rh_module_info = {
    RH_MODULE_MAGIC, RH_DEVICE_ABI_VERSION, RH_MODULE_STAGE_VERTEX, (uint32_t)rh::N_VARYINGS, 0u,
    rh::fnv1a(RH_VERTEX_TAG), 0u, 0u
};

// One thread per cluster.  Writes the cluster's inclusive tile rectangle and the per-workgroup sum of tile counts
// (the scan of those sums gives every workgroup its base offset in the (tile, cluster) pair array).
extern "C" __global__ __launch_bounds__(256) void rh_bin_count(rh_bin_count_args a) {
    const rh_draw_params& p = a.p;
    const uint32_t cluster = blockIdx.x * 256u + threadIdx.x;
    uint32_t count = 0;
    if (cluster < p.n_clusters) {
        const uint32_t instance = cluster / p.cpi, local = cluster - instance * p.cpi;
        int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t tri = local * 2 + k;
            if (tri >= p.tpi) break;
            uint32_t idx[3]; rh::tri_vertices(p.topology, tri, idx);
            rh::SnappedVertex v0 = rh::run_vertex(p, idx[0], instance);
            rh::SnappedVertex v1 = rh::run_vertex(p, idx[1], instance);
            rh::SnappedVertex v2 = rh::run_vertex(p, idx[2], instance);
            int x0, y0, x1, y1;
            if (rh::tri_pixel_bbox(p, v0, v1, v2, x0, y0, x1, y1)) {
                bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1);
            }
        }
        rh_tile_range r; r.lo = 1u; r.hi = 0u;          // x0 = 1 > x1 = 0: empty
        if (bx1 >= bx0 && by1 >= by0) {
            uint32_t tx0 = (uint32_t)bx0 / RH_TILE_W, ty0 = (uint32_t)by0 / RH_TILE_H;
            uint32_t tx1 = (uint32_t)bx1 / RH_TILE_W, ty1 = (uint32_t)by1 / RH_TILE_H;
            r.lo = tx0 | (ty0 << 16); r.hi = tx1 | (ty1 << 16);
            count = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
        }
        a.ranges[cluster] = r;
    }
    // workgroup sum: wave reduce, then 4 partials through LDS
    __shared__ uint32_t partial[4];
    uint32_t s = count;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63u) == 0) partial[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.block_sums[blockIdx.x] = partial[0] + partial[1] + partial[2] + partial[3];
}
