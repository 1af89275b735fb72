// ff_api.cpp — the C ABI (include/firefly/ff_api.h): tracer state, scene upload, frame rendering, HIP-GL pixel-buffer
// interop and measurement.  Host side of the seam the reference has at kernel.cu:268-298 (upload) and
// kernel.cu:335-344 (per-frame map -> clear -> kernel -> unmap).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime.h>
// (the interop header relies on hip_runtime.h having been included first)
#include <hip/hip_gl_interop.h>

#include "ff_state.h"

using namespace ff;

namespace {

void free_scene(FfState* s)
{
    s->primary_valid = s->last_key_valid = false;
    if (s->d_geoms) (void)hipFree(s->d_geoms);
    if (s->d_tris) (void)hipFree(s->d_tris);
    if (s->d_normals) (void)hipFree(s->d_normals);
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_nodes4) (void)hipFree(s->d_nodes4);
    s->top_count = s->top_depth = 0;
    s->num_scan = 0;
    if (s->d_parent) (void)hipFree(s->d_parent);
    if (s->d_role) (void)hipFree(s->d_role);
    s->d_nodes4 = nullptr;
    s->num_nodes4 = s->max_depth4 = 0;
    s->scene_block_threads = s->lds_cap = 0;
    s->d_geoms = nullptr;
    s->d_tris = nullptr;
    s->d_normals = nullptr;
    s->d_nodes = nullptr;
    s->d_parent = nullptr;
    s->d_role = nullptr;
    s->h_geoms.clear();
    s->slots.clear();
    s->node_capacity = 0;
    s->has_scene = false;
    s->num_geoms = s->num_nodes = s->max_depth = 0;
    s->num_tris = 0;
}

// hipMalloc for the scene arrays (with the test hook above).
hipError_t scene_alloc(FfState* s, void** ptr, size_t bytes)
{
    if (s->alloc_countdown >= 0 && s->alloc_countdown-- == 0) {
        *ptr = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(ptr, bytes);
}

} // namespace

namespace ff {

int ensure_bytes(void** ptr, size_t* cap, size_t need)
{
    if (*cap >= need && *ptr) return FF_OK;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc(ptr, need);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FF_ERR_OOM : FF_ERR_HIP, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    *cap = need;
    return FF_OK;
}

} // namespace ff

namespace {

int check_params(const FfRenderParams* p)
{
    if (!p) return fail(FF_ERR_INVALID_ARG, "render params are null");
    if (p->width <= 0 || p->height <= 0) return fail(FF_ERR_INVALID_ARG, "image size %dx%d is invalid", p->width, p->height);
    if (p->width > 65535 || p->height > 65535) return fail(FF_ERR_INVALID_ARG, "image size %dx%d exceeds 65535 per side", p->width, p->height);
    if (p->bounces < 1 || p->bounces > 255) return fail(FF_ERR_INVALID_ARG, "bounces must be in 1..255 (got %d)", p->bounces);
    if (p->spp < 1 || p->spp >= (1 << 24)) return fail(FF_ERR_INVALID_ARG, "spp must be in 1..2^24-1 (got %d)", p->spp);
    if (p->trace_mode != FF_TRACE_BRUTE_FORCE && p->trace_mode != FF_TRACE_BVH) return fail(FF_ERR_INVALID_ARG, "unknown trace_mode %d", p->trace_mode);
    if (p->shade_mode != FF_SHADE_NORMAL_DEBUG && p->shade_mode != FF_SHADE_DIFFUSE_PATH && p->shade_mode != FF_SHADE_DIFFUSE_PATH_SMOOTH) return fail(FF_ERR_INVALID_ARG, "unknown shade_mode %d", p->shade_mode);
    if (p->grid_mode != FF_GRID_FULL && p->grid_mode != FF_GRID_REFERENCE_FLOOR) return fail(FF_ERR_INVALID_ARG, "unknown grid_mode %d", p->grid_mode);
    if (p->spp_per_launch < 0) return fail(FF_ERR_INVALID_ARG, "spp_per_launch must be >= 0");
    return FF_OK;
}

} // namespace

namespace ff {

int check_render_call(const FfState* s, const FfCamera* camera, const FfRenderParams* params, const char* who)
{
    if (!s) return fail(FF_ERR_INVALID_ARG, "%s: state is null", who);
    if (!camera) return fail(FF_ERR_INVALID_ARG, "%s: camera is null", who);
    const int st = check_params(params);
    if (st != FF_OK) return st;
    if (!s->has_scene) return fail(FF_ERR_NO_SCENE, "%s: no scene uploaded", who);
    return FF_OK;
}

} // namespace ff

namespace {

// Geometry records the BVH kernels keep in LDS: all of them up to kMaxLdsRecords, none beyond (read from global memory).
int lds_records(const FfState* s) { return s->num_geoms > kMaxLdsRecords ? 0 : s->num_geoms; }

// Workgroup size of the BVH kernel for the uploaded scene: the preferred size if the lane-strided traversal stacks
// (4 bytes x workgroup size per tree level) and the geometry records fit the 160 KiB of LDS, else the largest smaller
// instantiation that does; 0 if even 512 threads do not fit (a degenerate, chain-like device-built tree).
int bvh_block_threads(const FfState* s, int preferred, int stack_levels)
{
    const int sizes[3] = { 1024, 768, 512 };
    for (int b : sizes) {
        if (b > preferred) continue;
        if (bvh_lds_bytes(0, stack_levels, b, lds_records(s)) + (s->use_pool ? pool_lds_bytes(b) : 0) <= (size_t)kLdsBudgetBytes) return b;
    }
    return 0;
}

// After every change of the trees (upload, rebuild) or of the records (transform update): choose the BVH kernel's
// workgroup size for the scene, divide the LDS node slots among the meshes (each mesh caches the top of its 4-wide tree:
// its first lds_nodes nodes) and write that into the geometry records, then copy the records to the device.
int finalize_layout(FfState* s)
{
    int nodes4 = 0, depth4 = 0;
    for (size_t i = 0; i < s->slots.size(); ++i) {
        if (s->h_geoms[i].type != FF_GEOM_TRIANGLEMESH || s->slots[i].node4_count == 0) continue;
        nodes4 += s->slots[i].node4_count;
        depth4 = std::max(depth4, s->slots[i].depth4);
    }
    s->num_nodes4 = nodes4;
    s->max_depth4 = depth4;
    // Scenes of more than kChunkGeometries geometries: the tree over the geometries' world boxes (rebuilt here because
    // transforms and refits move those boxes; a few hundred records take microseconds).
    // It lives behind the meshes' trees in the 4-wide node array (the upload reserved num_geoms nodes there).
    s->top_count = s->top_depth = 0;
    if (s->num_geoms > kChunkGeometries) {
        std::vector<BvhNode> top2;
        std::vector<Bvh4Node> top4;
        s->num_scan = s->sw.no_scan_planes ? 0 : count_scan_planes(s->h_geoms, s->num_quads);
        build_geometry_tree(s->h_geoms, top2, s->num_scan);
        s->top_depth = collapse_geometry_tree(top2, top4);
        s->top_count = (int)top4.size();
        if (top4.size() > (size_t)s->num_geoms) return fail(FF_ERR_HIP, "geometry tree of %zu nodes for %d geometries", top4.size(), s->num_geoms);
        FF_HIP(hipMemcpyAsync(s->d_nodes4 + s->node_capacity, top4.data(), top4.size() * sizeof(Bvh4Node), hipMemcpyHostToDevice, s->stream));
        FF_HIP(hipStreamSynchronize(s->stream)); // (`top4` goes out of scope)
    }
    // the axis-aligned walls among the planes every query screens (all planes of a small scene, the leading num_scan of a big one)
    build_wall_table(s->h_geoms.data(), s->sw.no_wall_table ? 0 : (s->num_geoms > kChunkGeometries ? s->num_scan : s->num_quads), s->walls, !s->sw.no_wall_pairs);
    if (s->num_geoms <= kChunkGeometries) add_mesh_boxes(s->h_geoms.data(), s->num_planes, s->num_geoms, s->walls);
    // one entry per visited node above the cursor (inner_step) plus a spare; in big scenes the pending entries of the
    // geometry tree sit below a mesh's own
    s->stack_entries = depth4 + 1 + (s->top_depth > 0 ? s->top_depth + 1 : 0);
    if (!s->setup_threshold_forced) {
        // Traversal time slice, in inner-node rounds (trace_bvh_kernel): long enough for most queries of the scene's biggest
        // tree to finish inside one slice.  Measured best on one MI355X (round 3's kernel, profiles/r03_r_slice_sweep.txt): 7 for C2
        // (1 200 nodes), 12 for the 983 040-triangle sphere (180 000 nodes); both sit on 1.5 log4(nodes) - 0.7.
        int biggest = 1;
        for (size_t i = 0; i < s->slots.size(); ++i) biggest = std::max(biggest, s->slots[i].node4_count);
        s->setup_threshold = std::max(4, std::min(24, (int)std::lround(1.5 * std::log((double)biggest) / std::log(4.0) - 0.7)));
        s->setup_threshold = std::min(32, s->setup_threshold + 2 * s->top_depth); // big scenes: plus the walk through the geometry tree
    }
    // The job-pool kernel (trace_pool_kernel; scenes of up to kChunkGeometries geometries) parks its traversal jobs in LDS: 48 bytes
    // per thread + the queue, taken off the node cache.  Its stacks keep a fixed few levels in LDS whether or not the trees fit
    // (kPoolStackLevels; the deeper entries go to the global spill area): with the jobs in LDS a tree of more than a few hundred
    // nodes does not fit anyway, and a level costs 36 nodes.
    s->use_pool = s->sw.pool > 0 && s->num_geoms <= kChunkGeometries && nodes4 > 0; // (FF_POOL=1; the library's own choice is the lane-owned kernel until the pool kernel beats it)
    constexpr int kPoolStackLevels = 5;
    const int pool_levels = std::min(s->stack_entries, s->sw.pool_stack_levels > 0 ? s->sw.pool_stack_levels : kPoolStackLevels);
    s->scene_block_threads = bvh_block_threads(s, s->block_threads, s->use_pool ? pool_levels : s->stack_entries);
    if (s->use_pool && s->scene_block_threads != 1024) { // (the experiment is instantiated for 1 024 threads only)
        s->use_pool = false;
        s->scene_block_threads = bvh_block_threads(s, s->block_threads, s->stack_entries);
    }
    // (a tree too deep even for 512 threads still uploads: brute-force rendering works, BVH rendering reports it)
    const int block = s->scene_block_threads > 0 ? s->scene_block_threads : kBlockThreads;
    // LDS holds tree nodes and the lanes' traversal stacks.  A stack level costs 4 bytes x workgroup size = 36 nodes at 1 024
    // threads, and the deepest levels are hardly ever reached: when the scene's trees would fit LDS whole but for the stacks, the
    // stacks keep as many levels as are left (at least kMinLdsStack) and the deeper entries go to global memory (stack_push /
    // stack_pop).  The benchmark scene: 1 202 nodes + 5 of 11 levels instead of 1 040 nodes + 11 levels; a node fetched from
    // global memory costs a query hundreds of cycles, a spilled stack entry is rare.  Trees that do not fit anyway keep the
    // whole stack in LDS.
    constexpr int kMinLdsStack = 4;
    const size_t reserve = s->use_pool ? pool_lds_bytes(block) : 0;
    s->stack_lds_levels = s->stack_entries;
    if (s->use_pool) {
        s->stack_lds_levels = pool_levels;
        // (where everything fits with more levels, take them)
        while (s->stack_lds_levels < s->stack_entries && max_lds_nodes(s->stack_lds_levels + 1, block, lds_records(s), reserve) >= nodes4) ++s->stack_lds_levels;
    } else if (s->scene_block_threads > 0 && !s->sw.no_stack_spill) {
        const int want = nodes4 + s->top_count;
        if (want > max_lds_nodes(s->stack_entries, block, lds_records(s))) {
            int levels = s->stack_entries;
            while (levels > kMinLdsStack && max_lds_nodes(levels, block, lds_records(s)) < want) --levels;
            if (max_lds_nodes(levels, block, lds_records(s)) >= want) s->stack_lds_levels = levels;
        }
    }
    int cap = s->scene_block_threads > 0 ? std::max(0, max_lds_nodes(s->stack_lds_levels, block, lds_records(s), reserve)) : 0;
    if (s->sw.lds_node_cap >= 0) cap = std::min(cap, s->sw.lds_node_cap); // FF_DEBUG_LDS_NODE_CAP: experiments on partial residency
    s->lds_cap = std::min(cap, nodes4 + s->top_count);
    // the geometry tree first (every query of a big scene starts there), the meshes share the rest
    s->top_lds_count = std::min(s->top_count, cap);
    const int mesh_cap = cap - s->top_lds_count;
    // Every mesh first gets the top of its tree up to kSmallTree nodes (a crowd of small objects: their whole trees; a query
    // that enters one must not pay a global fetch for a three-node tree), then the big trees share what is left in proportion
    // to their sizes.
    constexpr int kSmallTree = 8;
    std::vector<int> share(s->h_geoms.size(), 0);
    int left = mesh_cap, wanting = 0;
    for (size_t i = 0; i < s->h_geoms.size(); ++i) {
        if (s->h_geoms[i].type != FF_GEOM_TRIANGLEMESH || s->slots[i].node4_count == 0) continue;
        share[i] = std::min(std::min(s->slots[i].node4_count, kSmallTree), left);
        left -= share[i];
        wanting += s->slots[i].node4_count - share[i];
    }
    const int pool = left;
    for (size_t i = 0; i < s->h_geoms.size() && wanting > 0; ++i) {
        if (s->h_geoms[i].type != FF_GEOM_TRIANGLEMESH || s->slots[i].node4_count == 0) continue;
        const int want = s->slots[i].node4_count - share[i];
        share[i] += wanting <= pool ? want : (int)((int64_t)pool * want / wanting);
    }
    int next = s->top_lds_count;
    for (size_t i = 0; i < s->h_geoms.size(); ++i) {
        GeomRecord& r = s->h_geoms[i];
        r.node4_first = 0;
        r.lds_nodes = 0;
        r.wmin[3] = 0.0f;
        if (r.type != FF_GEOM_TRIANGLEMESH || s->slots[i].node4_count == 0) continue;
        r.node4_first = s->slots[i].node_first;
        r.lds_nodes = std::min(s->slots[i].node4_count, share[i]);
        std::memcpy(&r.wmin[3], &next, sizeof(int));
        next += r.lds_nodes;
    }
    FF_HIP(hipMemcpyAsync(s->d_geoms, s->h_geoms.data(), s->h_geoms.size() * sizeof(GeomRecord), hipMemcpyHostToDevice, s->stream));
    FF_HIP(hipStreamSynchronize(s->stream));
    return FF_OK;
}

// Derive mesh slot `gi`'s 4-wide tree from its binary tree (after a build: with_info, one stream synchronisation; after a
// refit: boxes only, asynchronous).
int collapse_slot(FfState* s, size_t gi, bool with_info)
{
    FfState::MeshSlot& slot = s->slots[gi];
    if (slot.node_count <= 0) return FF_OK;
    int st = FF_OK;
    if (!slot.parents_linked) {
        st = gpu_link_parents(s->stream, s->d_nodes, slot.node_first, slot.node_count, s->d_parent + slot.node_first);
        if (st != FF_OK) return st;
        slot.parents_linked = true;
    }
    Collapse4Info info;
    st = gpu_collapse_mesh(s->stream, s->scratch, s->d_nodes, slot.node_first, slot.node_count, s->d_parent + slot.node_first, s->d_nodes4, slot.node_first, s->d_role + slot.node_first,
                           with_info ? &info : nullptr);
    if (st != FF_OK) return st;
    if (with_info) {
        if (info.node_count >= (1 << 22)) return fail(FF_ERR_UNSUPPORTED, "a mesh's tree has %d 4-wide nodes; the traversal stack encodes at most %d", info.node_count, (1 << 22) - 1);
        slot.node4_count = info.node_count;
        slot.depth4 = info.depth;
    }
    return FF_OK;
}

} // namespace

namespace ff {

// Core of every render entry point (declared in ff_state.h).
int render_enqueue(FfState* s, const FfCamera* camera, const FfRenderParams* prm, int strip_rows, int part, int num_parts, int local_rows,
                   unsigned char* rgb8_dev, float* radiance_dev, int x0, int y0, int win_w)
{
    const int W = prm->width, H = prm->height;
    if (win_w < 0) win_w = W;
    const size_t local_pixels = (size_t)local_rows * (size_t)win_w;
    s->stats = FfStats();
    s->pending = false;
    if (local_pixels == 0) return FF_OK;

    const bool debug = prm->shade_mode == FF_SHADE_NORMAL_DEBUG;
    const int spp = debug ? 1 : prm->spp;
    const int bounces = debug ? 1 : prm->bounces;
    // Samples are accumulated in blocks (a multiple of 64, at most 16 blocks per pixel up to 1024 spp and beyond): a
    // block sums its samples sequentially, the combine kernel adds a pixel's blocks in order.  (pixel, block) is the unit
    // of work, which keeps the persistent lanes balanced at the end of a frame and when a frame is split over GPUs.
    const int block_spp = 64 * ((spp + 1023) / 1024);
    const int num_blocks = (spp + block_spp - 1) / block_spp;
    int blocks_per_launch = num_blocks;
    if (prm->spp_per_launch > 0 && !debug) blocks_per_launch = std::max(1, (prm->spp_per_launch + block_spp - 1) / block_spp);
    const int launches = (num_blocks + blocks_per_launch - 1) / blocks_per_launch;

    KParams k;
    std::memset(&k, 0, sizeof k);
    FfMat4 cm;
    ff_camera_ray_matrix(camera, &cm);
    std::memcpy(k.cam_c0, &cm.m[0], 16);
    std::memcpy(k.cam_c1, &cm.m[4], 16);
    std::memcpy(k.cam_c2, &cm.m[8], 16);
    std::memcpy(k.cam_c3, &cm.m[12], 16);
    k.cam_pos[0] = camera->m_position.x;
    k.cam_pos[1] = camera->m_position.y;
    k.cam_pos[2] = camera->m_position.z;
    k.far_clip = camera->m_farClip;
    k.screen_w = camera->m_screenWidth;
    k.screen_h = camera->m_screenHeight;
    k.width = W;
    k.height = H;
    k.xlim = W;
    k.ylim = H;
    if (prm->grid_mode == FF_GRID_REFERENCE_FLOOR) { // kernel.cu:306-309
        k.xlim = (W / 16) * 16;
        k.ylim = (H / 16) * 16;
    }
    k.strip_rows = strip_rows;
    k.part = part;
    k.num_parts = num_parts;
    k.local_rows = local_rows;
    k.x0 = x0;
    k.y0 = y0;
    k.local_width = win_w;
    k.tiles_per_row = (win_w + 7) / 8;
    const uint64_t tiles = (uint64_t)k.tiles_per_row * (uint64_t)((local_rows + 7) / 8);
    if (tiles * 64 * (uint64_t)(num_blocks + 64) >= (1ull << 31)) return fail(FF_ERR_INVALID_ARG, "image too large for the work queue");
    k.pix_items = (unsigned)(tiles * 64);
    k.bounces = bounces;
    k.spp_total = spp;
    k.block_spp = block_spp;
    k.num_blocks = num_blocks;
    {
        int bst = ensure_bytes((void**)&s->d_blocksums, &s->blocksums_bytes, (size_t)k.pix_items * (size_t)num_blocks * 4 * sizeof(float));
        if (bst != FF_OK) return bst;
    }
    k.blocksums = reinterpret_cast<float4*>(s->d_blocksums);
    k.key = (unsigned)prm->seed ^ (unsigned)(prm->seed >> 32);
    k.shade_mode = prm->shade_mode;
    k.setup_threshold = s->setup_threshold;
    k.leaf_threshold = s->leaf_threshold;
    k.num_geoms = s->num_geoms;
    k.num_planes = s->num_planes;
    k.num_quads = s->num_quads;
    k.has_specular = s->has_specular ? 1 : 0;
    k.geoms = s->d_geoms;
    k.tris = s->d_tris;
    k.trinormals = prm->shade_mode == FF_SHADE_DIFFUSE_PATH_SMOOTH ? reinterpret_cast<const float4*>(s->d_normals) : nullptr;
    k.nodes4 = s->d_nodes4;
    int block_threads = kBlockThreads;
    if (prm->trace_mode == FF_TRACE_BVH) {
        block_threads = s->scene_block_threads;
        if (block_threads == 0)
            return fail(FF_ERR_UNSUPPORTED, "4-wide BVH of depth %d does not fit the LDS traversal stack (512 threads x %d levels + %d geometry records > 160 KiB); "
                        "upload with FF_BUILD_HOST_SAH or render with FF_TRACE_BRUTE_FORCE", s->max_depth4, s->max_depth4 + 1, s->num_geoms);
    }
    if (s->sw.lds_fill) {
        const unsigned long words = s->sw.lds_fill_words, pattern = s->sw.lds_fill_pattern;
        if (prm->trace_mode == FF_TRACE_BVH) {
            const size_t bytes = bvh_lds_bytes(s->lds_cap, s->stack_lds_levels, block_threads, lds_records(s)); // (the pool behind it is initialised by the kernel)
            k.debug_lds_words = (unsigned)std::min<size_t>(words, bytes / 4);
            k.debug_lds_pattern = (unsigned)pattern;
        }
    }
    k.stack_depth = s->stack_lds_levels;
    k.stack_spill = nullptr;
    k.lds_nodes = s->lds_cap;
    k.top_first = (int)s->node_capacity;
    k.top_lds_first = 0;
    k.top_lds_count = s->top_lds_count;
    k.num_scan = s->num_scan;
    k.walls = s->walls;
    k.emitter_mask = 0u;
    k.cut_last = 0;
    if (prm->trace_mode == FF_TRACE_BVH && !debug && !s->sw.no_last_bounce_cut) {
        // The last segment of a path adds radiance only when it ends on an emitter.  If every emitter is one of the analytic records
        // all queries screen before anything else (all planes and spheres of a small scene, the scan planes of a big one), a
        // last-bounce query that holds no emitter after that screening is over (scan_records).
        const int screened = s->num_geoms > kChunkGeometries ? s->num_scan : s->num_planes;
        bool all_screened = true;
        for (int i = 0; i < s->num_geoms; ++i) {
            if (s->h_geoms[i].bxdf_type != FF_BXDF_EMITTER) continue;
            if (i < screened && i < 32) k.emitter_mask |= 1u << i;
            else all_screened = false;
        }
        k.cut_last = all_screened ? 1 : 0;
    }
    k.cull_mask = nullptr;
    s->pending_culled_rays_per_pixel = 0;
    bool cull = false;
    {
        // the padded world box around everything (the records' own boxes are padded); a camera outside it lets the work queue
        // drop the pixels whose primary ray misses it (acquire_pixel)
        float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        for (const GeomRecord& r : s->h_geoms) {
            if (!(r.wmin[0] <= r.wmax[0])) continue;
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], r.wmin[a]); mx[a] = std::max(mx[a], r.wmax[a]); }
        }
        const float cp[3] = { camera->m_position.x, camera->m_position.y, camera->m_position.z };
        bool outside = false, valid = true;
        for (int a = 0; a < 3; ++a) {
            k.scene_min[a] = mn[a];
            k.scene_max[a] = mx[a];
            valid = valid && mn[a] <= mx[a] && std::fabs(mn[a]) < 1e30f && std::fabs(mx[a]) < 1e30f;
            outside = outside || !(cp[a] >= mn[a] && cp[a] <= mx[a]);
        }
        cull = prm->trace_mode == FF_TRACE_BVH && valid && outside && !s->sw.no_primary_cull;
    }
    k.rgb8 = rgb8_dev;
    k.radiance = radiance_dev;
    k.queue = s->d_queue;
    k.counters = s->d_counters;
    k.timeline = nullptr;
    k.timeline_ticks = 1;
    if (s->collect_stats && s->timeline_bucket_us > 0) {
        const size_t rows = (size_t)s->num_cus * 2 * (kBlockThreads / 64); // one row per wave of the largest grid
        if (!s->d_timeline) FF_HIP(hipMalloc((void**)&s->d_timeline, rows * kTimelineBuckets * sizeof(unsigned)));
        FF_HIP(hipMemsetAsync(s->d_timeline, 0, rows * kTimelineBuckets * sizeof(unsigned), s->stream));
        k.timeline = s->d_timeline;
        k.timeline_ticks = (unsigned)s->timeline_bucket_us * 100u; // the wall clock ticks at 100 MHz
    }

    const int blocks_per_cu = prm->trace_mode == FF_TRACE_BVH ? 1 : 2;
    int grid = s->num_cus * blocks_per_cu;
    const uint64_t max_useful = ((uint64_t)k.pix_items * (uint64_t)num_blocks + (uint64_t)block_threads - 1) / (uint64_t)block_threads;
    if ((uint64_t)grid > max_useful) grid = (int)max_useful;
    if (grid < 1) grid = 1;
    k.primary_hits = nullptr;
    k.reuse_quorum = s->sw.reuse_quorum;
    // job-pool kernel: the defaults are where the same-box sweeps put them (profiles/r04_*pool_sweep*)
    k.pool_quorum = s->sw.pool_quorum > 0 ? s->sw.pool_quorum : 40;
    k.pool_quorum_min = std::min(k.pool_quorum, s->sw.pool_quorum_min > 0 ? s->sw.pool_quorum_min : 16);
    k.pool_refill = s->sw.pool_refill > 0 ? s->sw.pool_refill : 16;
    k.pool_slice = s->sw.pool_slice > 0 ? s->sw.pool_slice : 4;
    k.pool_leave = s->sw.pool_leave >= 0 ? s->sw.pool_leave : 24;
    k.pool_batch_min = s->sw.pool_batch_min > 0 ? s->sw.pool_batch_min : 48;
    // (instrumented launches with a timeline run the lane-owned kernel, which keeps it; the LDS layout serves both)
    const bool pool = s->use_pool && prm->trace_mode == FF_TRACE_BVH && !(s->collect_stats && s->timeline_bucket_us > 0);
    // Every sample of a pixel starts with the same ray (kernel.cu:200-205 has no jitter): a pre-pass traces it once per pixel and the
    // frame's samples start from the stored hit (trace_bvh_kernel).  One slot of 3 x 16 bytes per pixel item.
    // The stored hits are KEPT: the next frame from the same camera, pixel mapping and scene starts from them without a pre-pass
    // (a viewer that accumulates 1-spp frames with the camera at rest, kernel.cu:266,342).  A frame of 2 spp or more always runs on
    // stored hits (the pre-pass pays within the frame: +1.3 % at 2 spp).  A 1-spp frame does when the hits are there already (+8 %),
    // or when the frame before it had the same key - the camera has come to rest, so this frame's pre-pass (-4 % for this frame) is
    // the last one; a one-off 1-spp frame traces its primary rays itself (profiles/r04_j_*).
    const bool reuse_possible = prm->trace_mode == FF_TRACE_BVH && !debug && !s->sw.no_primary_reuse;
    FfState::PrimaryKey key;
    std::memset(&key, 0, sizeof key);
    if (reuse_possible) {
        std::memcpy(key.cam, k.cam_c0, 16 * sizeof(float));
        std::memcpy(key.cam + 16, k.cam_pos, 3 * sizeof(float));
        key.cam[19] = k.far_clip; key.cam[20] = k.screen_w; key.cam[21] = k.screen_h;
        // (the stored normal is the interpolated one when the frame shades with vertex normals: part of the key)
        const int dims[12] = { W, H, k.xlim, k.ylim, strip_rows, part, num_parts, local_rows, x0, y0, win_w, k.trinormals != nullptr ? 1 : 0 };
        std::memcpy(key.dims, dims, sizeof dims);
        key.pix_items = k.pix_items;
    }
    const bool keeping = reuse_possible && !s->sw.no_primary_cache;
    const bool hits_kept = keeping && s->primary_valid && std::memcmp(&key, &s->primary_key, sizeof key) == 0;
    const bool at_rest = keeping && s->last_key_valid && std::memcmp(&key, &s->last_key, sizeof key) == 0;
    const bool reuse = reuse_possible && (spp >= s->sw.reuse_min_spp || hits_kept || at_rest);
    s->last_key = key;
    s->last_key_valid = reuse_possible;
    if (reuse) {
        if (s->primary_cache_bytes < (size_t)3 * (size_t)k.pix_items * sizeof(float4)) s->primary_valid = false; // (the buffer is about to move)
        const int cst = ensure_bytes((void**)&s->d_primary_cache, &s->primary_cache_bytes, (size_t)3 * (size_t)k.pix_items * sizeof(float4));
        if (cst != FF_OK) return cst;
    }
    k.park = nullptr;
    if (pool) {
        // where a lane's own path and query state waits between setup passes (trace_pool_kernel): 7 x 16 bytes per thread of the launch
        const int pst = ensure_bytes((void**)&s->d_park, &s->park_bytes, (size_t)7 * (size_t)grid * (size_t)block_threads * sizeof(float4));
        if (pst != FF_OK) return pst;
        k.park = s->d_park;
    }
    if (prm->trace_mode == FF_TRACE_BVH && s->stack_lds_levels < s->stack_entries) {
        // the stack levels that did not get LDS (finalize_layout): one int per level and thread of the launch
        const int st = ensure_bytes((void**)&s->d_stack_spill, &s->stack_spill_bytes,
                                    (size_t)(s->stack_entries - s->stack_lds_levels) * (size_t)grid * (size_t)block_threads * sizeof(int));
        if (st != FF_OK) return st;
        k.stack_spill = s->d_stack_spill;
    }

    {
        // Work queue (csrc/ff_kernels.hip acquire_pixel): 16 counters in different memory channels, and a wave takes at least
        // queue_chunk consecutive items per atomic.  One counter asked for every item serves about 10^8 requests a second, which
        // held every frame of short items far below the saturated rate (1080p C2, before -> after: the reference's own 1-ray frame
        // 0.40 -> 0.19 ms, path-traced 1 spp 2.98 -> 1.33 ms, 4 spp 7.3 -> 4.8, 16 spp 20.8 -> 18.0; the 1 024-spp frame from the
        // reference's default camera, whose 64-sample items are mostly one-ray paths, 164 -> 84 ms).  With 16 counters the chunk
        // hardly matters for the atomics between 8 and 64 items (profiles/r02_q_queue_sweep.txt); what it decides is WHICH items a
        // wave's lanes hold.  Whole 64-sample blocks: 64 items, i.e. the blocks of four neighbouring pixels (one 8x8 tile when a
        // pixel has one block): the lanes' rays start from the same few surface points and walk the same corner of the trees
        // (1 024 spp, ms with chunks of 4 / 16 / 64 / 128: C4, whose tree lives in L2, 570 / 530 / 491 / 488; C2 977 / 968 / 956 /
        // 959; at 64-512 spp C4 gains 10-14 %, C2 1-2 %: profiles/r02_y_chunk_sweep_whole_blocks.txt).  Shorter items (frames of a few samples): 64
        // samples of work, at most 32 items; there what a wave holds back at the end of the launch counts (1 spp: 1.33 ms with
        // 16-32, 1.36 with 64).
        const int samples_per_item = std::max(1, std::min(k.block_spp, k.spp_total));
        // (the strips of a multi-GPU rank: 32; slowest of eight ranks 125.5 ms against 126.9 with 64 and 125.9 with 16)
        k.queue_chunk = samples_per_item >= 64 ? (num_parts > 1 ? 32u : 64u) : (unsigned)std::min(32, std::max(4, 64 / samples_per_item));
        if (s->sw.queue_chunk > 0) k.queue_chunk = (unsigned)s->sw.queue_chunk;
        k.queue_counters = std::min(kQueueCountersDefault, grid);
        if (s->sw.queue_counters > 0) k.queue_counters = std::min(std::min(kQueueCounters, grid), s->sw.queue_counters);
        // The last items of every counter's share go out exactly as asked for (acquire_pixel: no private stock at the end of a launch):
        // one per lane of the waves that draw from the counter where items are long (sample blocks), an eighth of that where they are
        // short (a 1-spp frame's paths: single-item requests cost an atomic each, which is what the chunks are there to avoid).
        {
            const int waves_per_counter = (grid * (block_threads / 64) + k.queue_counters - 1) / k.queue_counters;
            // (measured on one box, profiles/r04_b_queue_tail.txt: long items 64 per wave; frames that drop most of their items at the
            // queue - a camera outside the scene - lose with a long zone, every dropped item there being a request of its own: 8)
            const int per_wave = s->sw.queue_tail >= 0 ? s->sw.queue_tail : (samples_per_item >= 16 && !cull ? 64 : 8);
            k.queue_tail_items = (unsigned)(waves_per_counter * per_wave);
        }
    }
    hipStream_t st = s->stream;
    // cudaMemset(pbo, 0) of kernel.cu:340: untraced pixels read 0.  With the full grid the combine pass writes every pixel
    // of the window (a missed pixel gets its zero sum), so the clears are only needed for the reference's floor grid.
    if (prm->grid_mode == FF_GRID_REFERENCE_FLOOR) {
        if (rgb8_dev) FF_HIP(hipMemsetAsync(rgb8_dev, 0, local_pixels * 3, st));
        if (radiance_dev) FF_HIP(hipMemsetAsync(radiance_dev, 0, local_pixels * 3 * sizeof(float), st));
    }
    FF_HIP(hipMemsetAsync(s->d_counters, 0, (size_t)(1 + k.queue_counters) * kQueueStride * sizeof(unsigned), st)); // counters and the work queue behind them
    FF_HIP(hipEventRecord(s->ev_begin, st));
    // Fine-grained tail: in the launch that finishes the frame, the last sample block is traced as items of a few samples
    // with per-sample storage (see KParams::tail_samples), so that the launch runs dry on short items.
    k.tail_block = -1;
    const int last_launch_blocks = num_blocks - (launches - 1) * blocks_per_launch;
    // It pays where a launch is short against its items: the ranks of a multi-GPU frame (+3.3 % at eight ranks) and one-GPU
    // frames of up to eight blocks (1080p C2 with / without, ms: 64 spp 66.0 / 71.4, 128 spp 128.4 / 132.3, 256 spp 247.8 / 251.1).
    // It costs a per-sample buffer (2.1 GB written and read back at 1080p), which the 1 024-spp one-GPU frame does not earn back
    // (+0.4 %, and 4x its HBM traffic): off there.  FF_TAIL_GROUP forces it (with that group size) wherever a launch has
    // FF_TAIL_MIN_BLOCKS blocks.  Frames whose buffer would pass 16 GiB render without it (FfStats::flags).
    const bool short_frame = num_parts == 1 && !s->tail_forced && num_blocks <= 8;
    // How many of the frame's last blocks go out that way.  A lane that takes one of the last WHOLE blocks finishes up to two block
    // times later (path lengths vary by that much between pixels); the short items must last that long for the launch to end on
    // them, so a rank's strips - where a block time is 6 % of the launch - hand out TWO blocks in groups (eight ranks 94.5 -> 96 % of
    // ideal, profiles/r04_n_*).  The kernels see one "tail block" of up to 2 x block_spp samples starting at block tail_block; the
    // combine pass adds each block's samples in order.
    // (two where a lane gets fewer than 24 whole blocks in the launch - eight ranks at 1080p and 1 024 spp: 16 -, one where it gets
    // more: there the short items' own overhead outweighs the shorter end, 4 ranks 97.7 -> 97.3 %, 2 ranks 98.9 -> 98.5 %)
    const double blocks_per_lane = (double)k.pix_items * (double)last_launch_blocks / ((double)grid * (double)block_threads);
    const int tail_blocks_wanted = s->sw.tail_blocks > 0 ? s->sw.tail_blocks : (num_parts > 1 && blocks_per_lane < 24.0 ? 2 : 1);
    const int tail_blocks = std::max(1, std::min(std::min(tail_blocks_wanted, 3), last_launch_blocks - 1));
    const int tail_n = spp - (num_blocks - tail_blocks) * block_spp; // samples of the frame's last block(s)
    // group size: the given one (multi-part frames: 32 samples), scaled with the block size beyond 1 024 spp and at least an
    // eighth of the block; short one-GPU frames: a quarter of the block, at least 4 samples
    const int tail_step = short_frame ? std::max(4, (tail_n + 3) / 4) : std::max(s->tail_group_spp * (block_spp / 64), (tail_n + 15) / 16);
    const bool tail_wanted = !debug && prm->trace_mode == FF_TRACE_BVH && s->tail_group_spp > 0 && tail_step < tail_n &&
                             (short_frame || ((num_parts > 1 || s->tail_forced) && last_launch_blocks >= s->tail_min_blocks));
    const size_t tail_bytes = (size_t)k.pix_items * (size_t)tail_n * sizeof(float4); // [sample of the block][pixel item]
    // The buffer stays allocated between frames.  Where the tail is this library's own choice (short one-GPU frames: a viewer at
    // 64 spp) it may take at most 4 GiB and 2 % of the device's memory (1080p at up to 1 024 spp: 2.1 GB; a 64-spp 4K frame would
    // need 8.5 GB and goes without); where the caller set up a multi-GPU frame or forced it, up to 16 GiB (also keeps slot indices
    // in 31 bits; C5's ranks at eight GPUs need 4.2 GB).  FfStats::flags tells which way a frame went.
    const size_t tail_cap = short_frame ? std::min<size_t>(4ull << 30, s->device_mem_bytes / 50) : (16ull << 30);
    const bool tail_mode = tail_wanted && tail_bytes <= tail_cap;
    if (!tail_mode && s->d_tail_samples && s->tail_samples_bytes > (256ull << 20)) {
        // a frame that does not use it gives a large buffer back (the stream has drained: every render call is synchronous)
        (void)hipFree(s->d_tail_samples);
        s->d_tail_samples = nullptr;
        s->tail_samples_bytes = 0;
    }
    s->pending_flags = (tail_mode ? FF_STATS_TAIL_ITEMS : 0u) | (tail_wanted && !tail_mode ? FF_STATS_TAIL_SKIPPED_TOO_LARGE : 0u);
    if (tail_mode) {
        int tst = ensure_bytes((void**)&s->d_tail_samples, &s->tail_samples_bytes, tail_bytes);
        if (tst != FF_OK) return tst;
        k.tail_samples = s->d_tail_samples;
        k.tail_samples_in_block = tail_n;
        // (Grading the groups down to an eighth of a block on multi-GPU ranks was measured: the 8-sample items cost more
        // than the tail they save, 92.3 % instead of 93.3 % of ideal at eight ranks.)
        int g = 0;
        k.tail_start[0] = 0;
        while (k.tail_start[g] < tail_n && g < 16) { k.tail_start[g + 1] = std::min(tail_n, k.tail_start[g] + tail_step); ++g; }
        k.tail_groups = g;
    }
    // (after the tail decision: the tail block's samples are stored one by one and are not part of the cull - a frame whose only
    // block is the tail block has nothing to drop and skips the mask pass.  Dropping tail items too was measured in round 4: a dropped
    // item costs the queue what its 16 samples cost a lane that starts them from a stored miss; no gain, profiles/r04_f_*)
    // (... and so does the reference's own frame - one primary ray per pixel, shaded by its normal: tracing a ray that misses
    // everything costs what the mask pass costs, and the pass is a second launch: 0.089 instead of 0.064 ms at 1080p)
    cull = cull && num_blocks - (tail_mode ? tail_blocks : 0) > 0 && !debug;
    // ... and where a pre-pass stores every pixel's primary hit (below), it marks the pixels that hit NOTHING in the same mask: the
    // exact version of the box test, also for a camera inside the scene's box (an open room seen from within)
    const bool exact_cull = reuse && !s->sw.no_primary_cull && num_blocks - (tail_mode ? tail_blocks : 0) > 0;
    // Can this frame start from the hits (and the mask) the last one stored?  Same camera, same pixel mapping, no change of the scene
    // since (every upload / update clears primary_valid), and a mask there if this frame wants one.
    const bool kept = reuse && hits_kept && (!(cull || exact_cull) || s->primary_has_mask);
    s->pending_mask_reused = s->pending_mask_built = false;
    if (cull || exact_cull) {
        const size_t mask_bytes = ((size_t)k.pix_items / 64 + 2) * sizeof(unsigned long long);
        if (!kept) {
            const int mst = ensure_bytes((void**)&s->d_cull_mask, &s->cull_mask_bytes, mask_bytes);
            if (mst != FF_OK) return mst;
            if (cull) FF_HIP(launch_cull_mask(k, s->d_cull_mask, st));
            else FF_HIP(hipMemsetAsync(s->d_cull_mask, 0, mask_bytes, st));
            s->pending_mask_built = true;
            if (!reuse) s->primary_has_mask = false; // (the mask that went with the stored hits has just been overwritten; the hits stay)
        } else {
            s->pending_mask_reused = true;
        }
        k.cull_mask = s->d_cull_mask;
        // (a culled pixel's whole-block items are dropped; with a fine-grained tail its last block is still traced sample by sample)
        s->pending_culled_rays_per_pixel = (unsigned)(spp - (tail_mode ? tail_n : 0));
    }
    if (reuse && kept) k.primary_hits = s->d_primary_cache;
    if (reuse && !kept) {
        // The pre-pass: the same persistent kernel, one item per pixel, one primary ray each, the hit stored per pixel (settle_hit).  Its
        // rays are not path segments of the frame: the kernel does not count them, and the work queue starts from zero again behind it.  (Always the lane-
        // owned kernel, never instrumented: the frame's own launches are what the statistics describe.)
        KParams kp = k;
        kp.shade_mode = kShadePrimaryPass;
        kp.primary_hits = s->d_primary_cache;
        kp.bounces = 1;
        kp.spp_total = 1;
        kp.block_spp = 1;
        kp.num_blocks = 1;
        kp.block_begin = 0;
        kp.block_end = 1;
        kp.whole_blocks = 1u;
        kp.total_items = kp.pix_items;
        kp.tail_block = -1;
        kp.cull_mask = nullptr; // (every pixel gets its stored hit: a culled pixel's tail items - its last block, traced sample by sample - read it too)
        kp.cull_mask_out = exact_cull ? s->d_cull_mask : nullptr;
        kp.frame_blocks = num_blocks;
        kp.cut_last = 0;
        kp.timeline = nullptr;
        kp.queue_chunk = 32u;
        if (s->sw.queue_chunk > 0) kp.queue_chunk = (unsigned)s->sw.queue_chunk;
        FF_HIP(launch_trace(kp, FF_TRACE_BVH, false, grid, block_threads, st, nullptr, false, /*prepass=*/true));
        FF_HIP(hipMemsetAsync(s->d_queue, 0, (size_t)k.queue_counters * kQueueStride * sizeof(unsigned), st));
        k.primary_hits = s->d_primary_cache;
        s->primary_key = key;
        s->primary_valid = true;
        s->primary_has_mask = cull || exact_cull;
    }
    for (int l = 0; l < launches; ++l) {
        k.block_begin = l * blocks_per_launch;
        k.block_end = std::min(num_blocks, (l + 1) * blocks_per_launch);
        k.whole_blocks = (unsigned)(k.block_end - k.block_begin);
        k.total_items = k.pix_items * k.whole_blocks;
        if (tail_mode && l == launches - 1) {
            const unsigned groups = (unsigned)k.tail_groups;
            k.tail_block = num_blocks - tail_blocks;
            k.whole_blocks = (unsigned)(k.block_end - tail_blocks - k.block_begin);
            k.tail_first_item = k.pix_items * k.whole_blocks;
            k.total_items = k.tail_first_item + k.pix_items * groups;
        }
        if (l > 0) FF_HIP(hipMemsetAsync(s->d_queue, 0, (size_t)k.queue_counters * kQueueStride * sizeof(unsigned), st));
        FF_HIP(launch_trace(k, prm->trace_mode, s->collect_stats, grid, block_threads, st, &s->last_kernel_name, pool));
    }
    FF_HIP(launch_combine(k, st));
    FF_HIP(hipEventRecord(s->ev_end, st));
    FF_HIP(hipMemcpyAsync(s->h_counters, s->d_counters, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (k.timeline) {
        const size_t rows = (size_t)s->num_cus * 2 * (kBlockThreads / 64);
        s->h_timeline_rows.resize(rows * kTimelineBuckets);
        FF_HIP(hipMemcpyAsync(s->h_timeline_rows.data(), s->d_timeline, rows * kTimelineBuckets * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    }
    s->pending = true;
    s->pending_launches = launches;
    return FF_OK;
}

int render_finish(FfState* s)
{
    if (!s->pending) return FF_OK; // nothing was enqueued (an empty part)
    s->pending = false;
    FF_HIP(hipStreamSynchronize(s->stream));
    float ms = 0.f;
    FF_HIP(hipEventElapsedTime(&ms, s->ev_begin, s->ev_end));
    unsigned long long c[32];
    std::memcpy(c, s->h_counters, sizeof c);
    std::memcpy(s->raw_counters, c, sizeof c);
    if (c[0] != 0)
        return fail(FF_ERR_HIP, "the traversal loop guard cut %llu queries short (a malformed or absurdly deep tree): the frame is not valid", c[0]);
    s->stats.rays_traced = 0;
    for (int j = 0; j < kRaySlots; ++j) s->stats.rays_traced += s->h_counters[kRaySlotStride * (kRaySlotFirst + j)];
    s->stats.rays_answered = 0;
    {
        // the rays of culled pixels: counted by the passes that built the mask - or, for a frame that took the mask over from the
        // last one, as many pixels as that frame counted
        unsigned long long culled = 0;
        for (int j = 0; j < kRaySlots; ++j) culled += s->h_counters[kCulledPixelsWord + kRaySlotStride * j];
        if (s->pending_mask_reused) culled = s->primary_culled_pixels;
        else if (s->pending_mask_built) s->primary_culled_pixels = culled;
        s->stats.rays_answered += culled * s->pending_culled_rays_per_pixel;
    }
    for (int j = 0; j < kRaySlots; ++j) s->stats.rays_answered += s->h_counters[kAnsweredWord + kRaySlotStride * j]; // + repeated primaries
    s->stats.rays_cut_short = 0;
    for (int j = 0; j < kRaySlots; ++j) s->stats.rays_cut_short += s->h_counters[kCutShortWord + kRaySlotStride * j];
    s->raw_counters[0] = s->stats.rays_traced;
    s->stats.nodes_visited = c[1];
    s->stats.tris_tested = c[2];
    s->stats.planes_tested = c[3];
    s->stats.kernel_ms = ms;
    s->stats.kernel_launches = (uint32_t)s->pending_launches;
    s->stats.flags = s->pending_flags;
    s->stats.scene_bytes_nodes = (uint64_t)s->num_nodes4 * sizeof(Bvh4Node);
    s->stats.scene_bytes_tris = s->num_tris * sizeof(TriRecord);
    return FF_OK;
}

} // namespace ff

namespace {

int render_local(FfState* s, const FfCamera* camera, const FfRenderParams* prm, int strip_rows, int part, int num_parts, int local_rows,
                 unsigned char* rgb8_dev, float* radiance_dev, int x0 = 0, int y0 = 0, int win_w = -1)
{
    const int st = render_enqueue(s, camera, prm, strip_rows, part, num_parts, local_rows, rgb8_dev, radiance_dev, x0, y0, win_w);
    if (st != FF_OK) return st;
    return render_finish(s);
}

} // namespace

extern "C" {

namespace {
// The experiment switches of FfState::Switches, from the environment (ff_create; ff_debug_reload_switches).
void read_switches(FfState* s)
{
    FfState::Switches w;
    w.no_last_bounce_cut = std::getenv("FF_NO_LAST_BOUNCE_CUT") != nullptr;
    w.no_primary_cull = std::getenv("FF_NO_PRIMARY_CULL") != nullptr;
    w.no_primary_reuse = std::getenv("FF_NO_PRIMARY_REUSE") != nullptr;
    w.no_primary_cache = std::getenv("FF_NO_PRIMARY_CACHE") != nullptr;
    if (const char* e = std::getenv("FF_TAIL_BLOCKS")) w.tail_blocks = std::max(1, std::min(3, std::atoi(e)));
    if (const char* e = std::getenv("FF_REUSE_MIN_SPP")) w.reuse_min_spp = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("FF_REUSE_QUORUM")) w.reuse_quorum = std::max(1, std::min(65, std::atoi(e)));
    if (const char* e = std::getenv("FF_QUEUE_CHUNK")) w.queue_chunk = std::max(1, std::min(4096, std::atoi(e)));
    if (const char* e = std::getenv("FF_QUEUE_COUNTERS")) w.queue_counters = std::max(1, std::min(kQueueCounters, std::atoi(e)));
    w.no_wall_table = std::getenv("FF_NO_WALL_TABLE") != nullptr;
    w.no_wall_pairs = std::getenv("FF_NO_WALL_PAIRS") != nullptr;
    w.no_stack_spill = std::getenv("FF_NO_STACK_SPILL") != nullptr;
    w.no_scan_planes = std::getenv("FF_NO_SCAN_PLANES") != nullptr;
    if (const char* e = std::getenv("FF_DEBUG_LDS_FILL")) w.lds_fill = std::sscanf(e, "%lu,%lx", &w.lds_fill_words, &w.lds_fill_pattern) == 2;
    if (const char* e = std::getenv("FF_POOL")) w.pool = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("FF_POOL_QUORUM")) w.pool_quorum = std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("FF_POOL_QUORUM_MIN")) w.pool_quorum_min = std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("FF_POOL_REFILL")) w.pool_refill = std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("FF_POOL_SLICE")) w.pool_slice = std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("FF_POOL_BATCH_MIN")) w.pool_batch_min = std::max(1, std::min(1024, std::atoi(e)));
    if (const char* e = std::getenv("FF_DEBUG_LDS_NODE_CAP")) w.lds_node_cap = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("FF_QUEUE_TAIL")) w.queue_tail = std::max(0, std::min(4096, std::atoi(e)));
    if (const char* e = std::getenv("FF_POOL_LEAVE")) w.pool_leave = std::max(0, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("FF_POOL_STACK_LEVELS")) w.pool_stack_levels = std::max(1, std::min(64, std::atoi(e)));
    s->sw = w;
}
} // namespace

int ff_debug_reload_switches(FfState* s)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_debug_reload_switches: state is null");
    read_switches(s);
    s->primary_valid = s->last_key_valid = false; // (what the stored hits and their mask were computed under may just have changed)
    return FF_OK;
}

int ff_create(FfState** out_state, int device_id)
{
    clear_error();
    if (!out_state) return fail(FF_ERR_INVALID_ARG, "ff_create: out_state is null");
    *out_state = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(FF_ERR_NO_DEVICE, "ff_create: no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(FF_ERR_INVALID_ARG, "ff_create: device %d out of range (0..%d)", device_id, count - 1);
    FF_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    FF_HIP(hipGetDeviceProperties(&prop, device_id));
    FfState* s = new (std::nothrow) FfState();
    if (!s) return fail(FF_ERR_OOM, "ff_create: out of host memory");
    s->device = device_id;
    s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    s->device_mem_bytes = prop.totalGlobalMem;
    if (const char* bt = std::getenv("FF_BLOCK_THREADS")) {
        const int v = std::atoi(bt);
        if (v == 512 || v == 768 || v == 1024) {
            s->block_threads = v;
        }
    }
    read_switches(s);
    if (const char* e = std::getenv("FF_DEBUG_FAIL_ALLOC")) s->debug_fail_alloc = std::atoi(e);
    if (const char* e = std::getenv("FF_DEBUG_TIMELINE_US")) s->timeline_bucket_us = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("FF_TAIL_MIN_BLOCKS")) s->tail_min_blocks = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("FF_TAIL_GROUP")) {
        s->tail_group_spp = std::max(0, std::min(64, std::atoi(e)));
        s->tail_forced = true;
    }
    if (const char* e = std::getenv("FF_SETUP_THRESHOLD")) {
        s->setup_threshold = std::max(0, std::min(1 << 14, std::atoi(e)));
        s->setup_threshold_forced = true;
    }
    if (const char* e = std::getenv("FF_LEAF_THRESHOLD")) s->leaf_threshold = std::max(1, std::min(64, std::atoi(e)));
    hipError_t pe = prepare_kernels();
    if (pe != hipSuccess) {
        delete s;
        return fail(FF_ERR_HIP, "ff_create: kernel preparation failed: %s (is this a gfx950 device?)", hipGetErrorString(pe));
    }
    if (hipMalloc((void**)&s->d_counters, (size_t)(1 + kQueueCounters) * kQueueStride * sizeof(unsigned)) != hipSuccess || // counters, then the work-queue counters 4 KiB apart
        hipHostMalloc((void**)&s->h_counters, kCounterWords * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&s->ev_begin) != hipSuccess || hipEventCreate(&s->ev_end) != hipSuccess) {
        ff_destroy(s);
        return fail(FF_ERR_HIP, "ff_create: allocating work buffers failed");
    }
    s->d_queue = reinterpret_cast<unsigned*>(s->d_counters) + kQueueStride;
    *out_state = s;
    return FF_OK;
}

int ff_destroy(FfState* s)
{
    if (!s) return FF_OK;
    (void)hipSetDevice(s->device);
    if (s->pbo_resource) (void)hipGraphicsUnregisterResource(s->pbo_resource);
    dist_release(s);
    free_scene(s);
    if (s->d_stage) (void)hipFree(s->d_stage);
    if (s->d_tail_samples) (void)hipFree(s->d_tail_samples);
    if (s->d_stack_spill) (void)hipFree(s->d_stack_spill);
    if (s->d_cull_mask) (void)hipFree(s->d_cull_mask);
    if (s->d_primary_cache) (void)hipFree(s->d_primary_cache);
    if (s->d_park) (void)hipFree(s->d_park);
    if (s->d_accum) (void)hipFree(s->d_accum);
    if (s->d_frame) (void)hipFree(s->d_frame);
    if (s->d_mean) (void)hipFree(s->d_mean);
    free_build_scratch(s->scratch);
    if (s->d_blocksums) (void)hipFree(s->d_blocksums);
    if (s->d_rgb8) (void)hipFree(s->d_rgb8);
    if (s->d_radiance) (void)hipFree(s->d_radiance);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->d_timeline) (void)hipFree(s->d_timeline);
    if (s->h_counters) (void)hipHostFree(s->h_counters);
    if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
    if (s->ev_end) (void)hipEventDestroy(s->ev_end);
    delete s;
    return FF_OK;
}

int ff_set_stream(FfState* s, void* hip_stream)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_set_stream: state is null");
    s->stream = (hipStream_t)hip_stream;
    return FF_OK;
}

namespace {

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Copy a caller triangle array into the device staging buffer (input of the device builder and of refit).
int stage_triangles(FfState* s, const FfTriangle* tris, int count, double* copy_ms)
{
    const auto t0 = std::chrono::steady_clock::now();
    int st = ensure_bytes((void**)&s->d_stage, &s->stage_bytes, (size_t)count * sizeof(FfTriangle));
    if (st != FF_OK) return st;
    FF_HIP(hipMemcpyAsync(s->d_stage, tris, (size_t)count * sizeof(FfTriangle), hipMemcpyHostToDevice, s->stream));
    FF_HIP(hipStreamSynchronize(s->stream));
    *copy_ms += ms_since(t0);
    return FF_OK;
}

// Leaf size of the device builders: subtrees of up to this many triangles collapse into one leaf (FF_GPU_LEAF overrides).
int device_leaf_tris(const BvhBuildParams& bp)
{
    int v = bp.device_leaf_tris;
    if (const char* e = std::getenv("FF_GPU_LEAF")) v = std::atoi(e);
    return std::max(1, std::min(bp.max_leaf_tris, v));
}

// A mesh that fits one leaf: the host builder's single node, re-based into the device arrays.
int place_single_leaf_mesh(FfState* s, const FfTriangle* tris, int count, const BvhBuildParams& bp, int tri_first, int node_base, int* out_nodes, int* out_depth)
{
    std::vector<BvhNode> tn;
    std::vector<TriRecord> tt;
    std::vector<TriNormals> nn;
    int depth = 0;
    build_mesh_bvh(tris, count, bp, tn, tt, &depth, &nn);
    for (BvhNode& nd : tn) {
        int* links[2] = { &nd.left, &nd.right };
        for (int* l : links) {
            if (*l >= 0) {
                *l += node_base;
            } else {
                const int ref = ~*l;
                *l = ~((((ref >> 3) + tri_first) << 3) | (ref & 7));
            }
        }
    }
    FF_HIP(hipMemcpy(s->d_nodes + node_base, tn.data(), tn.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    FF_HIP(hipMemcpy(s->d_tris + tri_first, tt.data(), tt.size() * sizeof(TriRecord), hipMemcpyHostToDevice));
    FF_HIP(hipMemcpy(s->d_normals + tri_first, nn.data(), nn.size() * sizeof(TriNormals), hipMemcpyHostToDevice));
    *out_nodes = (int)tn.size();
    *out_depth = depth;
    return FF_OK;
}

void refresh_scene_extent(FfState* s)
{
    int end = 0, depth = 0;
    for (size_t i = 0; i < s->slots.size(); ++i) {
        if (s->h_geoms[i].type != FF_GEOM_TRIANGLEMESH || s->slots[i].node_count == 0) continue;
        end = std::max(end, s->slots[i].node_first + s->slots[i].node_count);
        depth = std::max(depth, s->slots[i].depth);
    }
    s->num_nodes = end;
    s->max_depth = depth;
    s->build_stats.builder = s->scene_builder;
    s->build_stats.bvh_nodes = end;
    s->build_stats.bvh_max_depth = depth;
    s->build_stats.num_triangles = s->num_tris;
}

int upload_with_device_builder(FfState* s, const FfGeometry* host_geometries, int n, const BvhBuildParams& bp)
{
    FfBuildStats& bs = s->build_stats;
    CompiledScene cs;
    int st = compile_scene(host_geometries, n, bp, cs, /*build_bvh=*/false);
    if (st != FF_OK) return st;
    FF_HIP(hipSetDevice(s->device));
    free_scene(s);
    s->alloc_countdown = s->debug_fail_alloc;
    const int device_leaf = device_leaf_tris(bp);
    size_t node_cap = 0;
    for (const GeomRecord& g : cs.geoms)
        if (g.type == FF_GEOM_TRIANGLEMESH && g.tri_count > 0) node_cap += gpu_build_max_nodes(g.tri_count);
    FF_HIP(scene_alloc(s, (void**)&s->d_geoms, cs.geoms.size() * sizeof(GeomRecord)));
    FF_HIP(scene_alloc(s, (void**)&s->d_tris, (cs.total_tris ? (size_t)cs.total_tris : 1) * sizeof(TriRecord)));
    FF_HIP(scene_alloc(s, (void**)&s->d_normals, (cs.total_tris ? (size_t)cs.total_tris : 1) * sizeof(TriNormals)));
    FF_HIP(scene_alloc(s, (void**)&s->d_nodes, (node_cap ? node_cap : 1) * sizeof(BvhNode)));
    // (the kernels address a node by a 32-bit byte offset from the array's base)
    if ((uint64_t)(node_cap + cs.geoms.size() + 1) * sizeof(Bvh4Node) >= (1ull << 32))
        return fail(FF_ERR_UNSUPPORTED, "scene needs %llu 4-wide BVH nodes: more than the 38 million (4 GiB) the kernels address", (unsigned long long)(node_cap + cs.geoms.size() + 1));
    FF_HIP(scene_alloc(s, (void**)&s->d_nodes4, (node_cap + cs.geoms.size() + 1) * sizeof(Bvh4Node))); // (+ the geometry tree of a big scene)
    FF_HIP(scene_alloc(s, (void**)&s->d_parent, (node_cap ? node_cap : 1) * sizeof(int)));
    FF_HIP(scene_alloc(s, (void**)&s->d_role, node_cap ? node_cap : 1));
    s->node_capacity = node_cap;
    s->slots.assign(cs.geoms.size(), FfState::MeshSlot());
    int node_base = 0;
    for (size_t gi = 0; gi < cs.geoms.size(); ++gi) {
        GeomRecord& r = cs.geoms[gi];
        if (r.type != FF_GEOM_TRIANGLEMESH || r.tri_count <= 0) continue;
        const FfTriangle* src = host_geometries[r.orig_index].m_triangles;
        FfState::MeshSlot& slot = s->slots[gi];
        slot.node_first = node_base;
        slot.node_capacity = (int)gpu_build_max_nodes(r.tri_count);
        if (r.tri_count <= device_leaf) {
            st = place_single_leaf_mesh(s, src, r.tri_count, bp, r.tri_first, node_base, &slot.node_count, &slot.depth);
            if (st != FF_OK) return st;
        } else {
            st = stage_triangles(s, src, r.tri_count, &bs.copy_ms);
            if (st != FF_OK) return st;
            const auto t0 = std::chrono::steady_clock::now();
            MeshBuildInfo info;
            st = gpu_build_mesh(s->stream, s->scratch, s->d_stage, r.tri_count, r.tri_first, node_base, device_leaf, s->d_tris, s->d_normals, s->d_nodes, &info,
                                s->builder == FF_BUILD_GPU_PLOC);
            if (st != FF_OK) return st;
            FF_HIP(hipStreamSynchronize(s->stream));
            bs.build_ms += ms_since(t0);
            slot.node_count = info.node_count;
            slot.depth = info.depth;
        }
        {
            const auto t0 = std::chrono::steady_clock::now();
            st = collapse_slot(s, gi, /*with_info=*/true);
            if (st != FF_OK) return st;
            bs.build_ms += ms_since(t0);
        }
        r.bvh_root = node_base;
        node_base += slot.node_capacity;
    }
    s->h_geoms = cs.geoms;
    s->num_geoms = (int)cs.geoms.size();
    s->num_planes = s->num_quads = 0;
    for (const GeomRecord& g : cs.geoms) {
        s->num_planes += g.type != FF_GEOM_TRIANGLEMESH ? 1 : 0; // analytic shapes lead the records: planes, then spheres
        s->num_quads += g.type == FF_GEOM_PLANE ? 1 : 0;
    }
    s->has_specular = false;
    for (const GeomRecord& g : cs.geoms) s->has_specular = s->has_specular || g.bxdf_type == FF_BXDF_MIRROR || g.bxdf_type == FF_BXDF_GLASS;
    s->num_tris = cs.total_tris;
    s->scene_builder = s->builder;
    refresh_scene_extent(s);
    st = finalize_layout(s);
    if (st != FF_OK) return st;
    s->has_scene = true;
    return FF_OK;
}

} // namespace

namespace {
int upload_host_built(FfState* s, const FfGeometry* host_geometries, int n, const BvhBuildParams& bp);
}

int ff_set_builder(FfState* s, int builder)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_set_builder: state is null");
    if (builder != FF_BUILD_HOST_SAH && builder != FF_BUILD_GPU_LBVH && builder != FF_BUILD_GPU_PLOC) return fail(FF_ERR_INVALID_ARG, "ff_set_builder: unknown builder %d", builder);
    s->builder = builder;
    return FF_OK;
}

int ff_upload_scene(FfState* s, const FfGeometry* host_geometries, int n)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_upload_scene: state is null");
    s->primary_valid = s->last_key_valid = false; // (the stored primary hits belong to the scene that goes)
    const auto t_call = std::chrono::steady_clock::now();
    s->build_stats = FfBuildStats();
    const BvhBuildParams bp = default_bvh_params();
    if (s->builder != FF_BUILD_HOST_SAH) {
        const int st = upload_with_device_builder(s, host_geometries, n, bp);
        s->build_stats.total_ms = ms_since(t_call);
        if (st != FF_OK) free_scene(s);
        return st;
    }
    const int st = upload_host_built(s, host_geometries, n, bp);
    if (st != FF_OK) free_scene(s); // never leave a half-allocated scene behind (has_scene is false again)
    s->build_stats.total_ms = ms_since(t_call);
    return st;
}

} // extern "C"

namespace {

int upload_compiled(FfState* s, const CompiledScene& cs);

int upload_host_built(FfState* s, const FfGeometry* host_geometries, int n, const BvhBuildParams& bp)
{
    CompiledScene cs;
    const auto t_build = std::chrono::steady_clock::now();
    int st = compile_scene(host_geometries, n, bp, cs);
    if (st != FF_OK) return st;
    s->build_stats.build_ms = ms_since(t_build);
    return upload_compiled(s, cs);
}

// A scene compiled on the host (records, triangle records in leaf order, binary trees) onto the state's device: the copies,
// then the 4-wide trees derived there.  The same compiled scene may go to several devices (ff_multi_upload_scene).
int upload_compiled(FfState* s, const CompiledScene& cs)
{
    int st = FF_OK;
    FF_HIP(hipSetDevice(s->device));
    free_scene(s);
    s->alloc_countdown = s->debug_fail_alloc;
    const auto t_copy = std::chrono::steady_clock::now();
    // One allocation + one copy per array (the reference issues a cudaMallocManaged + two cudaMemcpy per geometry, kernel.cu:277-298).
    FF_HIP(scene_alloc(s, (void**)&s->d_geoms, cs.geoms.size() * sizeof(GeomRecord)));
    // Keep the triangle / node arrays non-null so the kernels can form addresses even for plane-only scenes.
    const size_t tri_bytes = (cs.tris.size() ? cs.tris.size() : 1) * sizeof(TriRecord);
    const size_t node_bytes = (cs.nodes.size() ? cs.nodes.size() : 1) * sizeof(BvhNode);
    FF_HIP(scene_alloc(s, (void**)&s->d_tris, tri_bytes));
    FF_HIP(scene_alloc(s, (void**)&s->d_normals, tri_bytes));
    static_assert(sizeof(TriNormals) == sizeof(TriRecord), "parallel arrays of equal stride");
    if (!cs.normals.empty()) FF_HIP(hipMemcpy(s->d_normals, cs.normals.data(), cs.normals.size() * sizeof(TriNormals), hipMemcpyHostToDevice));
    FF_HIP(scene_alloc(s, (void**)&s->d_nodes, node_bytes));
    // (the kernels address a node by a 32-bit byte offset from the array's base)
    if ((uint64_t)(cs.nodes.size() + cs.geoms.size() + 1) * sizeof(Bvh4Node) >= (1ull << 32))
        return fail(FF_ERR_UNSUPPORTED, "scene needs %llu 4-wide BVH nodes: more than the 38 million (4 GiB) the kernels address", (unsigned long long)(cs.nodes.size() + cs.geoms.size() + 1));
    FF_HIP(scene_alloc(s, (void**)&s->d_nodes4, (cs.nodes.size() + cs.geoms.size() + 1) * sizeof(Bvh4Node))); // (+ the geometry tree of a big scene)
    FF_HIP(scene_alloc(s, (void**)&s->d_parent, (cs.nodes.size() ? cs.nodes.size() : 1) * sizeof(int)));
    FF_HIP(scene_alloc(s, (void**)&s->d_role, cs.nodes.size() ? cs.nodes.size() : 1));
    if (!cs.tris.empty()) FF_HIP(hipMemcpy(s->d_tris, cs.tris.data(), cs.tris.size() * sizeof(TriRecord), hipMemcpyHostToDevice));
    if (!cs.nodes.empty()) FF_HIP(hipMemcpy(s->d_nodes, cs.nodes.data(), cs.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    s->build_stats.copy_ms = ms_since(t_copy);
    s->num_geoms = (int)cs.geoms.size();
    s->num_planes = s->num_quads = 0;
    for (const GeomRecord& g : cs.geoms) {
        s->num_planes += g.type != FF_GEOM_TRIANGLEMESH ? 1 : 0; // analytic shapes lead the records: planes, then spheres
        s->num_quads += g.type == FF_GEOM_PLANE ? 1 : 0;
    }
    s->has_specular = false;
    for (const GeomRecord& g : cs.geoms) s->has_specular = s->has_specular || g.bxdf_type == FF_BXDF_MIRROR || g.bxdf_type == FF_BXDF_GLASS;
    s->num_tris = cs.tris.size();
    s->node_capacity = cs.nodes.size();
    // Each mesh's nodes are contiguous with the root first: slot = [root, next mesh's root).
    s->h_geoms = cs.geoms;
    s->slots.assign(cs.geoms.size(), FfState::MeshSlot());
    std::vector<int> roots;
    for (const GeomRecord& g : cs.geoms)
        if (g.type == FF_GEOM_TRIANGLEMESH && g.bvh_root >= 0) roots.push_back(g.bvh_root);
    std::sort(roots.begin(), roots.end());
    for (size_t gi = 0; gi < cs.geoms.size(); ++gi) {
        const GeomRecord& g = cs.geoms[gi];
        if (g.type != FF_GEOM_TRIANGLEMESH || g.bvh_root < 0) continue;
        const auto next = std::upper_bound(roots.begin(), roots.end(), g.bvh_root);
        FfState::MeshSlot& slot = s->slots[gi];
        slot.node_first = g.bvh_root;
        slot.node_count = (next == roots.end() ? (int)cs.nodes.size() : *next) - g.bvh_root;
        slot.node_capacity = slot.node_count;
        slot.depth = cs.max_depth; // per-mesh depths are not kept by the host compiler; the scene maximum is a valid bound
    }
    s->scene_builder = FF_BUILD_HOST_SAH;
    refresh_scene_extent(s);
    s->num_nodes = (int)cs.nodes.size();
    s->max_depth = cs.max_depth;
    s->build_stats.bvh_nodes = s->num_nodes;
    s->build_stats.bvh_max_depth = s->max_depth;
    // the 4-wide trees the kernels traverse, derived on the device from the binary ones just copied
    const auto t_collapse = std::chrono::steady_clock::now();
    for (size_t gi = 0; gi < s->slots.size(); ++gi) {
        st = collapse_slot(s, gi, /*with_info=*/true);
        if (st != FF_OK) return st;
    }
    s->build_stats.build_ms += ms_since(t_collapse);
    st = finalize_layout(s);
    if (st != FF_OK) return st;
    s->has_scene = true;
    return FF_OK;
}

} // namespace

namespace ff {

// ff_upload_scene with a scene the caller compiled (ff_multi_upload_scene: one host build for all devices).  build_ms: what the
// compilation took, for the state's FfBuildStats.
int upload_compiled_scene(FfState* s, const CompiledScene& cs, double build_ms)
{
    const auto t_call = std::chrono::steady_clock::now();
    s->primary_valid = s->last_key_valid = false;
    s->build_stats = FfBuildStats();
    s->build_stats.build_ms = build_ms;
    const int st = upload_compiled(s, cs);
    if (st != FF_OK) free_scene(s);
    s->build_stats.total_ms = ms_since(t_call) + build_ms;
    return st;
}

} // namespace ff

extern "C" {

int ff_update_transforms(FfState* s, const FfGeometry* host_geometries, int n)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_update_transforms: state is null");
    if (!s->has_scene) return fail(FF_ERR_NO_SCENE, "ff_update_transforms: no scene uploaded");
    s->primary_valid = s->last_key_valid = false;
    const auto t_call = std::chrono::steady_clock::now();
    CompiledScene cs;
    int st = compile_scene(host_geometries, n, default_bvh_params(), cs, /*build_bvh=*/false);
    if (st != FF_OK) return st;
    if (cs.geoms.size() != s->h_geoms.size()) return fail(FF_ERR_INVALID_ARG, "ff_update_transforms: %d geometries, the uploaded scene has %zu", n, s->h_geoms.size());
    // The compiler orders the planes by the size of their world boxes, which a transform changes: match the new records to the
    // uploaded ones by the caller's index and keep the UPLOADED order (the mesh slots are parallel to it).  That the leading
    // planes are the largest is only an optimisation (count_scan_planes); a stale order stays correct.
    {
        std::vector<int> where(cs.geoms.size(), -1);
        for (size_t i = 0; i < cs.geoms.size(); ++i) {
            const int o = cs.geoms[i].orig_index;
            if (o >= 0 && (size_t)o < where.size()) where[o] = (int)i;
        }
        std::vector<GeomRecord> ordered(cs.geoms.size());
        for (size_t i = 0; i < s->h_geoms.size(); ++i) {
            const GeomRecord& b = s->h_geoms[i];
            const int j = b.orig_index >= 0 && (size_t)b.orig_index < where.size() ? where[b.orig_index] : -1;
            if (j < 0) return fail(FF_ERR_INVALID_ARG, "ff_update_transforms: geometry %d of the uploaded scene is missing", b.orig_index);
            GeomRecord a = cs.geoms[j];
            if (a.type != b.type || a.tri_count != b.tri_count || a.tri_first != b.tri_first)
                return fail(FF_ERR_INVALID_ARG, "ff_update_transforms: geometry %d differs in kind or triangle count from the uploaded one", a.orig_index);
            a.bvh_root = b.bvh_root;
            ordered[i] = a;
        }
        cs.geoms.swap(ordered);
    }
    FF_HIP(hipSetDevice(s->device));
    s->h_geoms = cs.geoms;
    st = finalize_layout(s); // (fills the records' tree fields again and copies them to the device)
    if (st != FF_OK) return st;
    s->has_specular = false; // materials may have changed
    for (const GeomRecord& g : cs.geoms) s->has_specular = s->has_specular || g.bxdf_type == FF_BXDF_MIRROR || g.bxdf_type == FF_BXDF_GLASS;
    s->build_stats.last_operation = 3;
    s->build_stats.total_ms = ms_since(t_call);
    s->build_stats.copy_ms = s->build_stats.total_ms;
    s->build_stats.build_ms = 0.0;
    return FF_OK;
}

int ff_update_mesh(FfState* s, int geometry_index, const FfTriangle* triangles, int count, int mode)
{
    clear_error();
    if (!s || !triangles) return fail(FF_ERR_INVALID_ARG, "ff_update_mesh: null argument");
    if (!s->has_scene) return fail(FF_ERR_NO_SCENE, "ff_update_mesh: no scene uploaded");
    s->primary_valid = s->last_key_valid = false;
    if (mode != FF_UPDATE_REFIT && mode != FF_UPDATE_REBUILD) return fail(FF_ERR_INVALID_ARG, "ff_update_mesh: unknown mode %d", mode);
    int gi = -1;
    for (size_t i = 0; i < s->h_geoms.size(); ++i)
        if (s->h_geoms[i].orig_index == geometry_index) gi = (int)i;
    if (gi < 0 || s->h_geoms[gi].type != FF_GEOM_TRIANGLEMESH) return fail(FF_ERR_INVALID_ARG, "ff_update_mesh: geometry %d is not an uploaded mesh", geometry_index);
    GeomRecord& rec = s->h_geoms[gi];
    if (count != rec.tri_count || count <= 0) return fail(FF_ERR_INVALID_ARG, "ff_update_mesh: %d triangles, the uploaded mesh has %d", count, rec.tri_count);
    if (mode == FF_UPDATE_REBUILD && s->scene_builder == FF_BUILD_HOST_SAH)
        return fail(FF_ERR_UNSUPPORTED, "ff_update_mesh: rebuilding in place needs a scene uploaded with a device builder (host-built trees are packed)");
    const auto t_call = std::chrono::steady_clock::now();
    FfBuildStats& bs = s->build_stats;
    bs.copy_ms = bs.build_ms = 0.0;
    FF_HIP(hipSetDevice(s->device));
    FfState::MeshSlot& slot = s->slots[gi];
    const BvhBuildParams bp = default_bvh_params();
    int st = stage_triangles(s, triangles, count, &bs.copy_ms);
    if (st != FF_OK) return st;
    const auto t_build = std::chrono::steady_clock::now();
    if (mode == FF_UPDATE_REBUILD) {
        if (count <= device_leaf_tris(bp)) {
            st = place_single_leaf_mesh(s, triangles, count, bp, rec.tri_first, slot.node_first, &slot.node_count, &slot.depth);
        } else {
            MeshBuildInfo info = MeshBuildInfo();
            st = gpu_build_mesh(s->stream, s->scratch, s->d_stage, count, rec.tri_first, slot.node_first, device_leaf_tris(bp), s->d_tris, s->d_normals, s->d_nodes, &info,
                                s->scene_builder == FF_BUILD_GPU_PLOC);
            if (st == FF_OK) {
                slot.node_count = info.node_count;
                slot.depth = info.depth;
            }
        }
        if (st != FF_OK) {
            // the mesh's records and nodes may be half rewritten: nothing may render from them
            s->has_scene = false;
            return st;
        }
        slot.parents_linked = false;
        st = collapse_slot(s, (size_t)gi, /*with_info=*/true);
        if (st != FF_OK) {
            s->has_scene = false;
            return st;
        }
        bs.last_operation = 2;
    } else {
        if (!slot.parents_linked) {
            st = gpu_link_parents(s->stream, s->d_nodes, slot.node_first, slot.node_count, s->d_parent + slot.node_first);
            if (st != FF_OK) return st;
            slot.parents_linked = true;
        }
        st = gpu_refit_mesh(s->stream, s->scratch, s->d_stage, count, rec.tri_first, slot.node_first, slot.node_count, s->d_parent + slot.node_first, s->d_tris,
                            s->d_normals, s->d_nodes);
        if (st != FF_OK) return st;
        st = collapse_slot(s, (size_t)gi, /*with_info=*/false); // same topology: the boxes of the 4-wide nodes follow
        if (st != FF_OK) return st;
        bs.last_operation = 1;
    }
    // the geometry's world box follows the new vertices
    float omn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, omx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int t = 0; t < count; ++t) {
        const FfVec3* v[3] = { &triangles[t].m_v0, &triangles[t].m_v1, &triangles[t].m_v2 };
        for (const FfVec3* p : v) {
            omn[0] = std::min(omn[0], p->x); omn[1] = std::min(omn[1], p->y); omn[2] = std::min(omn[2], p->z);
            omx[0] = std::max(omx[0], p->x); omx[1] = std::max(omx[1], p->y); omx[2] = std::max(omx[2], p->z);
        }
    }
    set_world_box(rec, omn, omx);
    refresh_scene_extent(s);
    if (s->scene_builder == FF_BUILD_HOST_SAH) s->num_nodes = (int)s->node_capacity;
    st = finalize_layout(s); // (a rebuilt tree may differ in size and depth: LDS shares and workgroup size follow; copies the records)
    if (st != FF_OK) return st;
    bs.build_ms = ms_since(t_build);
    bs.total_ms = ms_since(t_call);
    return FF_OK;
}

int ff_build_stats(FfState* s, FfBuildStats* out)
{
    clear_error();
    if (!s || !out) return fail(FF_ERR_INVALID_ARG, "ff_build_stats: null argument");
    *out = s->build_stats;
    return FF_OK;
}

int ff_debug_download_bvh(FfState* s, void* nodes, int max_nodes, int* out_nodes, void* tris, int max_tris, int* out_tris, int* mesh_table,
                          int max_geometries)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_debug_download_bvh: state is null");
    if (!s->has_scene) return fail(FF_ERR_NO_SCENE, "ff_debug_download_bvh: no scene uploaded");
    if (out_nodes) *out_nodes = s->num_nodes;
    if (out_tris) *out_tris = (int)s->num_tris;
    FF_HIP(hipSetDevice(s->device));
    if (nodes && max_nodes > 0) FF_HIP(hipMemcpy(nodes, s->d_nodes, (size_t)std::min(max_nodes, s->num_nodes) * sizeof(BvhNode), hipMemcpyDeviceToHost));
    if (tris && max_tris > 0) FF_HIP(hipMemcpy(tris, s->d_tris, (size_t)std::min<uint64_t>((uint64_t)max_tris, s->num_tris) * sizeof(TriRecord), hipMemcpyDeviceToHost));
    if (mesh_table) {
        for (size_t i = 0; i < s->h_geoms.size(); ++i) {
            const GeomRecord& g = s->h_geoms[i];
            if (g.orig_index < 0 || g.orig_index >= max_geometries) continue;
            int* row = mesh_table + 5 * g.orig_index;
            const bool mesh = g.type == FF_GEOM_TRIANGLEMESH && g.bvh_root >= 0;
            row[0] = mesh ? g.bvh_root : -1;
            row[1] = mesh ? s->slots[i].node_count : 0;
            row[2] = g.tri_first;
            row[3] = g.tri_count;
            row[4] = mesh ? s->slots[i].depth : 0;
        }
    }
    return FF_OK;
}

int ff_debug_download_bvh4(FfState* s, void* nodes4, int max_nodes, int* out_capacity, int* mesh_table, int max_geometries)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_debug_download_bvh4: state is null");
    if (!s->has_scene) return fail(FF_ERR_NO_SCENE, "ff_debug_download_bvh4: no scene uploaded");
    if (out_capacity) *out_capacity = (int)s->node_capacity;
    FF_HIP(hipSetDevice(s->device));
    if (nodes4 && max_nodes > 0)
        FF_HIP(hipMemcpy(nodes4, s->d_nodes4, (size_t)std::min<size_t>((size_t)max_nodes, s->node_capacity) * sizeof(Bvh4Node), hipMemcpyDeviceToHost));
    if (mesh_table) {
        for (size_t i = 0; i < s->h_geoms.size(); ++i) {
            const GeomRecord& g = s->h_geoms[i];
            if (g.orig_index < 0 || g.orig_index >= max_geometries) continue;
            int* row = mesh_table + 6 * g.orig_index;
            const bool mesh = g.type == FF_GEOM_TRIANGLEMESH && g.bvh_root >= 0;
            int lds_first = 0;
            std::memcpy(&lds_first, &g.wmin[3], sizeof(int));
            row[0] = mesh ? g.node4_first : -1;
            row[1] = mesh ? s->slots[i].node4_count : 0;
            row[2] = mesh ? s->slots[i].depth4 : 0;
            row[3] = mesh ? lds_first : 0;
            row[4] = mesh ? g.lds_nodes : 0;
            row[5] = s->lds_cap;
        }
    }
    return FF_OK;
}

int ff_strips_local_rows(int height, int strip_rows, int part, int num_parts)
{
    if (height <= 0 || strip_rows <= 0 || num_parts <= 0 || part < 0 || part >= num_parts) return 0;
    const int nstrips = (height + strip_rows - 1) / strip_rows;
    int rows = 0;
    for (int sidx = part; sidx < nstrips; sidx += num_parts) {
        const int y0 = sidx * strip_rows;
        rows += (y0 + strip_rows <= height) ? strip_rows : height - y0;
    }
    return rows;
}

int ff_render_strips(FfState* s, const FfCamera* camera, const FfRenderParams* params, int strip_rows, int part, int num_parts, void* rgb8,
                     int rgb8_on_device, float* radiance, int radiance_on_device, int* out_local_rows)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    int st = check_render_call(s, camera, params, "ff_render");
    if (st != FF_OK) return st;
    if (strip_rows <= 0 || num_parts <= 0 || part < 0 || part >= num_parts) return fail(FF_ERR_INVALID_ARG, "ff_render_strips: bad strip partition (%d rows, part %d of %d)", strip_rows, part, num_parts);
    FF_HIP(hipSetDevice(s->device));
    const int local_rows = ff_strips_local_rows(params->height, strip_rows, part, num_parts);
    if (out_local_rows) *out_local_rows = local_rows;
    const size_t local_pixels = (size_t)local_rows * (size_t)params->width;

    unsigned char* rgb8_dev = nullptr;
    float* rad_dev = nullptr;
    if (rgb8) {
        if (rgb8_on_device) rgb8_dev = (unsigned char*)rgb8;
        else {
            st = ensure_bytes((void**)&s->d_rgb8, &s->rgb8_bytes, local_pixels * 3 + 16);
            if (st != FF_OK) return st;
            rgb8_dev = s->d_rgb8;
        }
    }
    if (radiance) {
        if (radiance_on_device) rad_dev = radiance;
        else {
            st = ensure_bytes((void**)&s->d_radiance, &s->radiance_bytes, local_pixels * 3 * sizeof(float) + 16);
            if (st != FF_OK) return st;
            rad_dev = s->d_radiance;
        }
    }
    st = render_local(s, camera, params, strip_rows, part, num_parts, local_rows, rgb8_dev, rad_dev);
    if (st != FF_OK) return st;
    if (rgb8 && !rgb8_on_device && local_pixels) FF_HIP(hipMemcpy(rgb8, rgb8_dev, local_pixels * 3, hipMemcpyDeviceToHost));
    if (radiance && !radiance_on_device && local_pixels) FF_HIP(hipMemcpy(radiance, rad_dev, local_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FF_OK;
}

int ff_render_tile(FfState* s, const FfCamera* camera, const FfRenderParams* params, int x0, int y0, int w, int h, void* rgb8, int rgb8_on_device,
                   float* radiance, int radiance_on_device)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    int st = check_render_call(s, camera, params, "ff_render_tile");
    if (st != FF_OK) return st;
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > params->width || y0 + h > params->height)
        return fail(FF_ERR_INVALID_ARG, "ff_render_tile: tile %dx%d at (%d, %d) is not inside the %dx%d image", w, h, x0, y0, params->width, params->height);
    FF_HIP(hipSetDevice(s->device));
    const size_t pixels = (size_t)w * (size_t)h;
    unsigned char* rgb8_dev = nullptr;
    float* rad_dev = nullptr;
    if (rgb8) {
        if (rgb8_on_device) rgb8_dev = (unsigned char*)rgb8;
        else {
            st = ensure_bytes((void**)&s->d_rgb8, &s->rgb8_bytes, pixels * 3 + 16);
            if (st != FF_OK) return st;
            rgb8_dev = s->d_rgb8;
        }
    }
    if (radiance) {
        if (radiance_on_device) rad_dev = radiance;
        else {
            st = ensure_bytes((void**)&s->d_radiance, &s->radiance_bytes, pixels * 3 * sizeof(float) + 16);
            if (st != FF_OK) return st;
            rad_dev = s->d_radiance;
        }
    }
    st = render_local(s, camera, params, h, 0, 1, h, rgb8_dev, rad_dev, x0, y0, w);
    if (st != FF_OK) return st;
    if (rgb8 && !rgb8_on_device) FF_HIP(hipMemcpy(rgb8, rgb8_dev, pixels * 3, hipMemcpyDeviceToHost));
    if (radiance && !radiance_on_device) FF_HIP(hipMemcpy(radiance, rad_dev, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FF_OK;
}

int ff_render(FfState* s, const FfCamera* camera, const FfRenderParams* params, void* rgb8, int rgb8_on_device, float* radiance,
              int radiance_on_device)
{
    const int h = params ? params->height : 1;
    return ff_render_strips(s, camera, params, h > 0 ? h : 1, 0, 1, rgb8, rgb8_on_device, radiance, radiance_on_device, nullptr);
}

int ff_deinterleave_strips(FfState* s, const void* src_dev, void* dst_dev, int width, int height, int strip_rows, int num_parts, int elem_bytes)
{
    clear_error();
    if (!s || !src_dev || !dst_dev) return fail(FF_ERR_INVALID_ARG, "ff_deinterleave_strips: null argument");
    if (width <= 0 || height <= 0 || strip_rows <= 0 || num_parts <= 0 || elem_bytes <= 0) return fail(FF_ERR_INVALID_ARG, "ff_deinterleave_strips: bad geometry");
    FF_HIP(hipSetDevice(s->device));
    FF_HIP(launch_deinterleave(src_dev, dst_dev, width, height, strip_rows, num_parts, elem_bytes, s->stream));
    FF_HIP(hipStreamSynchronize(s->stream));
    return FF_OK;
}

int ff_intersect_rays(FfState* s, const FfRay* rays, int n, FfIntersect* out, int trace_mode)
{
    clear_error();
    if (!s || !rays || !out) return fail(FF_ERR_INVALID_ARG, "ff_intersect_rays: null argument");
    if (n < 0) return fail(FF_ERR_INVALID_ARG, "ff_intersect_rays: negative count");
    if (trace_mode != FF_TRACE_BRUTE_FORCE && trace_mode != FF_TRACE_BVH) return fail(FF_ERR_INVALID_ARG, "unknown trace_mode %d", trace_mode);
    if (!s->has_scene) return fail(FF_ERR_NO_SCENE, "ff_intersect_rays: no scene uploaded");
    if (n == 0) return FF_OK;
    if (trace_mode == FF_TRACE_BVH && s->scene_block_threads == 0)
        return fail(FF_ERR_UNSUPPORTED, "ff_intersect_rays: 4-wide BVH of depth %d does not fit the LDS traversal stack; use FF_TRACE_BRUTE_FORCE or a host-built tree", s->max_depth4);
    FF_HIP(hipSetDevice(s->device));
    FfRay* d_rays = nullptr;
    FfIntersect* d_out = nullptr;
    FF_HIP(hipMalloc((void**)&d_rays, (size_t)n * sizeof(FfRay)));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n * sizeof(FfIntersect));
    if (e != hipSuccess) {
        (void)hipFree(d_rays);
        return fail(FF_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    RayBatchParams p;
    p.rays = d_rays;
    p.out = d_out;
    p.n = n;
    p.num_geoms = s->num_geoms;
    p.num_planes = s->num_planes;
    p.num_quads = s->num_quads;
    p.geoms = s->d_geoms;
    p.tris = s->d_tris;
    p.nodes4 = s->d_nodes4;
    p.stack_depth = s->stack_lds_levels;
    p.stack_spill = nullptr;
    if (trace_mode == FF_TRACE_BVH && s->stack_lds_levels < s->stack_entries) {
        const size_t threads = (size_t)((n + kBlockThreads - 1) / kBlockThreads) * (size_t)kBlockThreads;
        const int sst = ensure_bytes((void**)&s->d_stack_spill, &s->stack_spill_bytes, (size_t)(s->stack_entries - s->stack_lds_levels) * threads * sizeof(int));
        if (sst != FF_OK) {
            (void)hipFree(d_rays);
            (void)hipFree(d_out);
            return sst;
        }
        p.stack_spill = s->d_stack_spill;
    }
    p.top_first = (int)s->node_capacity;
    p.top_lds_first = 0;
    p.top_lds_count = s->top_lds_count;
    p.num_scan = s->num_scan;
    p.walls = s->walls;
    p.guard_hits = s->d_counters;
    e = hipMemsetAsync(s->d_counters, 0, sizeof(unsigned long long), s->stream);
    p.lds_nodes = s->lds_cap; // (the records' LDS shares were laid out for the trace kernel's workgroup; 512 threads leave more room, never less)
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * sizeof(FfRay), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_ray_batch(p, trace_mode, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    unsigned long long cut = 0;
    if (e == hipSuccess) e = hipMemcpy(&cut, s->d_counters, sizeof cut, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * sizeof(FfIntersect), hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(FF_ERR_HIP, "ff_intersect_rays failed: %s", hipGetErrorString(e));
    if (cut != 0) return fail(FF_ERR_HIP, "ff_intersect_rays: the traversal loop guard cut %llu queries short (a malformed or absurdly deep tree)", cut);
    return FF_OK;
}

// ---- OpenGL pixel-buffer interop ---------------------------------------------------------------------------------

int ff_register_gl_pbo(FfState* s, unsigned int pbo, int width, int height)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_register_gl_pbo: state is null");
    if (width <= 0 || height <= 0) return fail(FF_ERR_INVALID_ARG, "ff_register_gl_pbo: bad size");
    (void)hipSetDevice(s->device);
    if (s->pbo_resource) {
        (void)hipGraphicsUnregisterResource(s->pbo_resource);
        s->pbo_resource = nullptr;
    }
    // utilities.h:618: cudaGraphicsGLRegisterBuffer(&pboCudaResource, pbo, cudaGraphicsRegisterFlagsWriteDiscard)
    hipGraphicsResource* res = nullptr;
    hipError_t e = hipGraphicsGLRegisterBuffer(&res, (GLuint)pbo, hipGraphicsRegisterFlagsWriteDiscard);
    if (e != hipSuccess || !res) {
        (void)hipGetLastError();
        return fail(FF_ERR_GL_UNAVAILABLE, "hipGraphicsGLRegisterBuffer(%u) failed: %s (is a GL context current on this thread?)", pbo, hipGetErrorString(e));
    }
    s->pbo_resource = res;
    s->pbo_width = width;
    s->pbo_height = height;
    return FF_OK;
}

int ff_unregister_gl_pbo(FfState* s)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_unregister_gl_pbo: state is null");
    if (!s->pbo_resource) return FF_OK;
    (void)hipSetDevice(s->device);
    hipError_t e = hipGraphicsUnregisterResource(s->pbo_resource); // utilities.h:516
    s->pbo_resource = nullptr;
    if (e != hipSuccess) return fail(FF_ERR_HIP, "hipGraphicsUnregisterResource failed: %s", hipGetErrorString(e));
    return FF_OK;
}

int ff_render_to_pbo(FfState* s, const FfCamera* camera, const FfRenderParams* params)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    if (s && !s->pbo_resource) return fail(FF_ERR_GL_UNAVAILABLE, "ff_render_to_pbo: no pixel buffer registered");
    int st = check_render_call(s, camera, params, "ff_render_to_pbo");
    if (st != FF_OK) return st;
    if (params->width != s->pbo_width || params->height != s->pbo_height)
        return fail(FF_ERR_INVALID_ARG, "ff_render_to_pbo: params are %dx%d but the registered buffer is %dx%d", params->width, params->height, s->pbo_width, s->pbo_height);
    FF_HIP(hipSetDevice(s->device));
    // kernel.cu:335-344
    void* dptr = nullptr;
    size_t nbytes = 0;
    FF_HIP(hipGraphicsMapResources(1, &s->pbo_resource, s->stream));                 // :338
    hipError_t e = hipGraphicsResourceGetMappedPointer(&dptr, &nbytes, s->pbo_resource); // :339
    if (e == hipSuccess && nbytes < (size_t)params->width * (size_t)params->height * 3) e = hipErrorInvalidValue;
    if (e == hipSuccess) {
        st = render_local(s, camera, params, params->height, 0, 1, params->height, (unsigned char*)dptr, nullptr); // :340-342
    } else {
        st = fail(FF_ERR_HIP, "mapping the pixel buffer failed: %s", hipGetErrorString(e));
    }
    hipError_t ue = hipGraphicsUnmapResources(1, &s->pbo_resource, s->stream);       // :344
    if (st == FF_OK && ue != hipSuccess) st = fail(FF_ERR_HIP, "hipGraphicsUnmapResources failed: %s", hipGetErrorString(ue));
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

// ---- progressive accumulation ("progressive accumulation across frames while the camera is still") ----

namespace {

// Frame `frame_index` of a progressive sequence into device buffers: an ordinary frame with seed + frame_index, added to
// the running sum; outputs are the mean over frames 0..frame_index.
int progressive_frame(FfState* s, const FfCamera* camera, const FfRenderParams* params, int frame_index, unsigned char* rgb8_dev, float* mean_dev)
{
    if (frame_index < 0) return fail(FF_ERR_INVALID_ARG, "ff_render_progressive: frame index %d", frame_index);
    if (frame_index > 0 && (params->width != s->accum_width || params->height != s->accum_height || frame_index != s->accum_frames))
        return fail(FF_ERR_INVALID_ARG, "ff_render_progressive: frame %d does not continue the running sequence (%d frames of %dx%d); restart with frame 0",
                    frame_index, s->accum_frames, s->accum_width, s->accum_height);
    const size_t values = (size_t)params->width * (size_t)params->height * 3;
    int st = ensure_bytes((void**)&s->d_frame, &s->frame_bytes, values * sizeof(float) + 16);
    if (st == FF_OK) st = ensure_bytes((void**)&s->d_accum, &s->accum_bytes, values * sizeof(float) + 16);
    if (st != FF_OK) return st;
    FfRenderParams p = *params;
    p.seed = params->seed + (uint64_t)frame_index;
    st = render_local(s, camera, &p, params->height, 0, 1, params->height, nullptr, s->d_frame);
    if (st != FF_OK) return st;
    const float inv = 1.0f / (float)(frame_index + 1);
    FF_HIP(launch_accumulate(s->d_accum, s->d_frame, mean_dev, rgb8_dev, values, frame_index == 0 ? 1 : 0, inv, s->stream));
    FF_HIP(hipStreamSynchronize(s->stream));
    s->accum_width = params->width;
    s->accum_height = params->height;
    s->accum_frames = frame_index + 1;
    return FF_OK;
}

} // namespace

int ff_render_progressive(FfState* s, const FfCamera* camera, const FfRenderParams* params, int frame_index, void* rgb8, int rgb8_on_device, float* radiance,
                          int radiance_on_device)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    int st = check_render_call(s, camera, params, "ff_render_progressive");
    if (st != FF_OK) return st;
    FF_HIP(hipSetDevice(s->device));
    const size_t pixels = (size_t)params->width * (size_t)params->height;
    unsigned char* rgb8_dev = nullptr;
    float* mean_dev = nullptr;
    if (rgb8) {
        if (rgb8_on_device) rgb8_dev = (unsigned char*)rgb8;
        else {
            st = ensure_bytes((void**)&s->d_rgb8, &s->rgb8_bytes, pixels * 3 + 16);
            if (st != FF_OK) return st;
            rgb8_dev = s->d_rgb8;
        }
    }
    if (radiance) {
        if (radiance_on_device) mean_dev = radiance;
        else {
            st = ensure_bytes((void**)&s->d_mean, &s->mean_bytes, pixels * 3 * sizeof(float) + 16);
            if (st != FF_OK) return st;
            mean_dev = s->d_mean;
        }
    }
    st = progressive_frame(s, camera, params, frame_index, rgb8_dev, mean_dev);
    if (st != FF_OK) return st;
    if (rgb8 && !rgb8_on_device) FF_HIP(hipMemcpy(rgb8, rgb8_dev, pixels * 3, hipMemcpyDeviceToHost));
    if (radiance && !radiance_on_device) FF_HIP(hipMemcpy(radiance, mean_dev, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FF_OK;
}

int ff_render_to_pbo_progressive(FfState* s, const FfCamera* camera, const FfRenderParams* params, int frame_index)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    if (s && !s->pbo_resource) return fail(FF_ERR_GL_UNAVAILABLE, "ff_render_to_pbo_progressive: no pixel buffer registered");
    int st = check_render_call(s, camera, params, "ff_render_to_pbo_progressive");
    if (st != FF_OK) return st;
    if (params->width != s->pbo_width || params->height != s->pbo_height)
        return fail(FF_ERR_INVALID_ARG, "ff_render_to_pbo_progressive: params are %dx%d but the registered buffer is %dx%d", params->width, params->height,
                    s->pbo_width, s->pbo_height);
    FF_HIP(hipSetDevice(s->device));
    void* dptr = nullptr;
    size_t nbytes = 0;
    FF_HIP(hipGraphicsMapResources(1, &s->pbo_resource, s->stream));
    hipError_t e = hipGraphicsResourceGetMappedPointer(&dptr, &nbytes, s->pbo_resource);
    if (e == hipSuccess && nbytes < (size_t)params->width * (size_t)params->height * 3) e = hipErrorInvalidValue;
    if (e == hipSuccess) st = progressive_frame(s, camera, params, frame_index, (unsigned char*)dptr, nullptr);
    else st = fail(FF_ERR_HIP, "mapping the pixel buffer failed: %s", hipGetErrorString(e));
    hipError_t ue = hipGraphicsUnmapResources(1, &s->pbo_resource, s->stream);
    if (st == FF_OK && ue != hipSuccess) st = fail(FF_ERR_HIP, "hipGraphicsUnmapResources failed: %s", hipGetErrorString(ue));
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

// ---- measurement ---------------------------------------------------------------------------------------------------

int ff_set_collect_stats(FfState* s, int on)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_set_collect_stats: state is null");
    s->collect_stats = on != 0;
    return FF_OK;
}

int ff_debug_timeline(FfState* s, unsigned* out1024, int* bucket_us)
{
    clear_error();
    if (!s || !out1024 || !bucket_us) return fail(FF_ERR_INVALID_ARG, "ff_debug_timeline: null argument");
    std::memset(s->h_timeline, 0, sizeof s->h_timeline);
    for (size_t i = 0; i < s->h_timeline_rows.size(); ++i) s->h_timeline[i % kTimelineBuckets] += s->h_timeline_rows[i];
    std::memcpy(out1024, s->h_timeline, sizeof s->h_timeline);
    *bucket_us = s->timeline_bucket_us;
    return FF_OK;
}

int ff_debug_counters(FfState* s, unsigned long long* out32)
{
    clear_error();
    if (!s || !out32) return fail(FF_ERR_INVALID_ARG, "ff_debug_counters: null argument");
    std::memcpy(out32, s->raw_counters, sizeof s->raw_counters);
    return FF_OK;
}

int ff_debug_check_ieee(FfState* s, unsigned long long* out_mismatches2)
{
    clear_error();
    if (!s || !out_mismatches2) return fail(FF_ERR_INVALID_ARG, "ff_debug_check_ieee: null argument");
    FF_HIP(hipSetDevice(s->device));
    FF_HIP(hipMemsetAsync(s->d_counters, 0, 2 * sizeof(unsigned long long), s->stream));
    FF_HIP(launch_ieee_check(s->d_counters, s->stream));
    FF_HIP(hipMemcpyAsync(out_mismatches2, s->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    FF_HIP(hipStreamSynchronize(s->stream));
    return FF_OK;
}

const char* ff_debug_kernel_name(FfState* s) { return s && s->last_kernel_name ? s->last_kernel_name : ""; }

int ff_stats(FfState* s, FfStats* out)
{
    clear_error();
    if (!s || !out) return fail(FF_ERR_INVALID_ARG, "ff_stats: null argument");
    *out = s->stats;
    return FF_OK;
}

} // extern "C"
