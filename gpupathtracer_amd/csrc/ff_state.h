// ff_state.h — the tracer state behind the opaque FfState handle and the internal render steps shared by the
// translation units of the library (ff_api.cpp: single-device entry points; ff_dist.cpp: multi-GPU entry points).
#pragma once

#include <vector>

#include <hip/hip_runtime.h>

#include "ff_build.h"
#include "ff_internal.h"
#include "ff_kernels.h"

struct FfDistContext; // ff_dist.cpp

struct FfState {
    int device = 0;
    int num_cus = 0;
    size_t device_mem_bytes = 0; // total global memory of the device (caps the buffers the library allocates on its own initiative)
    hipStream_t stream = nullptr;
    // scene (device)
    ff::GeomRecord* d_geoms = nullptr;
    ff::TriRecord* d_tris = nullptr;
    ff::TriNormals* d_normals = nullptr; // vertex normals, parallel to d_tris
    ff::BvhNode* d_nodes = nullptr;   // binary trees: what the builders write and refit works on
    ff::WallTable walls = {};          // the axis-aligned planes among the records every query screens (finalize_layout)
    int num_scan = 0;                  // big scenes: leading plane records kept out of the geometry tree (count_scan_planes)
    ff::Bvh4Node* d_nodes4 = nullptr; // 4-wide trees derived from them (gpu_collapse_mesh): what the trace kernels traverse;
                                      // mesh i's nodes start at its binary slot's index (slots[i].node_first)
    int num_geoms = 0, num_planes = 0, num_quads = 0, num_nodes = 0, max_depth = 0;
    int num_nodes4 = 0, max_depth4 = 0;   // 4-wide nodes in use (sum over meshes), deepest 4-wide tree
    // scenes of more than kChunkGeometries geometries: the 4-wide tree over the geometries' world boxes, kept in d_nodes4
    // from index node_capacity on (0 nodes: a small scene)
    int top_count = 0, top_depth = 0, top_lds_count = 0;
    int stack_entries = 1;                // traversal stack entries per lane the BVH kernels need for this scene
    int stack_lds_levels = 1;             // ... of which this many live in LDS (finalize_layout); the rest in d_stack_spill
    unsigned long long* d_cull_mask = nullptr; // primary-ray cull: one bit per pixel item (KParams::cull_mask)
    size_t cull_mask_bytes = 0;
    float4* d_primary_cache = nullptr; // KParams::primary_cache
    size_t primary_cache_bytes = 0;
    // The stored primary hits (and the mask of pixels that see nothing) belong to a camera, a pixel mapping and a scene: while those
    // stay what they were - a viewer that accumulates 1-spp frames with the camera at rest (kernel.cu:266,342) - the next frame starts
    // from them without a pre-pass.  primary_key: what they were computed for; any change of the scene clears primary_valid.
    struct PrimaryKey {
        float cam[24];
        int dims[12];
        unsigned pix_items;
    } primary_key = {};
    bool primary_valid = false;
    PrimaryKey last_key = {};                // the key of the frame before this one, stored hits or not ("has the camera come to rest?")
    bool last_key_valid = false;
    bool primary_has_mask = false;           // ... and d_cull_mask holds the mask that goes with them
    unsigned long long primary_culled_pixels = 0; // pixels marked in it (FfStats::rays_answered of frames that reuse it)
    bool pending_mask_reused = false, pending_mask_built = false;
    float4* d_park = nullptr;          // KParams::park (job-pool kernel)
    size_t park_bytes = 0;
    int* d_stack_spill = nullptr;         // (stack_entries - stack_lds_levels) x launch threads ints
    size_t stack_spill_bytes = 0;
    bool use_pool = false;                    // the scene renders with the job-pool kernel (finalize_layout): its LDS layout leaves room for the pool
    int scene_block_threads = 0, lds_cap = 0; // BVH kernel workgroup size and LDS node slots chosen for this scene (finalize_layout)
    bool has_specular = false;
    uint64_t num_tris = 0;
    bool has_scene = false;
    // scene bookkeeping for updates (ff_update_transforms / ff_update_mesh)
    struct MeshSlot {
        int node_first = 0, node_count = 0, node_capacity = 0, depth = 0;
        int node4_count = 0, depth4 = 0; // its 4-wide tree: nodes [node_first, node_first + node4_count) of d_nodes4
        bool parents_linked = false;
    };
    int builder = FF_BUILD_HOST_SAH;       // builder for the next upload (ff_set_builder)
    int scene_builder = FF_BUILD_HOST_SAH; // builder that produced the scene on the device
    std::vector<ff::GeomRecord> h_geoms;       // the uploaded records, processing order
    std::vector<MeshSlot> slots;           // parallel to h_geoms (meshes only)
    size_t node_capacity = 0;              // nodes allocated in d_nodes
    int* d_parent = nullptr;               // node_capacity ints (refit)
    unsigned char* d_role = nullptr;       // node_capacity bytes: which binary nodes are 4-wide nodes (gpu_collapse_mesh keeps it across refits)
    FfTriangle* d_stage = nullptr;         // staging copy of a caller triangle array (device builder / refit)
    size_t stage_bytes = 0;
    ff::BuildScratch scratch;
    FfBuildStats build_stats = {};
    // work buffers (device)
    float* d_blocksums = nullptr;
    size_t blocksums_bytes = 0;
    unsigned char* d_rgb8 = nullptr;
    size_t rgb8_bytes = 0;
    float* d_radiance = nullptr;
    size_t radiance_bytes = 0;
    // fine-grained tail (KParams::tail_samples)
    float4* d_tail_samples = nullptr;
    size_t tail_samples_bytes = 0;
    bool tail_forced = false; // FF_TAIL_GROUP given: the fine-grained tail also for one-part frames
    int tail_min_blocks = 4;  // launches with fewer sample blocks keep whole-block items (FF_TAIL_MIN_BLOCKS)
    int tail_group_spp = 8;  // FF_TAIL_GROUP (0 = off): samples per tail item of a multi-part frame.  8 since the samples start from stored hits (r04: eight ranks +0.5, four +0.7 points; rounds 2-3: 16)
                             // (slowest of 8 ranks at 95.6 % of full-frame time / 8 instead of 93.8 %, of 4 ranks 97.4 / 96.7: tools/strip_scaling.py)
    // progressive accumulation (ff_render_progressive)
    float* d_accum = nullptr;
    size_t accum_bytes = 0;
    float* d_frame = nullptr;
    size_t frame_bytes = 0;
    float* d_mean = nullptr;
    size_t mean_bytes = 0;
    int accum_width = 0, accum_height = 0, accum_frames = 0;
    unsigned* d_queue = nullptr;               // work-queue counter: lives right behind the counters (one memset clears both)
    unsigned long long* d_counters = nullptr;  // 28 counters + 4 queue words
    unsigned long long* h_counters = nullptr;  // pinned mirror for the per-frame read-back
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool collect_stats = false;
    // Experiment switches (DESIGN.md "knobs"): read from the environment ONCE, at ff_create (ff_debug_reload_switches re-reads them on
    // request: the A/B tests); the render and layout paths only look here, so a state behaves the same from frame to frame and no
    // getenv runs next to another thread's setenv.
    struct Switches {
        bool no_last_bounce_cut = false, no_primary_cull = false, no_primary_reuse = false; // per frame (render_enqueue)
        bool no_primary_cache = false; // FF_NO_PRIMARY_CACHE: every frame runs its own pre-pass (the stored hits are not kept from frame to frame)
        int tail_blocks = 0;           // FF_TAIL_BLOCKS: how many of the frame's last sample blocks go out as short items (0: the library's choice)
        int reuse_min_spp = 2;         // FF_REUSE_MIN_SPP: frames of fewer samples per pixel trace their primary rays themselves unless the hits are there
        int reuse_quorum = 1;
        int queue_chunk = 0, queue_counters = 0; // 0: the library's choice
        int queue_tail = -1;           // FF_QUEUE_TAIL: items per wave in the as-asked tail zone of the work queue (-1: the library's choice, 0: off)
        bool no_wall_table = false, no_wall_pairs = false, no_stack_spill = false, no_scan_planes = false; // layout (finalize_layout / scene compile)
        bool lds_fill = false;         // FF_DEBUG_LDS_FILL=words,pattern
        unsigned long lds_fill_words = 0, lds_fill_pattern = 0;
        int pool = -1;                 // FF_POOL: -1 the library's choice, 0 the lane-owned traversal kernel, 1 the job-pool kernel
        int pool_quorum = 0;           // FF_POOL_QUORUM: ready lanes a wave waits for before a setup pass (0: the library's choice)
        int pool_quorum_min = 0;       // FF_POOL_QUORUM_MIN: ... when the job queue is empty
        int pool_refill = 0;           // FF_POOL_REFILL: free lanes of a traversing wave that trigger a fetch of new jobs
        int pool_slice = 0;            // FF_POOL_SLICE: inner-node rounds per traversal slice
        int pool_batch_min = 0;        // FF_POOL_BATCH_MIN: jobs the queue must hold before a wave goes and takes some (fewer only after a few empty looks)
        int lds_node_cap = -1;         // FF_DEBUG_LDS_NODE_CAP: at most this many tree nodes in LDS (experiments on partial residency)
        int pool_leave = -1;           // FF_POOL_LEAVE: jobs a wave may still hold when it leaves the traverse role for a setup pass (they are put down)
        int pool_stack_levels = 0;     // FF_POOL_STACK_LEVELS: traversal stack levels kept in LDS (the deeper ones spill)
    } sw;
    int block_threads = ff::kBlockThreadsMax; // BVH kernel workgroup size (512 or 1024); FF_BLOCK_THREADS overrides for experiments
    bool setup_threshold_forced = false;
    int setup_threshold = 14, leaf_threshold = 20; // BVH kernel scheduling knobs: traversal time slice in inner rounds (0 = none) and
                                                   // early-leaf quorum (FF_SETUP_THRESHOLD / FF_LEAF_THRESHOLD)
    FfStats stats;
    const char* last_kernel_name = nullptr; // trace kernel instantiation of the last frame (rocprofv3's spelling)
    unsigned long long raw_counters[32] = {};
    // GL interop
    hipGraphicsResource* pbo_resource = nullptr;
    int pbo_width = 0, pbo_height = 0;
    // multi-GPU (ff_dist_init): communicator, rank and the packed strip / gather buffers
    FfDistContext* dist = nullptr;
    // a frame enqueued by render_enqueue and not yet finished by render_finish
    int pending_launches = 0;
    uint32_t pending_flags = 0;
    unsigned pending_culled_rays_per_pixel = 0; // primary-ray cull: rays each culled pixel stands for in FfStats::rays_answered
    bool pending = false;
    // fault injection for tests (FF_DEBUG_FAIL_ALLOC=k: the k-th scene allocation of every upload reports out-of-memory)
    int debug_fail_alloc = -1, alloc_countdown = -1;
    // FF_DEBUG_TIMELINE_US=bucket: instrumented launches histogram ray completions over the launch's wall clock (ff_debug_timeline)
    int timeline_bucket_us = 0;
    unsigned* d_timeline = nullptr;
    unsigned h_timeline[1024] = {};
    std::vector<unsigned> h_timeline_rows; // one row per wave, added up by ff_debug_timeline
};


#define FF_HIP(call)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (call);                                                                               \
        if (_e != hipSuccess) return fail(_e == hipErrorOutOfMemory ? FF_ERR_OOM : FF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

namespace ff {

int ensure_bytes(void** ptr, size_t* cap, size_t need);

// What every render entry point checks before it touches the device.
int check_render_call(const FfState* s, const FfCamera* camera, const FfRenderParams* params, const char* who);

// One frame (or this part's strips / one tile of it), in two steps so that several devices can work at once:
// render_enqueue puts clears, trace launches, the combine pass and the counter read-back on the state's stream and
// returns; render_finish waits for them and fills the state's FfStats.  rgb8_dev / radiance_dev are device pointers to
// the LOCAL image (local_rows x win_w).  The window is [x0, x0 + win_w) x (y0 + the strip layout's rows); whole-width
// strips pass x0 = y0 = 0, win_w = -1.
int render_enqueue(FfState* s, const FfCamera* camera, const FfRenderParams* prm, int strip_rows, int part, int num_parts, int local_rows,
                   unsigned char* rgb8_dev, float* radiance_dev, int x0 = 0, int y0 = 0, int win_w = -1);
int render_finish(FfState* s);

void dist_release(FfState* s); // ff_dist.cpp: frees s->dist (called by ff_destroy)

// ff_upload_scene for a scene already compiled on the host (ff_api.cpp).
int upload_compiled_scene(FfState* s, const CompiledScene& cs, double build_ms);

} // namespace ff
