// ff_kernels.h — host-visible launch interface of the gfx950 trace kernels (ff_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "ff_internal.h"

namespace ff {

constexpr int kBlockThreads = 512;        // default workgroup: 8 waves, one workgroup per CU shares one LDS copy of the BVH top
constexpr int kBlockThreadsMax = 1024;    // alternative: 16 waves per workgroup (4 per SIMD), smaller node cache next to the stacks
constexpr int kLdsBudgetBytes = 160 * 1024;
constexpr int kMaxLdsRecords = 128;       // geometry records (288 B each) stay in LDS up to this many: 36 KB of the 160
constexpr int kChunkGeometries = 32;      // BVH mode, up to this many geometries (the reference has 5): records resident in LDS (288 B each), one
                                          // candidate bit per record; beyond it a query walks a tree over the geometries' world boxes (KParams::tlas)
                                          // and reads the records from global memory
constexpr int kBruteBatchTris = 1024;     // triangles staged per LDS batch in brute-force mode (48 KiB)

// Kernel arguments (passed by value; everything here is wave-uniform and lives in SGPRs).
struct KParams {
    // camera: columns of invView*invProj (kernel.cu:203), position, far plane, screen size as floats (kernel.cu:200-201)
    float cam_c0[4], cam_c1[4], cam_c2[4], cam_c3[4];
    float cam_pos[3];
    float far_clip;
    float screen_w, screen_h;
    int width, height; // full image size in pixels (row stride = width)
    int xlim, ylim;    // pixels with x >= xlim or y >= ylim are not traced (FF_GRID_REFERENCE_FLOOR)
    // work decomposition: local rows of this part, in strips
    int strip_rows, part, num_parts, local_rows;
    int x0, y0, local_width; // window inside the image: local pixel (lx, ly) is global (x0 + lx, y0 + row of the strip layout);
                             // whole-width strips: x0 = y0 = 0, local_width = width
    unsigned whole_blocks; // sample blocks this launch hands out as whole (pixel, block) items: block_end - block_begin, minus the tail block
    unsigned total_items; // work items of this launch: pix_items x whole_blocks (+ the tail block's group items)
    unsigned queue_chunk; // items a wave takes from the work queue per atomic, at least (a few hundred samples of work)
    unsigned queue_tail_items; // ... except the last this many of every counter's share, which go out exactly as asked for (acquire_pixel)
    unsigned pix_items;   // 64 per 8x8 pixel tile of the local image (tile padding included)
    int tiles_per_row;
    // integrator
    // samples are accumulated in blocks of block_spp (a block sums its samples sequentially from 0; the blocks of a pixel
    // are summed in order by the combine kernel), which makes (pixel, block) an independent work item
    int bounces, spp_total, block_spp, block_begin, block_end, num_blocks;
    unsigned key;       // Philox key (seed folded to 32 bits)
    int shade_mode;
    // BVH kernel scheduling knobs: setup_threshold = traversal time slice in inner-node rounds (0 = run every query to
    // completion before the wave shades); leaf_threshold = number of lanes holding a leaf that ends an inner-node phase early
    int setup_threshold, leaf_threshold;
    // scene
    int num_geoms;
    int num_planes;  // records [0, num_planes) are analytic shapes (planes, then spheres), the rest meshes (processing order)
    int num_quads;   // records [0, num_quads) are the planes among them
    int has_specular; // some surface is MIRROR or GLASS
    const GeomRecord* geoms;
    const TriRecord* tris;
    const float4* trinormals; // vertex normals (3 float4 per triangle, parallel to tris); null unless FF_SHADE_DIFFUSE_PATH_SMOOTH
    int num_scan;           // big scenes: the records [0, num_scan) are planes kept out of the geometry tree, screened first by every query
    const Bvh4Node* nodes4; // the 4-wide trees of all meshes (each mesh's nodes contiguous, level by level, links relative to its root)
    // scenes of more than kChunkGeometries geometries: the 4-wide tree over the geometries' padded world boxes sits in nodes4
    // from top_first on (leaf link = ~(0x40000000 | record index)); its first top_lds_count nodes are cached in LDS at
    // LDS node index top_lds_first
    int top_first, top_lds_first, top_lds_count;
    int lds_nodes;   // LDS node slots (which nodes of which mesh fill them: GeomRecord::lds_nodes / lds_first)
    int stack_depth; // entries per lane in the LDS traversal stack (depth of the deepest 4-wide tree + 1, or fewer with a spill area)
    int* stack_spill; // the deeper entries of every lane of the launch (global memory, lane-strided; null: the LDS stack holds them all)
    // outputs (local image: local_rows x width)
    float4* blocksums;       // [pix_items][num_blocks] radiance sums of the sample blocks (tile-major pixel order)
    unsigned char* rgb8;     // 3 bytes per local pixel, or null
    float* radiance;         // 3 floats per local pixel, or null
    unsigned* queue;         // work-item counters (zeroed before each launch): counter c sits kQueueStride words behind counter c - 1
    int queue_counters;      // how many of them the launch uses (workgroup b draws from counter b % queue_counters)
    // Fine-grained tail (tail_block < 0: off).  The frame's last sample block is handed out as (pixel, group of
    // tail_group_spp samples) items that store every sample's radiance separately; the combine pass adds that block's
    // samples in order, which is bit for bit what a lane summing the block in registers computes.  The launch then runs
    // dry on 16-sample items instead of 64-sample ones (a 14 ms tail per launch on the benchmark frame otherwise).
    int tail_block;          // global index of the (first) block handled that way: one block, or the frame's last two as one run of samples
    int tail_groups;         // tail items per pixel; group g covers the block's samples [tail_start[g], tail_start[g + 1])
    int tail_start[17];
    int tail_samples_in_block; // samples stored that way: those of the frame's last block (it may be partial), or of its last two
    unsigned tail_first_item; // queue index of the first tail item of this launch
    float4* tail_samples;    // [block_spp][pix_items] per-sample radiance of the tail block
    unsigned long long* counters; // [0] queries cut short by the traversal loop guard (must stay 0) [1] inner-node visits [2] triangle tests [3] plane tests (rays: the slots below)
    // debugging (FF_DEBUG_LDS_FILL=words,pattern): fill that many 4-byte words of dynamic LDS with the pattern before anything is
    // staged, to expose reads of LDS words nobody wrote
    unsigned debug_lds_words, debug_lds_pattern;
    // Primary-ray cull: when the camera sits outside the padded box around all geometries (scene_min / scene_max), cull_mask_kernel
    // marks the pixels whose primary ray misses that box (bit pitem of the mask) and zeroes their block sums; the work queue drops
    // their items (acquire_pixel).  Null: no cull (camera inside; brute-force mode, which stays the reference's loop as written).
    const unsigned long long* cull_mask;
    // ... and, in the frame's pre-pass (shade_mode kShadePrimaryPass) only: the same mask for writing - a pixel whose primary ray hits
    // nothing gets its bit there (settle_hit) - and the number of sample blocks of the frame whose sums are zeroed for it
    unsigned long long* cull_mask_out;
    int frame_blocks;
    // Primary hits (BVH mega-kernels): 3 float4 per pixel item, [3][pix_items] - the closest hit of every pixel's primary ray (distance and
    // point; object-space normal and geometry; triangle record), written by the frame's pre-pass (shade_mode kShadePrimaryPass: one
    // item per pixel) and read by every sample of the frame; null: every primary ray is traced
    float4* primary_hits;
    int reuse_quorum; // lanes of a wave that must wait with a parked primary hit before the wave spends an extra shading pass on them
    // Last segment of a path (bounce index bounces - 1): only an emitter can still add radiance (shade_and_advance).  cut_last != 0:
    // every emitter of the scene is among the analytic records all queries screen first, bit g of emitter_mask says which; a
    // last-bounce query that holds no emitter after that screening ends there (scan_records / begin_segment).
    unsigned emitter_mask;
    int cut_last;
    float scene_min[3], scene_max[3];
    WallTable walls; // axis-aligned planes, screened by a wave-uniform loop (empty for big scenes beyond their first num_scan records)
    // Job-pool kernel (trace_pool_kernel): a wave runs a setup pass when pool_quorum of its lanes are ready for one (pool_quorum_min
    // when the job queue is empty), takes new jobs when pool_refill of its lanes are free, traverses in slices of pool_slice inner rounds
    float4* park; // 7 float4 per thread of the launch, lane-strided: where a lane's own path and query state waits while the lane walks other lanes' jobs
    int pool_quorum, pool_quorum_min, pool_refill, pool_slice, pool_leave, pool_batch_min; // pool_leave: a wave that wants to leave the traverse role puts its jobs down once it holds no more than this many
    // debugging (FF_DEBUG_TIMELINE_US=bucket): instrumented launches count the rays that complete in each bucket of the launch's
    // wall clock (100 MHz ticks since the first wave started; counters[27] holds that epoch), kTimelineBuckets buckets
    unsigned* timeline;
    unsigned timeline_ticks;
};
// The ray count of a launch is added up in kRaySlots 64-bit slots of the counter block, kRaySlotStride words apart, from slot
// kRaySlotFirst on (the first 256 bytes hold the named counters): counters[kRaySlotStride * (kRaySlotFirst + j)].
constexpr int kRaySlots = 30, kRaySlotFirst = 2, kRaySlotStride = 16;
// ... and next to each ray slot the count of those rays that were answered without a traversal: counters[kAnsweredWord + kRaySlotStride * j]
constexpr int kAnsweredWord = kRaySlotStride * kRaySlotFirst + 1;
constexpr int kCutShortWord = kRaySlotStride * kRaySlotFirst + 3;     // (+ kRaySlotStride * j) last-bounce queries that ended after the analytic records
constexpr int kCulledPixelsWord = kRaySlotStride * kRaySlotFirst + 2; // (+ kRaySlotStride * j) cull_mask_kernel: pixels whose items the queue drops
constexpr int kCounterWords = 512; // 64-bit words of the counter block (4 KiB; the work-queue counters follow)
constexpr int kTimelineBuckets = 1024;
// Work queue: up to kQueueCounters counters, 4 KiB apart so that they sit in different memory channels (atomics on one address
// are served one after the other, about 10^8 a second for the whole GPU).
constexpr int kQueueCounters = 64;     // buffer size; a launch uses kQueueCountersDefault of them unless FF_QUEUE_COUNTERS says otherwise
constexpr int kQueueCountersDefault = 16;
constexpr int kQueueStride = 1024; // in 4-byte words
constexpr int kShadePrimaryPass = 100; // KParams::shade_mode of a frame's pre-pass (beyond FfShadeMode's values): every pixel's primary ray, its hit stored (settle_hit)
constexpr int kQueueStripe = 64;   // items: counter c owns the stripes c, c + n, c + 2n, ... of the item range
constexpr int kQueueTailWord = 64; // the counter of a share's last items sits this many words behind its chunk counter (another 256-byte line of the same 4 KiB)

struct RayBatchParams {
    const FfRay* rays;
    FfIntersect* out;
    int n;
    int num_geoms;
    int num_planes;
    int num_quads;
    const GeomRecord* geoms;
    const TriRecord* tris;
    const Bvh4Node* nodes4;
    int top_first, top_lds_first, top_lds_count;
    int lds_nodes;
    int stack_depth;
    int* stack_spill;
    int num_scan;
    WallTable walls;
    unsigned long long* guard_hits; // += 1 per query the traversal loop guard cut short (must stay 0)
};

// LDS bytes the BVH kernels need for (lds_nodes, stack_depth).
size_t bvh_lds_bytes(int lds_nodes, int stack_depth, int block_threads, int num_geoms);
// Largest node count that fits LDS next to a stack of `stack_depth` entries per lane.
int max_lds_nodes(int stack_depth, int block_threads, int num_geoms, size_t reserve = 0);
// LDS the job-pool kernel needs on top of bvh_lds_bytes: the jobs (48 B per thread), the queue ring and its two counters.
size_t pool_lds_bytes(int block_threads);

// block_threads: 512 or 1024 for the BVH kernel; the brute-force kernel always runs 512.
// *kernel_name (optional) receives the name of the instantiation launched, as rocprofv3 prints it.
// pool: the job-pool kernel (scenes of up to kChunkGeometries geometries; the LDS layout must have left pool_lds_bytes free).
hipError_t launch_trace(const KParams& p, int trace_mode, bool collect_stats, int grid_blocks, int block_threads, hipStream_t stream,
                        const char** kernel_name = nullptr, bool pool = false, bool prepass = false);
// Sums every pixel's sample blocks in order, scales by 1/spp and writes radiance / rgb8 (row-major, coalesced).
hipError_t launch_combine(const KParams& p, hipStream_t stream);
// Fills mask[pix_items / 64] (see KParams::cull_mask) and zeroes the block sums of the culled pixels.
hipError_t launch_cull_mask(const KParams& p, unsigned long long* mask, hipStream_t stream);
// sum = first_frame ? frame : sum + frame; mean = sum * inv_frames (and its 8-bit quantisation); `values` floats.
hipError_t launch_accumulate(float* sum, const float* frame, float* mean, unsigned char* rgb8, size_t values, int first_frame, float inv_frames,
                             hipStream_t stream);
hipError_t launch_ray_batch(const RayBatchParams& p, int trace_mode, hipStream_t stream);
// Compares the kernels' lean correctly-rounded 1/x and sqrt(x) with the IEEE expansions on all 2^32 inputs; adds the
// mismatch counts to mismatches2[0] (reciprocal) and [1] (square root).
hipError_t launch_ieee_check(unsigned long long* mismatches2, hipStream_t stream);
hipError_t launch_deinterleave(const void* src, void* dst, int width, int height, int strip_rows, int num_parts, int elem_bytes,
                               hipStream_t stream);
// Scatter of the gathered packed strips (ff_dist.cpp: per part [radiance rows][rgb8 rows], 16-byte padded sections) to
// image order; rgb8 / radiance are the full-frame outputs (either may be null).
hipError_t launch_unpack_strips(const void* src, unsigned char* rgb8, float* radiance, int width, int height, int strip_rows, int num_parts,
                                hipStream_t stream);
hipError_t prepare_kernels(); // one-time function attributes (dynamic LDS limit)

} // namespace ff
