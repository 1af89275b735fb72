/*
 * firefly/ff_api.h — C ABI of the MI355X path-tracing core (libfirefly_hip.so).
 *
 * The reference has no library seam: host and device code share one translation unit
 * (PathTracer/FireflyEngine/kernel.cu).  This ABI is cut exactly at the two places where the
 * reference's main() talks to the GPU:
 *
 *   scene upload   kernel.cu:268-298   -> ff_upload_scene
 *   per-frame      kernel.cu:335-344   -> ff_render_to_pbo (GL viewer) / ff_render (headless twin)
 *   PBO register   utilities.h:605-618 -> ff_register_gl_pbo,  utilities.h:516 -> ff_unregister_gl_pbo
 *   kernel         kernel.cu:218-221   -> the trace kernels behind ff_render*
 *
 * Conventions: plain pointers and sizes only; every function returns an FfStatus (0 = ok) and never
 * calls exit() (the reference's cudaCheckErrors macro does, utilities.h:27-37); the message for the
 * last failure on the calling thread is available from ff_last_error().  All ff_render* calls are
 * synchronous on return, which is the implicit contract the viewer relies on before glTexSubImage2D
 * (kernel.cu:344-351).  The caller owns the host scene and all GL objects; the library owns all
 * device memory it allocates.
 */
#ifndef FIREFLY_FF_API_H
#define FIREFLY_FF_API_H

#include "ff_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define FF_API __declspec(dllexport)
#else
#define FF_API __attribute__((visibility("default")))
#endif

typedef enum FfStatus {
    FF_OK                 = 0,
    FF_ERR_INVALID_ARG    = 1,
    FF_ERR_NO_DEVICE      = 2,  /* no HIP device / HIP runtime failure at create */
    FF_ERR_HIP            = 3,  /* a HIP call failed; see ff_last_error() */
    FF_ERR_NO_SCENE       = 4,  /* render called before ff_upload_scene */
    FF_ERR_UNSUPPORTED    = 5,  /* e.g. a non-affine model matrix, an unknown geometry type (kernel.cu:170-173) */
    FF_ERR_GL_UNAVAILABLE = 6,  /* HIP-GL interop not usable (headless box) */
    FF_ERR_IO             = 7,  /* file could not be read / parsed */
    FF_ERR_OOM            = 8,
    FF_ERR_COMM           = 9   /* RCCL unavailable or a collective call failed (multi-GPU entry points) */
} FfStatus;

typedef struct FfState FfState; /* opaque; replaces PathTracerState (kernel.h:12-20) */

/* ---- lifetime --------------------------------------------------------------------------------- */

/* Create a tracer bound to HIP device `device_id`. */
FF_API int ff_create(FfState** out_state, int device_id);
FF_API int ff_destroy(FfState* state);

/* Message of the last failure on this thread ("" if none). Never NULL. */
FF_API const char* ff_last_error(void);

/* Library/ABI version: major*10000 + minor*100 + patch. */
FF_API int ff_version(void);

/* Launch stream for all subsequent work of this state (a hipStream_t; NULL = default stream,
 * which is what the reference uses, kernel.cu:342). */
FF_API int ff_set_stream(FfState* state, void* hip_stream);

/* ---- host-side helpers restating the reference's host code (bit-exact glm operation order) ---- */

/* Geometry::Geometry(type, position, rotation, scale, triangles, radius)  utilities.h:176-213.
 * Builds m_modelMatrix = T*Rx*Ry*Rz*S and its inverse.  `triangles` is borrowed, not copied. */
FF_API void ff_geometry_init(FfGeometry* g, int geometry_type, FfVec3 position, FfVec3 rotation_deg,
                             FfVec3 scale, FfTriangle* triangles, int number_of_triangles, float radius);

/* BXDF default member initialisers utilities.h:81-88. */
FF_API void ff_bxdf_init(FfBXDF* b);

/* Camera default member initialisers utilities.h:287-291 plus the literals of kernel.cu:312-322
 * (position (0,0,15), worldUp (0,1,0), fov 70, near .1, far 1000, yaw -90, pitch 0) for a W x H image,
 * followed by UpdateBasisAxis. Width/height are assigned un-swapped (see SURVEY hazard 2). */
FF_API void ff_camera_init_default(FfCamera* c, int width, int height);

/* Camera::UpdateBasisAxis  utilities.h:407-418. */
FF_API void ff_camera_update_basis(FfCamera* c);

/* invView * invProj of kernel.cu:203 (lookAtRH, perspectiveFovRH_NO, two inverses, one mat*mat),
 * hoisted out of the per-pixel path.  Exposed for tests. */
FF_API void ff_camera_ray_matrix(const FfCamera* c, FfMat4* out_inv_view_times_inv_proj);

/* ---- scene upload (kernel.cu:268-298) ---------------------------------------------------------- */

/* Deep-copies `n` geometries (and the triangles / BXDFs they point to) to the device, flattened to
 * 48-byte triangle records, per-geometry transform records and one object-space BVH per mesh.
 * The host arrays are only read and may be freed after the call.  A second call replaces the scene
 * and frees the previous one (the reference leaks it, kernel.cu:364-368).
 * SPHERE geometries (declared by the reference, utilities.h:193-195, but only printf'ed by its kernel, kernel.cu:166-169)
 * are intersected analytically in object space: radius m_sphereRadius about the origin, two-sided. */
FF_API int ff_upload_scene(FfState* state, const FfGeometry* host_geometries, int n);

/* ---- dynamic scenes (no counterpart in the reference, whose upload is one-off) ------- */

/* Selects the BVH builder used by the following ff_upload_scene calls (default FF_BUILD_HOST_SAH).  Rendered results do
 * not depend on the builder: every hit is decided by the reference arithmetic, the tree only prunes. */
FF_API int ff_set_builder(FfState* state, int builder);

/* The same geometries as the last upload (same count, types and triangle counts) with new transforms / materials:
 * rewrites the per-geometry records only.  Trees live in object space, so moving, rotating or scaling a mesh costs no
 * rebuild. */
FF_API int ff_update_transforms(FfState* state, const FfGeometry* host_geometries, int n);

/* New vertex positions for mesh `geometry_index` (index in the uploaded Geometry array; `count` must equal the uploaded
 * triangle count).  FF_UPDATE_REFIT keeps the tree and recomputes its boxes on the device; FF_UPDATE_REBUILD builds a
 * new tree on the device in the mesh's slot. */
FF_API int ff_update_mesh(FfState* state, int geometry_index, const FfTriangle* triangles, int count, int mode);

FF_API int ff_build_stats(FfState* state, FfBuildStats* out_stats);

/* Copies the compiled scene back for inspection (tests): up to `max_nodes` 64-byte binary nodes and `max_tris` 48-byte triangle
 * records; the counts in use are returned through out_nodes / out_tris.  mesh_table (optional) receives, for each of the
 * first `max_geometries` uploaded geometries in the caller's order, {bvh_root, node_count, tri_first, tri_count, depth}
 * (bvh_root = -1 for planes and empty meshes).  Buffers may be null to query the counts. */
FF_API int ff_debug_download_bvh(FfState* state, void* nodes, int max_nodes, int* out_nodes, void* tris, int max_tris, int* out_tris,
                                 int* mesh_table, int max_geometries);

/* The 4-wide trees the trace kernels traverse (derived on the device from the binary trees above): up to `max_nodes`
 * 112-byte nodes of the whole array (*out_capacity = its length; mesh i's nodes start at its binary root's index and link
 * to each other relative to it); mesh_table (optional) receives per uploaded geometry {first node, node count, depth,
 * first LDS slot, nodes cached in LDS, LDS node slots of the scene}. */
FF_API int ff_debug_download_bvh4(FfState* state, void* nodes4, int max_nodes, int* out_capacity, int* mesh_table, int max_geometries);

/* Host-only dry run of the scene compiler: sizes, BVH shape and a structural self-check.  Needs no GPU. */
FF_API int ff_scene_info(const FfGeometry* host_geometries, int n, FfSceneInfo* out_info);

/* Host-only: the planes the trace kernels screen in WORLD space - those whose model matrix maps the unit quad (kernel.cu:18) onto a
 * rectangle parallel to two world axes (rotations by multiples of 90 degrees, any translation, scales within 1 : 16): every wall
 * of a box scene.  Per wall 7 floats {caller's geometry index, normal axis 0/1/2, plane coordinate, centre and half extent along
 * the next axis, centre and half extent along the one after}; walls sorted by axis.  Returns the number of walls (at most 16 of a
 * scene's first 32 planes) or minus an FfStatus.  Results never depend on the table: a plane it leaves out, and every hit inside
 * its margins, goes through the exact reference test.  Needs no GPU. */
FF_API int ff_debug_wall_table(const FfGeometry* host_geometries, int n, float* out_walls7, int max_walls);
/* ... and the number of ENTRIES of that table: two walls normal to the same axis with the same rectangle (floor and ceiling of a box)
 * share one, so this is at most the count above.  Needs no GPU. */
FF_API int ff_debug_wall_entries(const FfGeometry* host_geometries, int n);

/* ---- rendering (kernel.cu:335-344 + launchPathTrace kernel.cu:218-221) -------------------------- */

/* Headless twin of the per-frame block.  Outputs (either may be NULL):
 *   rgb8      W*H*3 bytes, row-major, top row first, exactly the PBO contents the viewer uploads
 *             (kernel.cu:214; miss pixels stay 0 like the cudaMemset at kernel.cu:340)
 *   radiance  W*H*3 floats, mean radiance per pixel before 8-bit quantisation
 * `*_on_device` != 0 means the pointer is device memory of this state's device; otherwise it is host
 * memory and the library copies back (PCIe-inclusive). */
FF_API int ff_render(FfState* state, const FfCamera* camera, const FfRenderParams* params,
                     void* rgb8, int rgb8_on_device, float* radiance, int radiance_on_device);

/* Multi-GPU slice of the same frame: the image is cut into strips of `strip_rows` rows; this call
 * renders the strips s with s % num_parts == part and writes them compacted, in increasing s, into
 * rgb8/radiance (device or host as above), each local row W pixels wide.  part=0,num_parts=1 is
 * ff_render.  The RNG is keyed on the GLOBAL pixel index, so the union over parts is bit-identical to
 * the single-GPU image.  *out_local_rows receives the number of rows written (may be NULL). */
FF_API int ff_render_strips(FfState* state, const FfCamera* camera, const FfRenderParams* params,
                            int strip_rows, int part, int num_parts,
                            void* rgb8, int rgb8_on_device, float* radiance, int radiance_on_device,
                            int* out_local_rows);

/* One rectangular tile [x0, x0 + w) x [y0, y0 + h) of the frame into compact w x h buffers (row stride w): the other way
 * to partition a frame (SURVEY.md section 8b's ff_render_tile).  Pixels are identical to the same pixels of ff_render:
 * the random numbers are keyed on the global pixel index. */
FF_API int ff_render_tile(FfState* state, const FfCamera* camera, const FfRenderParams* params, int x0, int y0, int w, int h, void* rgb8,
                          int rgb8_on_device, float* radiance, int radiance_on_device);

/* Number of rows ff_render_strips writes for (height, strip_rows, part, num_parts). */
FF_API int ff_strips_local_rows(int height, int strip_rows, int part, int num_parts);

/* Scatter gathered per-part compact buffers back to image order on the device:
 * src holds, for part p = 0..num_parts-1 in order, that part's compact rows; elem_bytes is the size of
 * one pixel (3 for rgb8, 12 for float3 radiance). */
FF_API int ff_deinterleave_strips(FfState* state, const void* src_dev, void* dst_dev, int width, int height,
                                  int strip_rows, int num_parts, int elem_bytes);

/* ---- multi-GPU frames (no counterpart in the reference: its only trace of more than one GPU is the dead
 *      `const bool multi_gpu` of utilities.h:484-487; BASELINE north star: image tiles over the GPUs of a node, RCCL
 *      gather of the framebuffer over xGMI) ---------------------------------------------------------------------- */

/* Shape 1 — one process per GPU.  Rank 0 obtains an id (128 bytes) and hands it to the other ranks by any means it has
 * (a file, a socket, MPI, torch.distributed); every rank then joins with its own state.  One RCCL communicator per state. */
#define FF_DIST_ID_BYTES 128
FF_API int ff_dist_unique_id(void* out_id, int bytes);
FF_API int ff_dist_init(FfState* state, int rank, int world_size, const void* id, int bytes);
/* FF_OK if the RCCL library can be loaded in this process (what ff_dist_unique_id / ff_dist_init need).  Lets the ranks of a
 * job agree that all of them can join BEFORE any of them enters ff_dist_init, which blocks until every rank has. */
FF_API int ff_dist_available(void);
FF_API int ff_dist_shutdown(FfState* state);

/* Strip height the distributed renderers use when given strip_rows <= 0: of 1 .. 16 rows the height whose largest part has the
 * fewest rows (the slowest rank sets the frame time), the thinnest such of at least two rows: for 1080 rows 2-row strips over 2 or 4 GPUs, 3-row strips over 8
 * (equal shares: 135 rows each there).  ff_dist_strip_rows(world) is ff_dist_strip_rows_for(1080, world). */
FF_API int ff_dist_strip_rows_for(int height, int world_size);
FF_API int ff_dist_strip_rows(int world_size);

/* The gather's wire layout: bytes of the ONE message part `part` sends to rank 0 (its rows as float3 radiance, then as rgb8,
 * each section padded to 16 bytes; 0 for a part that owns no rows: then no message is posted on either side) and, through
 * out_offset (may be null), where it lands in rank 0's gather buffer.  -1 for invalid arguments. */
FF_API long long ff_dist_part_bytes(int width, int height, int strip_rows, int part, int num_parts, long long* out_offset);

/* One frame over all ranks; every rank calls it with the same camera and params.  Each rank renders the strips
 * s % world == rank (ff_render_strips' partition) into one packed buffer and sends it to rank 0 in a single message;
 * rank 0 receives every peer's message into its gather buffer (one grouped RCCL call) and scatters all strips to image
 * order.  On rank 0, rgb8 / radiance receive the full frame, bit-identical to ff_render's (device or host pointers, as
 * there); on other ranks they are ignored.  Synchronous on return on every rank; ff_stats reports this rank's share.
 *
 * Errors.  No rank is left waiting for a message that cannot come: every rank first does what can fail locally (argument
 * checks, buffers, the launches of its strips), then the ranks agree on a status (a 4-byte all-reduce), and only a frame that
 * is well on every rank is gathered.  A local failure returns its own status on the rank it happened on and FF_ERR_COMM on
 * every other rank; the communicator stays usable.  Waits on other ranks have a deadline (FF_DIST_TIMEOUT_S in the
 * environment at ff_dist_init, default 300 seconds) and watch ncclCommGetAsyncError: a missing or dead peer ends the call with
 * FF_ERR_COMM, the communicator is aborted (ncclCommAbort), and every later call returns FF_ERR_COMM until ff_dist_init has
 * made a new one.  After such an error the process should exit; a fresh process is the retry. */
FF_API int ff_render_distributed(FfState* state, const FfCamera* camera, const FfRenderParams* params, int strip_rows,
                                 void* rgb8, int rgb8_on_device, float* radiance, int radiance_on_device);
/* Tests: from the next frame on, rank `rank` of the job reports an injected local failure (FF_ERR_OOM) before its strips are
 * enqueued; -1 switches it off.  FF_DEBUG_DIST_FAIL_RANK in the environment at ff_dist_init sets the initial value. */
FF_API int ff_debug_dist_fail_rank(FfState* state, int rank);

/* Shape 2 — one process, several GPUs: what a single-process viewer (the reference's main(), kernel.cu:223-368) calls.
 * ff_multi_create makes one state per entry of device_ids, each with its own stream; device_ids[0] is the gathering device
 * (the one whose GL context owns the pixel buffer).  Transport: RCCL (ncclCommInitAll + grouped send/recv); peer copies
 * (hipMemcpyPeerAsync) when a device id repeats — RCCL refuses that, and it is how a one-GPU box rehearses the path — or
 * when FF_MULTI_TRANSPORT=peer. */
typedef struct FfMulti FfMulti;
FF_API int ff_multi_create(FfMulti** out_multi, const int* device_ids, int n);
FF_API int ff_multi_destroy(FfMulti* multi);
FF_API int ff_multi_count(const FfMulti* multi);
FF_API FfState* ff_multi_state(FfMulti* multi, int index);       /* per-device calls: ff_register_gl_pbo on index 0, ff_set_builder, ... */
FF_API int ff_multi_uses_rccl(const FfMulti* multi);
FF_API int ff_multi_upload_scene(FfMulti* multi, const FfGeometry* host_geometries, int n); /* replicated on every device */
/* The frame: all devices render their strips at once, device 0 gathers.  Outputs as in ff_render, on / from device 0. */
FF_API int ff_multi_render(FfMulti* multi, const FfCamera* camera, const FfRenderParams* params, int strip_rows,
                           void* rgb8, int rgb8_on_device, float* radiance, int radiance_on_device);
/* kernel.cu:335-344 with all GPUs behind it: the pixel buffer registered on ff_multi_state(multi, 0). */
FF_API int ff_multi_render_to_pbo(FfMulti* multi, const FfCamera* camera, const FfRenderParams* params, int strip_rows);
/* Sums over the devices of the last frame (kernel_ms: the slowest device). */
FF_API int ff_multi_stats(FfMulti* multi, FfStats* out);

/* Batch closest-hit query = intersectRays (kernel.cu:127-176) for `n` arbitrary world-space rays.
 * rays/out are host arrays. trace_mode is an FfTraceMode. */
FF_API int ff_intersect_rays(FfState* state, const FfRay* rays, int n, FfIntersect* out, int trace_mode);

/* ---- OpenGL pixel-buffer interop (utilities.h:605-618, kernel.cu:335-344) ----------------------- */

/* cudaGraphicsGLRegisterBuffer(&res, pbo, WriteDiscard) twin via hipGraphicsGLRegisterBuffer.
 * The GL context that owns `pbo` must be current on the calling thread.  The buffer must hold
 * width*height*3 bytes. */
FF_API int ff_register_gl_pbo(FfState* state, unsigned int pbo, int width, int height);
FF_API int ff_unregister_gl_pbo(FfState* state);

/* map -> clear -> trace -> unmap  (kernel.cu:337-344).  params->width/height must match the registration. */
FF_API int ff_render_to_pbo(FfState* state, const FfCamera* camera, const FfRenderParams* params);

/* ---- progressive accumulation (the reference's README sketches it) ------------------- */

/* Frame `frame_index` (0, 1, 2, ... while the camera is still) of a progressive sequence: an ordinary frame rendered with
 * seed + frame_index, added in fp32 to a running per-pixel sum kept in the state; the outputs are sum * (1 / frames) and
 * its 8-bit quantisation.  frame_index 0 restarts the sequence (call it when the camera or the scene changed); other
 * indices must continue it.  Buffers as in ff_render. */
FF_API int ff_render_progressive(FfState* state, const FfCamera* camera, const FfRenderParams* params, int frame_index, void* rgb8, int rgb8_on_device,
                                 float* radiance, int radiance_on_device);

/* The same into the registered pixel buffer (the per-frame block of kernel.cu:335-344). */
FF_API int ff_render_to_pbo_progressive(FfState* state, const FfCamera* camera, const FfRenderParams* params, int frame_index);

/* saveToPPM (utilities.h:842-856) for the 8-bit framebuffer: P3 text, one "r g b" line per pixel, top row first. */
FF_API int ff_save_ppm(const char* path, const unsigned char* rgb8, int width, int height);

/* ---- measurement -------------------------------------------------------------------------------- */

/* Turn per-launch node/triangle visit counters on (1) or off (0, default).  Ray counting is always on. */
FF_API int ff_set_collect_stats(FfState* state, int on);
FF_API int ff_stats(FfState* state, FfStats* out);

/* Name of the trace-kernel instantiation the last frame launched, spelled as rocprofv3 prints it ("" before the first frame). */
FF_API const char* ff_debug_kernel_name(FfState* state);

/* Raw device counters of the last instrumented render (ff_set_collect_stats(1)), 28 values: [0] rays [1] inner-node
 * visits [2] triangle tests [3] plane tests (all summed over lanes) [8] inner-step rounds [9] leaf rounds [10] triangle
 * rounds [11] plane rounds [12] segment rounds (wave-level executions of each phase): lanes / (64 * rounds) is the SIMD
 * occupancy of that phase; [4..7],[13] wave cycles spent in resolve / shade / acquire / begin / traverse; [14] plane-only queries [15] exact plane
 * tests; [16..18] wave cycles in mesh starts / inner-node phases / leaf phases; [19..21] the three parts of the begin phase; [22] the
 * slowest wave's loop cycles; [23..25] 100 MHz wall clock, complemented / complemented / plain: first lane to find the work
 * queue empty, first wave start, last wave end; [28..30] wave cycles of the leaf visits spent waiting for the
 * triangle records / in the triangle tests / in the pop that follows. */
FF_API int ff_debug_counters(FfState* state, unsigned long long* out32);

/* With FF_DEBUG_TIMELINE_US=<bucket> in the environment at ff_create, instrumented renders also count the rays that complete in
 * each bucket of the (first) launch's wall clock: 1 024 buckets from the start of the first wave, the last one open-ended.
 * *bucket_us comes back 0 when the histogram is off. */
FF_API int ff_debug_timeline(FfState* state, unsigned* out1024, int* bucket_us);

/* Self-check of the kernels' arithmetic: their correctly rounded 1/x and sqrt(x) against the compiler's IEEE expansions on
 * every one of the 2^32 float bit patterns; out_mismatches2[0] / [1] must come back 0 (a few milliseconds). */
FF_API int ff_debug_check_ieee(FfState* state, unsigned long long* out_mismatches2);
/* The experiment switches (FF_NO_PRIMARY_REUSE, FF_NO_LAST_BOUNCE_CUT, FF_NO_WALL_TABLE, FF_POOL, ...) are read from the
 * environment once, at ff_create; this re-reads them for `state` (A/B tests that flip a switch between two frames of one state).
 * Layout switches take effect at the next upload or transform update. */
FF_API int ff_debug_reload_switches(FfState* state);

/* ---- mesh loading (next-row scope: LoadMesh, utilities.h:781-840) ------------------------------- */

/* Reads a Wavefront OBJ with LoadMesh's semantics: one FfTriangle per face from the face's first three
 * indexed vertices; missing vt/vn yield zeros instead of the reference's out-of-bounds read.
 * *out_triangles is malloc'ed by the library; release with ff_free_triangles. */
FF_API int ff_load_obj(const char* path, FfTriangle** out_triangles, int* out_count);
FF_API void ff_free_triangles(FfTriangle* triangles);

/* ---- scene description file (next-row scope: the reference's "TODO: Load scene from file", kernel.cu:261) -------- */

/* A scene file replaces the literals of kernel.cu:227-259 (geometries, BXDFs) and kernel.cu:311-321 (camera).  Plain text,
 * one statement per line, '#' starts a comment:
 *
 *   camera position X Y Z yaw DEG pitch DEG fov DEG near N far F            (every key optional: kernel.cu:312-321 defaults)
 *   bxdf NAME diffuse|emitter|mirror|glass [albedo R G B] [specular R G B] [transmittance R G B] [ior N] [color R G B] [intensity I]
 *   mesh FILE.obj [position X Y Z] [rotation X Y Z] [scale X Y Z] bxdf NAME (path relative to the scene file)
 *   plane [position X Y Z] [rotation X Y Z] [scale X Y Z] bxdf NAME
 *   sphere radius R [position X Y Z] [rotation X Y Z] [scale X Y Z] bxdf NAME
 *
 * Geometries keep file order (it is the reference's iteration order, kernel.cu:133).  The returned object owns the
 * triangles and BXDFs its FfGeometry array points to. */
typedef struct FfSceneFile FfSceneFile;
FF_API int ff_scene_file_load(const char* path, FfSceneFile** out_scene);
FF_API const FfGeometry* ff_scene_file_geometries(const FfSceneFile* scene, int* out_count);
/* Camera of the file for a width x height image (UpdateBasisAxis applied). */
FF_API int ff_scene_file_camera(const FfSceneFile* scene, int width, int height, FfCamera* out_camera);
FF_API void ff_scene_file_free(FfSceneFile* scene);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* FIREFLY_FF_API_H */
