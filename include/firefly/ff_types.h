/*
 * firefly/ff_types.h — plain-C data model shared by the host application and the MI355X
 * path-tracing core.
 *
 * Every struct here is layout-compatible (same field order, same sizeof/offsets on x86-64 LP64)
 * with the reference's host<->device structs so that the reference's viewer code can hand its own
 * arrays to this library by pointer cast, without a glm dependency in the ABI:
 *
 *   FfTriangle   <-> struct Triangle   PathTracer/FireflyEngine/utilities.h:148-171   (96 B)
 *   FfGeometry   <-> struct Geometry   PathTracer/FireflyEngine/utilities.h:173-234   (208 B)
 *   FfBXDF       <-> struct BXDF       PathTracer/FireflyEngine/utilities.h:77-139    (60 B)
 *   FfCamera     <-> struct Camera     PathTracer/FireflyEngine/utilities.h:269-291   (108 B)
 *   FfRay        <-> struct Ray        PathTracer/FireflyEngine/utilities.h:257-267   (24 B)
 *   FfIntersect  <-> struct Intersect  PathTracer/FireflyEngine/utilities.h:57-66     (40 B)
 *   FfScene      <-> struct Scene      PathTracer/FireflyEngine/utilities.h:236-255   (16 B, dead in the reference)
 *
 * glm::vec3 == float[3], glm::vec2 == float[2], glm::mat4 == float[16] column-major
 * (m[col][row] at index col*4+row), exactly glm's default storage.
 */
#ifndef FIREFLY_FF_TYPES_H
#define FIREFLY_FF_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct FfVec2 { float x, y; } FfVec2;
typedef struct FfVec3 { float x, y, z; } FfVec3;
typedef struct FfMat4 { float m[16]; } FfMat4; /* column-major: element (col c, row r) = m[c*4+r] */

/* utilities.h:68-75 */
typedef enum FfBXDFType {
    FF_BXDF_EMITTER = 0,
    FF_BXDF_DIFFUSE = 1,
    FF_BXDF_MIRROR  = 2,
    FF_BXDF_GLASS   = 3,
    FF_BXDF_COUNT   = 4
} FfBXDFType;

/* utilities.h:141-146 */
typedef enum FfGeometryType {
    FF_GEOM_SPHERE       = 0,
    FF_GEOM_PLANE        = 1,
    FF_GEOM_TRIANGLEMESH = 2
} FfGeometryType;

/* utilities.h:77-88 (data members only; bsdf()/pdf() live in the kernel) */
typedef struct FfBXDF {
    int32_t m_type;               /* FfBXDFType; reference default COUNT */
    FfVec3  m_albedo;
    FfVec3  m_specularColor;
    float   m_refractiveIndex;
    FfVec3  m_emissiveColor;
    float   m_intensity;
    FfVec3  m_transmittanceColor;
} FfBXDF;

/* utilities.h:148-171 — only m_v0/m_v1/m_v2 are read on the hot path (kernel.cu:39-41) */
typedef struct FfTriangle {
    FfVec3 m_v0, m_v1, m_v2;
    FfVec2 m_uv0, m_uv1, m_uv2;
    FfVec3 m_n0, m_n1, m_n2;
} FfTriangle;

/* utilities.h:219-233 */
typedef struct FfGeometry {
    int32_t     m_geometryType;       /* FfGeometryType */
    FfVec3      m_position;
    FfVec3      m_rotation;           /* degrees, per axis (utilities.h:182-184) */
    FfVec3      m_scale;
    FfMat4      m_modelMatrix;        /* T * Rx * Ry * Rz * S (utilities.h:187) */
    FfMat4      m_inverseModelMatrix; /* glm::inverse(m_modelMatrix) (utilities.h:189) */
    float       m_sphereRadius;
    FfVec3      m_normal;             /* object-space plane normal, default (0,0,1) (utilities.h:229) */
    FfTriangle* m_triangles;          /* host pointer, owned by the caller */
    int32_t     m_numberOfTriangles;
    FfBXDF*     m_bxdf;               /* host pointer, owned by the caller */
} FfGeometry;

/* utilities.h:236-255 */
typedef struct FfScene {
    FfGeometry* m_geometries;
    int32_t     m_geometrySize;
} FfScene;

/* utilities.h:257-267 */
typedef struct FfRay {
    FfVec3 m_origin;
    FfVec3 m_direction;
} FfRay;

/* utilities.h:57-66 */
typedef struct FfIntersect {
    FfVec3  m_intersectionPoint;
    FfVec3  m_normal;
    float   m_t;
    uint8_t m_hit;                /* C++ bool */
    uint8_t _pad[3];
    int32_t geometryIndex;
    int32_t triangleIndex;
} FfIntersect;

/* utilities.h:269-291 (data members only) */
typedef struct FfCamera {
    FfVec3  m_position;
    FfVec3  m_up;
    FfVec3  m_right;
    FfVec3  m_forward;
    FfVec3  m_worldUp;
    float   m_yaw;
    float   m_pitch;
    float   m_screenWidth;
    float   m_screenHeight;
    float   m_fov;                /* degrees */
    float   m_nearClip;
    float   m_farClip;
    float   m_cameraMovementSpeed;
    float   m_cameraMouseSensitivity;
    uint8_t m_cameraFirstMouseInput; /* C++ bool */
    uint8_t _pad[3];
    float   m_xDelta;
    float   m_yDelta;
} FfCamera;

/* ---- build-defined parameter blocks (the reference hard-codes these in main(), kernel.cu:261-266,306-309) ---- */

/* How the closest hit is searched.  Both produce bit-identical hits; see DESIGN.md. */
typedef enum FfTraceMode {
    FF_TRACE_BRUTE_FORCE = 0, /* reference loop order kernel.cu:133-155, triangle batches staged in LDS */
    FF_TRACE_BVH         = 1  /* per-mesh object-space BVH, top levels resident in LDS */
} FfTraceMode;

typedef enum FfShadeMode {
    FF_SHADE_NORMAL_DEBUG = 0, /* kernel.cu:178-184: colour = abs(world normal); bounces/spp forced to 1 */
    FF_SHADE_DIFFUSE_PATH = 1, /* N-bounce integrator with the dormant BXDF semantics, utilities.h:90-138 */
    FF_SHADE_DIFFUSE_PATH_SMOOTH = 2 /* the same, shading triangles with the interpolated vertex normals Triangle carries
                                      * (utilities.h:163-170) through the barycentrics of kernel.cu:80-81 */
} FfShadeMode;

typedef enum FfGridMode {
    FF_GRID_FULL            = 0, /* every pixel of the W x H image is traced */
    FF_GRID_REFERENCE_FLOOR = 1  /* reproduce kernel.cu:308-309: only floor(W/16)*16 x floor(H/16)*16 pixels traced */
} FfGridMode;

typedef struct FfRenderParams {
    int32_t  width;       /* image width  W in pixels (row stride) */
    int32_t  height;      /* image height H in pixels */
    int32_t  bounces;     /* max path segments per sample, >= 1 (1 == the reference's primary-only behaviour) */
    int32_t  spp;         /* samples per pixel, >= 1 */
    uint64_t seed;        /* RNG seed (reference literal 1234, utilities.h:118) */
    int32_t  trace_mode;  /* FfTraceMode */
    int32_t  shade_mode;  /* FfShadeMode */
    int32_t  grid_mode;   /* FfGridMode */
    int32_t  spp_per_launch; /* 0 = all samples in one kernel launch; otherwise a frame is rendered in several launches of about
                                this many samples (rounded up to whole 64-sample blocks); the result does not depend on it */
} FfRenderParams;

/* Filled by ff_stats() after a render call. Counts are for the LAST ff_render* call on this state. */
typedef struct FfStats {
    uint64_t rays_traced;        /* closest-hit queries = path segments of the frame, counted on the device (wave-reduced counter) */
    uint64_t nodes_visited;      /* visits of 4-wide BVH nodes (112 B each); only when stats collection is on */
    uint64_t tris_tested;        /* ray/triangle tests (48 B each); only when stats collection is on */
    uint64_t planes_tested;      /* ray/plane tests */
    double   kernel_ms;          /* sum of trace-kernel durations, HIP events on the launch stream */
    double   total_ms;           /* host wall clock of the whole call */
    uint32_t kernel_launches;    /* number of trace-kernel launches in the call */
    uint32_t flags;              /* FF_STATS_* bits about how the frame was scheduled */
    uint64_t scene_bytes_nodes;  /* device bytes of BVH nodes */
    uint64_t scene_bytes_tris;   /* device bytes of triangle records */
    uint64_t rays_answered;      /* ... of rays_traced: path segments answered without a traversal - the primary segments of a frame of more than
                                    one sample per pixel (every sample of a pixel starts with the same ray, kernel.cu:200-205: a pre-pass traces
                                    it once per pixel, its rays are not counted) incl. those of pixels whose view of the scene box is empty
                                    (camera outside the scene); 0 in brute-force mode */
    uint64_t rays_cut_short;     /* ... of rays_traced: last segments of paths (bounce index bounces - 1) whose query ended after the planes and
                                    spheres because no emitter was among the candidates - only an emitter can still add radiance there and
                                    every emitter of the scene is a plane or a sphere; 0 in brute-force mode */
} FfStats;
#define FF_STATS_TAIL_ITEMS 1u             /* the frame's last sample block was handed out as fine-grained items (multi-part frames) */
#define FF_STATS_TAIL_SKIPPED_TOO_LARGE 2u /* ... was wanted, but its per-sample buffer would pass 16 GiB: rendered with whole-block items */

/* Which builder produces the BVH (ff_set_builder). */
typedef enum FfBuilder {
    FF_BUILD_HOST_SAH = 0, /* binned SAH on the host: best trees, for scenes uploaded once (the reference's case, kernel.cu:268-298) */
    FF_BUILD_GPU_LBVH = 1, /* Morton-code LBVH built on the device: for geometry that changes between frames */
    FF_BUILD_GPU_PLOC = 2  /* agglomerative clustering (PLOC) on the device: a few times the LBVH's build time, trees close to SAH quality */
} FfBuilder;

/* ff_update_mesh modes. */
typedef enum FfMeshUpdate {
    FF_UPDATE_REFIT   = 0, /* keep the tree, recompute its boxes from the moved vertices */
    FF_UPDATE_REBUILD = 1  /* rebuild the mesh's tree on the device (scene must have been uploaded with a device builder) */
} FfMeshUpdate;

/* Filled by ff_build_stats(): the last ff_upload_scene / ff_update_mesh / ff_update_transforms call. */
typedef struct FfBuildStats {
    int32_t  builder;          /* FfBuilder that produced the current trees */
    int32_t  bvh_nodes;        /* 64-byte binary nodes in use over all meshes (the builders' trees; the kernels traverse the 4-wide trees derived from them) */
    int32_t  bvh_max_depth;    /* deepest root-to-leaf path, in inner nodes */
    int32_t  last_operation;   /* 0 upload, 1 refit, 2 rebuild, 3 transforms */
    uint64_t num_triangles;
    double   total_ms;         /* host wall clock of the whole call */
    double   copy_ms;          /* of which: host-to-device copies of the caller's triangles */
    double   build_ms;         /* of which: tree construction / refit (host builder: on the CPU; device builder: kernels, synchronised) */
} FfBuildStats;

/* Filled by ff_scene_info(): what ff_upload_scene would build for a host scene (no GPU needed). */
typedef struct FfSceneInfo {
    int32_t  num_geometries;
    int32_t  num_meshes;
    int32_t  num_planes;
    int32_t  bvh_nodes;        /* 64-byte inner nodes over all meshes */
    int32_t  bvh_max_depth;    /* deepest root-to-leaf path, in inner nodes */
    int32_t  bvh_max_leaf;     /* largest leaf, in triangles */
    int32_t  lds_nodes;        /* nodes that stay resident in LDS for this scene */
    int32_t  lds_bytes;        /* LDS bytes per workgroup (nodes + traversal stacks) */
    uint64_t num_triangles;
    uint64_t device_bytes;     /* total device bytes of the compiled scene */
    int32_t  valid;            /* 1 if the structural self-check passed (every triangle in exactly one leaf, boxes enclose) */
    float    bvh_child_area;   /* summed half surface area of every child box of the binary trees (object space, padded): what the
                                  SAH build and the insertion-based optimisation pass reduce */
} FfSceneInfo;

#ifdef __cplusplus
} /* extern "C" */
#endif

#if defined(__cplusplus)
static_assert(sizeof(FfBXDF) == 60, "BXDF layout must match utilities.h:77-88");
static_assert(sizeof(FfTriangle) == 96, "Triangle layout must match utilities.h:148-171");
static_assert(sizeof(FfGeometry) == 208, "Geometry layout must match utilities.h:219-233");
static_assert(sizeof(FfRay) == 24, "Ray layout must match utilities.h:257-267");
static_assert(sizeof(FfIntersect) == 40, "Intersect layout must match utilities.h:57-66");
static_assert(sizeof(FfCamera) == 108, "Camera layout must match utilities.h:269-291");
#else
_Static_assert(sizeof(FfBXDF) == 60, "BXDF layout");
_Static_assert(sizeof(FfTriangle) == 96, "Triangle layout");
_Static_assert(sizeof(FfGeometry) == 208, "Geometry layout");
_Static_assert(sizeof(FfRay) == 24, "Ray layout");
_Static_assert(sizeof(FfIntersect) == 40, "Intersect layout");
_Static_assert(sizeof(FfCamera) == 108, "Camera layout");
#endif

#endif /* FIREFLY_FF_TYPES_H */
