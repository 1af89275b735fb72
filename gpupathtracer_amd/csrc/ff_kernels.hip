// ff_kernels.hip — the gfx950 (CDNA4, wave64) trace kernels.
//
// What the reference runs per pixel (kernel.cu:186-221: primary ray, brute-force closest hit over all geometries and
// triangles, shade, 8-bit store) is restructured here as a persistent mega-kernel:
//
//   * one workgroup per CU; the top of every mesh's 4-wide tree (the whole tree where it fits: the benchmark scene's does)
//     and all geometry records are staged ONCE per workgroup into LDS (112-byte nodes: six box planes for four slots + four
//     links, kept as seven planes of 16-byte quarters so that a wave's reads spread over all banks) and every lane keeps
//     its traversal stack in LDS (lane-strided: pushes / pops are bank-conflict free; one entry per visited node);
//   * lanes pull (pixel, sample block) work items with a wave-wide ballot + prefix compaction from the wave's own chunk of
//     the work queue (a chunk per atomic, 16 counters in different memory channels: acquire_pixel); a lane sums its block's
//     samples in order, terminated paths regenerate in place, and a combine pass adds a pixel's blocks in order;
//   * traversal is time-sliced: after a budget of inner-node rounds the lanes whose query is complete resolve, shade and
//     spawn their next ray TOGETHER while the long-tail lanes keep their traversal state; inside a slice the wave
//     alternates inner-node phases and leaf phases.  This keeps the 64 lanes occupied although neighbouring rays need
//     very different amounts of work;
//   * the per-pixel camera matrix work of kernel.cu:203 is hoisted to the host; the per-hit 4x4 inverse of
//     kernel.cu:117 is hoisted to the scene compiler.
//
// Numerics: the file is compiled with -ffp-contract=off and IEEE-correct sqrt/divide (1/x and sqrt through lean sequences
// that are verified bit-identical to the IEEE expansions on all 2^32 inputs).  Every value that decides or
// becomes part of a hit (object-space ray, Möller-Trumbore, world point, world distance, normal) is computed with
// the reference's / glm's exact operation order, so hits are bit-identical to the brute-force reference loop.  Only
// pruning (box tests, candidate screening) uses fused multiply-adds and approximate reciprocals, always with explicit
// margins: it can skip work that cannot matter, it never feeds a result.
#include "ff_kernels.h"

namespace ff {
namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kTriEpsilon = 0.000001f;  // kernel.cu:38
constexpr float kPlaneDenomMin = 1e-7f;   // kernel.cu:12 compares a float with the double 1e-7: (double)|d| > 1e-7 <=> |d| >= float(1e-7)
constexpr float kRayEps = 1.0e-4f;        // origin offset of bounce rays along the unit normal (build-defined)
constexpr int kWave = 64;

struct Ray {
    float ox, oy, oz, dx, dy, dz;
};

// Closest hit.  rec = TriRecord index for triangles, -1 for planes; (px,py,pz) = world-space hit point.
struct Best {
    float dist;
    int geom;
    int rec;
    float px, py, pz;
    float cx, cy, cz; // object-space normal as found: cross(e1, e2) (not normalised) for a triangle, m_normal for a plane
};

// What a query carries while it is in flight: the exact distance and identity of the best resolved candidate.  The hit
// point is produced once, at the end (finish_segment), to keep three registers out of the traversal loop.
struct BestId {
    float dist;
    int geom;
    int rec;
};

struct Counters {
    unsigned rays, nodes, tris, planes; // per lane and launch (flushed into 64-bit device counters)
    unsigned cut;      // WAVE-uniform: last-bounce queries that ended after the analytic records (no emitter among the candidates)
    unsigned reused;   // WAVE-uniform (a scalar register): of the wave's `rays`, the repeated primary rays answered from the block's cache
    // occupancy probes (instrumented launches only): wave-level rounds of each phase.  The active-lane totals of the
    // phases are the counters above (nodes = inner-step lanes, tris = triangle-test lanes, planes, rays).
    unsigned inner_rounds, leaf_rounds, tri_rounds, plane_rounds, segment_rounds;
    unsigned no_mesh; // queries that needed no mesh traversal (planes only)
    unsigned stack_overflow; // instrumented launches: pushes beyond the stack's depth (must stay 0: the depth is a bound)
    unsigned plane_exact; // plane tests that fell inside a screening margin and ran the exact reference test
    unsigned wall_rounds; // wave-level passes over the table of axis-aligned walls
    unsigned long long guard_hits; // ALL launches, wave-uniform (a scalar register pair): lanes whose query the traversal loop guard cut short
                                   // (must stay 0; the host turns it into an error)
    unsigned long long t_start, t_inner, t_leaf; // instrumented launches: wave cycles in mesh starts / inner phases / leaf phases
    unsigned long long t_b1, t_b2, t_b3;         // ... and in the three parts of begin_segment (quad boxes / quad screens / mesh boxes)
    unsigned long long t_l1, t_l2, t_l3;         // ... and of a leaf visit: waiting for the triangle records / the tests / the pop that follows
};

// Count one wave-level round of a phase: exactly one of the active lanes (the lowest) records it.
__device__ __forceinline__ void probe_round(unsigned& counter)
{
    const unsigned long long m = __ballot(true);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) counter += 1;
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    // glm dot(vec3): (x + y) + z  (GLM/detail/func_geometric.inl:52-53)
    const float px = ax * bx, py = ay * by, pz = az * bz;
    return (px + py) + pz;
}

// Correctly rounded 1/x and sqrt(x) (== the compiler's IEEE expansions, bit for bit, for every float: checked over all
// 2^32 inputs by tools/diag/ieee_check.hip and tests/test_gpu_properties.py).  In the range 2^-60 .. 2^60, where every
// operand of this renderer lives, one Newton step on the hardware estimate is already exact and replaces the 11 / 17
// instruction expansions with their denormal scaling; outside the range the full expansion runs.
__device__ __forceinline__ float ieee_rcp(float x)
{
    const unsigned a = __float_as_uint(x) & 0x7fffffffu;
    if (__builtin_expect(a - 0x21800000u < 0x3c000000u, 1)) { // 2^-60 <= |x| < 2^60
        const float y = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(e, y, y);
    }
    return 1.0f / x;
}
__device__ __forceinline__ float ieee_sqrt(float x)
{
    if (__builtin_expect(__float_as_uint(x) - 0x21800000u < 0x3c000000u, 1)) { // 2^-60 <= x < 2^60
        const float r = __builtin_amdgcn_rsqf(x);
        const float g = x * r, h = 0.5f * r;
        const float e = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(e, h, g);
    }
    return sqrtf(x);
}


// kernel.cu:138 — Ray(invM * vec4(o,1), normalize(invM * vec4(d,0))).  `len` is |invM*d| before normalisation: an
// object-space parameter t corresponds to the world distance t * |d_world| / len.
__device__ __forceinline__ void object_space_ray(const GeomRecord& G, const Ray& r, Ray& o, float& len)
{
    o.ox = (G.inv_c0[0] * r.ox + G.inv_c1[0] * r.oy) + (G.inv_c2[0] * r.oz + G.inv_c3[0]);
    o.oy = (G.inv_c0[1] * r.ox + G.inv_c1[1] * r.oy) + (G.inv_c2[1] * r.oz + G.inv_c3[1]);
    o.oz = (G.inv_c0[2] * r.ox + G.inv_c1[2] * r.oy) + (G.inv_c2[2] * r.oz + G.inv_c3[2]);
    const float tx = (G.inv_c0[0] * r.dx + G.inv_c1[0] * r.dy) + (G.inv_c2[0] * r.dz + G.inv_c0[3]);
    const float ty = (G.inv_c0[1] * r.dx + G.inv_c1[1] * r.dy) + (G.inv_c2[1] * r.dz + G.inv_c1[3]);
    const float tz = (G.inv_c0[2] * r.dx + G.inv_c1[2] * r.dy) + (G.inv_c2[2] * r.dz + G.inv_c2[3]);
    // normalize(vec4) with w == +-0: dot4 = (x*x + y*y) + (z*z + 0)
    const float dd = (tx * tx + ty * ty) + tz * tz;
    len = ieee_sqrt(dd);
    const float inv = ieee_rcp(len); // glm inversesqrt = 1 / sqrt
    o.dx = tx * inv;
    o.dy = ty * inv;
    o.dz = tz * inv;
}

// kernel.cu:110-125 on a candidate at object-space parameter t (brute-force kernels).  Ties on the world distance
// resolve like the reference's iteration order (lowest geometry index, then lowest triangle index).
__device__ __forceinline__ void consider(const GeomRecord& G, int g, int rec, int orig_tri, float t, const Ray& osr, const Ray& wr,
                                         const GeomRecord* __restrict__ geoms, const TriRecord* __restrict__ tris, Best& best)
{
    const float Px = osr.ox + osr.dx * t, Py = osr.oy + osr.dy * t, Pz = osr.oz + osr.dz * t; // kernel.cu:99 / :16
    const float wx = (G.mod_c0[0] * Px + G.mod_c1[0] * Py) + (G.mod_c2[0] * Pz + G.mod_c3[0]); // kernel.cu:113
    const float wy = (G.mod_c0[1] * Px + G.mod_c1[1] * Py) + (G.mod_c2[1] * Pz + G.mod_c3[1]);
    const float wz = (G.mod_c0[2] * Px + G.mod_c1[2] * Py) + (G.mod_c2[2] * Pz + G.mod_c3[2]);
    const float vx = wr.ox - wx, vy = wr.oy - wy, vz = wr.oz - wz;
    const float d2 = (vx * vx + vy * vy) + vz * vz;
    // sqrt is monotonic: a squared distance clearly above the best one cannot win or tie; skip the IEEE sqrt for it
    if (d2 > best.dist * best.dist * 1.00001f) return;
    const float dist = ieee_sqrt(d2); // glm distance, kernel.cu:114
    bool take = dist < best.dist; // kernel.cu:115
    if (!take && dist == best.dist && best.geom >= 0) {
        const int bo = geoms[best.geom].orig_index;
        if (G.orig_index < bo) take = true;
        else if (G.orig_index == bo && rec >= 0 && best.rec >= 0) take = orig_tri < tris[best.rec].orig_index;
    }
    if (take) {
        best.dist = dist;
        best.geom = g;
        best.rec = rec;
        best.px = wx;
        best.py = wy;
        best.pz = wz;
    }
}

// Object-space normal of a finished brute-force hit (the BVH path gets it from the exact evaluation of the winner).
__device__ __forceinline__ void fill_object_normal(const GeomRecord* __restrict__ geoms, const TriRecord* __restrict__ tris, Best& best)
{
    if (best.geom < 0) return;
    if (best.rec >= 0) {
        const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)best.rec * 3;
        const float4 b = tp[1], c = tp[2];
        const float e1x = b.x, e1y = b.y, e1z = b.z;
        const float e2x = c.x, e2y = c.y, e2z = c.z;
        best.cx = e1y * e2z - e2y * e1z;
        best.cy = e1z * e2x - e2z * e1x;
        best.cz = e1x * e2y - e2x * e1y;
    } else {
        const GeomRecord& G = geoms[best.geom];
        best.cx = G.plane_n[0];
        best.cy = G.plane_n[1];
        best.cz = G.plane_n[2];
    }
}


// kernel.cu:35-108 (Möller-Trumbore, division deferred, back faces culled).  Returns the object-space t or -1.
// A = (v0, original index), E1 = (v1 - v0, cull margin), E2 = (v2 - v0, -): the edges of :44-45 come with the record.
__device__ __forceinline__ float triangle_t(const float4 A, const float4 E1, const float4 E2, const Ray& r)
{
    const float e1x = E1.x, e1y = E1.y, e1z = E1.z; // :44
    const float e2x = E2.x, e2y = E2.y, e2z = E2.z; // :45
    const float nx = e1y * e2z - e2y * e1z, ny = e1z * e2x - e2z * e1x, nz = e1x * e2y - e2x * e1y; // :48 glm cross
    if (dot3(r.dx, r.dy, r.dz, nx, ny, nz) > 0.0f) return -1.0f;                                        // :49
    const float px = r.dy * e2z - e2y * r.dz, py = r.dz * e2x - e2z * r.dx, pz = r.dx * e2y - e2x * r.dy; // :53
    const float det = dot3(e1x, e1y, e1z, px, py, pz);                                                   // :54
    if (det < kTriEpsilon) return -1.0f;                                                                 // :57
    const float tx = r.ox - A.x, ty = r.oy - A.y, tz = r.oz - A.z;                                       // :61
    const float u = dot3(tx, ty, tz, px, py, pz);                                                        // :62
    if (u < 0.0f || u > det) return -1.0f;                                                               // :64
    const float qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;            // :68
    const float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz);                                                  // :70
    if (v < 0.0f || u + v > det) return -1.0f;                                                           // :71
    float t = dot3(e2x, e2y, e2z, qx, qy, qz);                                                           // :75
    const float invDet = ieee_rcp(det); // :77 (a double division narrowed to float == the float division)
    t = t * invDet;                  // :79
    return t > kTriEpsilon ? t : -1.0f; // :97
}

// The barycentrics of kernel.cu:62,70,80-81 (u = dot(tvec, pvec) * invDet, v = dot(d, qvec) * invDet) for a triangle the
// ray is known to hit, and the vertex normals interpolated with them: n = ((1 - u) - v) n0 + u n1 + v n2.  A triangle whose
// three vertex normals are zero (an OBJ without vn) keeps its geometric normal: returns false.
__device__ __forceinline__ bool smooth_normal(const float4 A, const float4 E1, const float4 E2, const float4* __restrict__ nrm, const Ray& r, float& nx,
                                              float& ny, float& nz)
{
    const float px = r.dy * E2.z - E2.y * r.dz, py = r.dz * E2.x - E2.z * r.dx, pz = r.dx * E2.y - E2.x * r.dy;
    const float det = dot3(E1.x, E1.y, E1.z, px, py, pz);
    const float tx = r.ox - A.x, ty = r.oy - A.y, tz = r.oz - A.z;
    float u = dot3(tx, ty, tz, px, py, pz);
    const float qx = ty * E1.z - E1.y * tz, qy = tz * E1.x - E1.z * tx, qz = tx * E1.y - E1.x * ty;
    float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz);
    const float invDet = ieee_rcp(det);
    u = u * invDet;
    v = v * invDet;
    const float4 n0 = nrm[0], n1 = nrm[1], n2 = nrm[2];
    const float w = (1.0f - u) - v;
    const float sx = (w * n0.x + u * n1.x) + v * n2.x;
    const float sy = (w * n0.y + u * n1.y) + v * n2.y;
    const float sz = (w * n0.z + u * n1.z) + v * n2.z;
    if (sx == 0.0f && sy == 0.0f && sz == 0.0f) return false;
    nx = sx;
    ny = sy;
    nz = sz;
    return true;
}

// kernel.cu:8-32 for an object-space ray and plane normal n.  Returns t or -1.
__device__ __forceinline__ float plane_t(float nx, float ny, float nz, const Ray& r)
{
    const float denom = dot3(nx, ny, nz, r.dx, r.dy, r.dz); // :11
    if (!(fabsf(denom) >= kPlaneDenomMin)) return -1.0f;    // :12
    const float t = dot3(-r.ox, -r.oy, -r.oz, nx, ny, nz) / denom; // :14-15
    const float Px = r.ox + t * r.dx, Py = r.oy + t * r.dy;        // :16
    if (!(Px >= -0.5f && Px <= 0.5f && Py >= -0.5f && Py <= 0.5f)) return -1.0f; // :18
    return t > 0.0f ? t : -1.0f;                                   // :23
}

// Sphere of radius `rad` about the object-space origin (build-defined: the reference only printf's at kernel.cu:166-169;
// oracle/ff_oracle.c orc_intersect_sphere is the definition).  Two-sided, nearest root above EPSILON.  Returns t or -1.
__device__ __forceinline__ float sphere_t(float rad, const Ray& r)
{
    const float b = dot3(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
    const float c = dot3(r.ox, r.oy, r.oz, r.ox, r.oy, r.oz) - rad * rad;
    const float disc = b * b - c;
    if (!(disc >= 0.0f)) return -1.0f;
    const float sq = ieee_sqrt(disc);
    float t = -b - sq;
    if (!(t > kTriEpsilon)) {
        t = -b + sq;
        if (!(t > kTriEpsilon)) return -1.0f;
    }
    return t;
}

// Unit object-space normal of a sphere hit at parameter t: P * (1 / rad).
__device__ __forceinline__ void sphere_normal(float rad, const Ray& r, float t, float& nx, float& ny, float& nz)
{
    const float inv = ieee_rcp(rad);
    nx = (r.ox + r.dx * t) * inv;
    ny = (r.oy + r.dy * t) * inv;
    nz = (r.oz + r.dz * t) * inv;
}

// The same for a sphere: the winning hit is evaluated once more (same arithmetic, same result) for its object-space point.
__device__ __forceinline__ void fill_sphere_normal(const GeomRecord* __restrict__ geoms, const Ray& wr, Best& best)
{
    if (best.geom < 0 || geoms[best.geom].type != FF_GEOM_SPHERE) return;
    const GeomRecord& G = geoms[best.geom];
    Ray osr;
    float len;
    object_space_ray(G, wr, osr, len);
    const float t = sphere_t(G.plane_n[3], osr);
    sphere_normal(G.plane_n[3], osr, t, best.cx, best.cy, best.cz);
}

// Brute-force path: the interpolated vertex normal of a finished triangle hit (FF_SHADE_DIFFUSE_PATH_SMOOTH).
__device__ __forceinline__ void fill_smooth_normal(const GeomRecord* __restrict__ geoms, const TriRecord* __restrict__ tris,
                                                   const float4* __restrict__ trinormals, const Ray& wr, Best& best)
{
    if (!trinormals || best.geom < 0 || best.rec < 0) return;
    Ray osr;
    float len;
    object_space_ray(geoms[best.geom], wr, osr, len);
    const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)best.rec * 3;
    smooth_normal(tp[0], tp[1], tp[2], trinormals + (size_t)best.rec * 3, osr, best.cx, best.cy, best.cz);
}

// ---- LDS layout of the BVH kernels -----------------------------------------------------------------------------------
//
//   [ nodes: 7 planes of node_cap x 16 B ][ traversal stacks: stack_depth x BLOCK x 4 B, lane-strided ][ geometry records: G x 288 B ]
//
// ff_smem is indexed directly (never through a generic pointer) so that every access compiles to ds_read/ds_write.
extern __shared__ uint4 ff_smem[];

constexpr int kGeomVec4 = (int)(sizeof(GeomRecord) / 16); // 18 float4 per geometry record
constexpr int kNodeVec4 = (int)(sizeof(Bvh4Node) / 16);   // 7 quarters per 4-wide node: six box planes + links
constexpr int kDone = 0x7fffffff;                         // traversal cursor of a lane with nothing left to visit
constexpr int kMeshDone = 0x7ffffffe;                     // big scenes: the current mesh is exhausted, the walk through the geometry tree resumes
constexpr int kGeomLeaf = 0x40000000;                     // big scenes: ~link of a geometry-tree leaf = kGeomLeaf | geometry record index
constexpr int kPackedEntry = 0x40000000;                  // stack entry that names a node and up to three of its slots (see inner_step)
constexpr unsigned kItemPixelMask = 0x1FFFFFFu;           // Path::item: the pixel item number (the host keeps pix_items below 2^25) ...
constexpr int kItemBlockShift = 25;                       // ... the sample block above it (at most 16 blocks per pixel) ...
constexpr unsigned kItemTail = 0x80000000u;               // ... and the sign bit for tail items

struct LdsBase {
    int node_cap;   // LDS node slots: quarter k of LDS node j lives at uint4 index k * node_cap + j
    int stack_base; // uint index of this lane's stack slot 0 (in units of 4 bytes from ff_smem)
    int stack_slot; // whose stack that is: the thread's own (its index in the workgroup), or, in the job-pool kernel, the job's slot
    int stack_depth; // entries per lane kept in LDS
    int* spill;      // deeper entries: entry e >= stack_depth of thread g of the launch at spill[(e - stack_depth) * threads + g] (null: none).
                     // Wave-uniform (scalar registers); the lane's own address is formed in the rare branch that needs it
    int block;       // workgroup size
    int stride;     // uints between consecutive stack entries of one lane (= block size)
    int geom_base;  // uint4 index of geometry record 0
    int num_quads;  // geometry records [0, num_quads) are planes; [num_quads, num_planes) spheres; meshes follow
    const float4* smooth_normals; // non-null: triangle hits carry the interpolated vertex normal (FF_SHADE_DIFFUSE_PATH_SMOOTH)
    // Scenes of more than 32 geometries ("big", a compile-time property of the kernel instantiation): the records stay in
    // global memory (L1/L2) and a query finds its candidates by walking a tree over the geometries' world boxes (tlas).
    // The geometry tree is a 4-wide tree like the meshes' (same nodes, same inner step, in WORLD space); its leaves are
    // geometries: link = ~(kGeomLeaf | record index).
    const float4* geoms_g;
    int top_first, top_lds_first, top_lds_count; // its first node in the node array and its share of the LDS node slots
    int num_scan; // ... and the records [0, num_scan) are planes that stay out of that tree (the walls of a room): every query screens them first
};
// (`big` is part of the TYPE, not a field: with a field the optimiser meets a select between an LDS and a global pointer in
// the record accessors before it has folded the flag, and this compiler crashes on it.)
// BIG: 0 = up to 32 geometries (records in LDS, every query screens them all); 1 = more, records still in LDS (up to
// kMaxLdsRecords); 2 = more than that, records read from global memory.
template <int BIG>
struct LdsT : LdsBase {
    static constexpr bool big = BIG != 0;
    static constexpr bool records_lds = BIG != 2;
};

template <int BIG = 0>
__device__ __forceinline__ LdsT<BIG> make_lds(int node_cap, int stack_depth, int block, int tid, int num_quads, const float4* smooth_normals = nullptr,
                                              const GeomRecord* geoms = nullptr, int top_first = 0, int top_lds_first = 0, int top_lds_count = 0,
                                              int num_scan = 0, int* spill = nullptr)
{
    LdsT<BIG> L;
    // Stack entries beyond the LDS levels live in global memory, lane-strided over the whole launch (the host trades the deepest,
    // rarely used stack levels for tree nodes in LDS: finalize_layout).  (Fetching the pointer from the kernel arguments only when
    // an entry spills, instead of keeping it in registers, was measured: no gain.)
    L.spill = spill;
    L.block = block;
    L.num_scan = num_scan;
    L.geoms_g = reinterpret_cast<const float4*>(geoms);
    L.top_first = top_first;
    L.top_lds_first = top_lds_first;
    L.top_lds_count = top_lds_count;
    L.num_quads = num_quads;
    L.smooth_normals = smooth_normals;
    L.node_cap = node_cap;
    L.stride = block;
    L.stack_base = node_cap * (kNodeVec4 * 4) + tid;
    L.stack_slot = tid;
    L.stack_depth = stack_depth;
    L.geom_base = node_cap * kNodeVec4 + (stack_depth * block) / 4;
    return L;
}

// Stage the top of every mesh's 4-wide tree and the geometry records: coalesced 16-byte loads, 1 KiB per wave-instruction.
// Persistent workgroups pay this once per launch, not per ray.  Which nodes of which mesh are cached was decided on the
// host (GeomRecord::lds_nodes nodes from the mesh's root on, at LDS node index lds_first: the trees are numbered level by
// level, so that is the top of each tree).
template <class LDS>
__device__ __forceinline__ void stage_scene(const LDS& L, const uint4* __restrict__ nodes4, const GeomRecord* __restrict__ geoms, int num_geoms,
                                            int num_planes, int tid, int block)
{
    // Nodes are stored as seven planes of 16-byte quarters: lanes fetch quarter k of unrelated nodes with one
    // ds_read_b128, and in this layout those addresses spread over all LDS banks, whereas whole nodes would put every
    // lane's quarter k on the same banks.
    for (int g = num_planes; g < num_geoms; ++g) {
        const int count = geoms[g].lds_nodes;
        if (count <= 0) continue;
        const int base = __float_as_int(geoms[g].wmin[3]);
        const uint4* src = nodes4 + (size_t)geoms[g].node4_first * kNodeVec4;
        for (int i = tid; i < count * kNodeVec4; i += block) {
            const int j = i / kNodeVec4, k = i - j * kNodeVec4;
            ff_smem[k * L.node_cap + base + j] = src[i];
        }
    }
    if constexpr (LDS::big) {
        const uint4* src = nodes4 + (size_t)L.top_first * kNodeVec4;
        for (int i = tid; i < L.top_lds_count * kNodeVec4; i += block) {
            const int j = i / kNodeVec4, k = i - j * kNodeVec4;
            ff_smem[k * L.node_cap + L.top_lds_first + j] = src[i];
        }
    }
    if constexpr (LDS::records_lds) {
        const uint4* gsrc = reinterpret_cast<const uint4*>(geoms);
        for (int i = tid; i < num_geoms * kGeomVec4; i += block) ff_smem[L.geom_base + i] = gsrc[i];
    }
    __syncthreads();
}

// Quarter k of geometry record g.
template <class LDS>
__device__ __forceinline__ float4 lds_geom4(const LDS& L, int g, int k)
{
    if constexpr (!LDS::records_lds) return L.geoms_g[(size_t)g * kGeomVec4 + k];
    return reinterpret_cast<const float4*>(ff_smem)[L.geom_base + g * kGeomVec4 + k];
}
template <class LDS>
__device__ __forceinline__ int4 lds_geom_i4(const LDS& L, int g, int k)
{
    if constexpr (!LDS::records_lds) return reinterpret_cast<const int4*>(L.geoms_g)[(size_t)g * kGeomVec4 + k];
    return reinterpret_cast<const int4*>(ff_smem)[L.geom_base + g * kGeomVec4 + k];
}
template <class LDS>
__device__ __forceinline__ void stack_push(const LDS& L, int sp, int v)
{
    if (__builtin_expect(sp < L.stack_depth, 1)) reinterpret_cast<int*>(ff_smem)[L.stack_base + sp * L.stride] = v;
    else L.spill[((size_t)(sp - L.stack_depth) * gridDim.x + blockIdx.x) * (size_t)L.block + (size_t)L.stack_slot] = v;
}
template <class LDS>
__device__ __forceinline__ int stack_pop(const LDS& L, int sp)
{
    // (each load pinned inside its branch: left alone the compiler merges the LDS and the global one into a single flat_load_dword
    // through a generic pointer, which takes the long way round for the LDS case and waits on both memory counters)
    int v;
    if (__builtin_expect(sp < L.stack_depth, 1)) {
        v = reinterpret_cast<const int*>(ff_smem)[L.stack_base + sp * L.stride];
        asm volatile("" : "+v"(v));
    } else {
        v = L.spill[((size_t)(sp - L.stack_depth) * gridDim.x + blockIdx.x) * (size_t)L.block + (size_t)L.stack_slot];
        asm volatile("" : "+v"(v));
    }
    return v;
}

// kernel.cu:138 with the geometry record gathered from LDS by a lane-varying index (same arithmetic as object_space_ray).
template <class LDS>
__device__ __forceinline__ void object_space_ray_lds(const LDS& L, int g, const Ray& r, Ray& o, float& len)
{
    const float4 c0 = lds_geom4(L, g, 0), c1 = lds_geom4(L, g, 1), c2 = lds_geom4(L, g, 2), c3 = lds_geom4(L, g, 3);
    o.ox = (c0.x * r.ox + c1.x * r.oy) + (c2.x * r.oz + c3.x);
    o.oy = (c0.y * r.ox + c1.y * r.oy) + (c2.y * r.oz + c3.y);
    o.oz = (c0.z * r.ox + c1.z * r.oy) + (c2.z * r.oz + c3.z);
    const float tx = (c0.x * r.dx + c1.x * r.dy) + (c2.x * r.dz + c0.w);
    const float ty = (c0.y * r.dx + c1.y * r.dy) + (c2.y * r.dz + c1.w);
    const float tz = (c0.z * r.dx + c1.z * r.dy) + (c2.z * r.dz + c2.w);
    const float dd = (tx * tx + ty * ty) + tz * tz;
    len = ieee_sqrt(dd);
    const float inv = ieee_rcp(len);
    o.dx = tx * inv;
    o.dy = ty * inv;
    o.dz = tz * inv;
}

// Per-ray constants for the conservative world-space AABB test of each geometry (pruning only).
struct WorldSlab {
    float ix, iy, iz, ox, oy, oz; // 1/d and -o/d
    float inv_len;                // 1 / |d|: converts a world distance into the ray parameter
};

__device__ __forceinline__ float safe_rcp(float d)
{
    const float s = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d;
    return __builtin_amdgcn_rcpf(s);
}

__device__ __forceinline__ WorldSlab make_world_slab(const Ray& wr)
{
    WorldSlab w;
    w.ix = safe_rcp(wr.dx);
    w.iy = safe_rcp(wr.dy);
    w.iz = safe_rcp(wr.dz);
    w.ox = -wr.ox * w.ix;
    w.oy = -wr.oy * w.iy;
    w.oz = -wr.oz * w.iz;
    w.inv_len = __builtin_amdgcn_rsqf(__builtin_fmaf(wr.dx, wr.dx, __builtin_fmaf(wr.dy, wr.dy, wr.dz * wr.dz)));
    return w;
}

// Can the ray reach a world box before world distance `limit`?  Conservative: approximate arithmetic, inflated bounds,
// padded boxes; a `false` only ever skips work that could not have produced the closest hit.
__device__ __forceinline__ bool slab_may_hit(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const WorldSlab& w, float limit)
{
    const float a0 = __builtin_fmaf(mnx, w.ix, w.ox), a1 = __builtin_fmaf(mxx, w.ix, w.ox);
    const float b0 = __builtin_fmaf(mny, w.iy, w.oy), b1 = __builtin_fmaf(mxy, w.iy, w.oy);
    const float c0 = __builtin_fmaf(mnz, w.iz, w.oz), c1 = __builtin_fmaf(mxz, w.iz, w.oz);
    const float bound = (limit * 1.001f + 1.0e-3f) * w.inv_len * 1.00001f;
    const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
    const float tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), bound));
    return tn <= tf * 1.000002f;
}

// ---- closest hit, BVH mode -------------------------------------------------------------------------------------------
//
// intersectRays (kernel.cu:127-176) reorganised for 64-wide waves.  The result is the reference's result bit for bit;
// what changes is WHEN the expensive exact arithmetic runs:
//
//   * Every lane first screens all geometries against their world boxes (wave-uniform loop, scalar loads) and keeps a
//     bit mask of candidates; it then works through ITS OWN candidates, so a lane never executes code for a geometry it
//     has culled while its neighbours test it.
//   * Hit tests run in a fast form: the exact reference arithmetic up to (not including) the IEEE division, an
//     approximate reciprocal to place the hit along the ray, and an explicit margin.  A test that is clearly a hit
//     becomes the lane's PENDING candidate when it is not clearly farther than what the lane already holds; a test
//     that is clearly a miss is dropped; anything within the margin is decided at once by the exact reference test.
//   * The exact world distance (kernel.cu:113-114: IEEE divide, model transform, IEEE sqrt) is computed only when a
//     pending candidate is resolved: once per ray in the common case, and immediately whenever two candidates are too
//     close to rank approximately.  Ranking therefore always happens on exact reference distances.

constexpr int kLoopGuard = 1 << 16;              // upper bound on wave-level traversal rounds per query
constexpr float kRel = 1.0e-4f, kAbs = 1.0e-4f; // screening margins, far above the rounding error of the fast forms

struct Pending {
    float dist; // approximate world distance, +inf when empty
    int geom;   // record index, -1 when empty
    int rec;    // TriRecord index, -1 for a plane
};

// Per-lane state of one closest-hit query in flight.
struct Segment {
    BestId best;
    Pending pend;
    unsigned meshes;           // candidate meshes not started yet (bit = record index; scenes of up to 32 geometries)
    int cur, sp, mesh;         // traversal cursor (4-wide node relative to the mesh's root >= 0, leaf < 0, kDone), stack height, record index of the current mesh
    int tl_sp;                 // big scenes, while a mesh is being traversed: stack entries [0, tl_sp) are the pending entries of the
                               // geometry tree, the mesh's own entries sit above them (0 otherwise)
    int node_base;             // the current mesh's first node in the global 4-wide node array
    int lds_first, lds_count;  // its nodes [0, lds_count) sit in LDS from LDS node index lds_first on
    int bnx, bny, bnz;         // box planes (quarters of a node) the ray enters through, as byte offsets into the LDS node image (set_box_planes)
    int bfx, bfy, bfz;         // ... and leaves through
    Ray osr;                   // object-space ray of the current mesh
    float ix, iy, iz, ox, oy, oz; // 1/d and -o/d of osr (box tests)
    float scale;               // object-space t per unit of world distance
    float tbound;              // object-space ray parameter beyond which nothing can beat what the lane holds (box pruning)
    int resume;                // > 0: triangle resume-1 of the leaf under the cursor met a near tie with the pending candidate;
                               //      the caller resolves the pending one exactly, then the leaf continues from that triangle
};

__device__ __forceinline__ float inv_length(const Ray& r)
{
    return __builtin_amdgcn_rsqf(__builtin_fmaf(r.dx, r.dx, __builtin_fmaf(r.dy, r.dy, r.dz * r.dz)));
}

// Exact reference evaluation (kernel.cu:35-125) of candidate (g, rec) for world ray wr: world distance and hit point.
// Returns false if the exact test rejects it (cannot happen for a screened candidate; kept so that a wrong margin
// could never corrupt a result).
struct HitPoint {
    float wx, wy, wz; // world-space point
    float cx, cy, cz; // object-space normal as found (see Best)
};

template <class LDS>
__device__ __forceinline__ bool exact_hit(const LDS& L, const TriRecord* __restrict__ tris, const Ray& wr, int g, int rec, float& dist, HitPoint& H,
                                          int& orig_tri)
{
    Ray osr;
    float len;
    object_space_ray_lds(L, g, wr, osr, len);
    float t;
    orig_tri = -1;
    if (rec >= 0) {
        const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)rec * 3;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        orig_tri = __float_as_int(a.w);
        t = triangle_t(a, b, c, osr);
        const float e1x = b.x, e1y = b.y, e1z = b.z;
        const float e2x = c.x, e2y = c.y, e2z = c.z;
        H.cx = e1y * e2z - e2y * e1z; // kernel.cu:101 cross(edge1, edge2), shading normalises it where the reference does
        H.cy = e1z * e2x - e2z * e1x;
        H.cz = e1x * e2y - e2x * e1y;
        if (L.smooth_normals) smooth_normal(a, b, c, L.smooth_normals + (size_t)rec * 3, osr, H.cx, H.cy, H.cz);
    } else {
        const float4 pn = lds_geom4(L, g, 11);
        if (g >= L.num_quads) {
            t = sphere_t(pn.w, osr);
            sphere_normal(pn.w, osr, t, H.cx, H.cy, H.cz);
        } else {
            t = plane_t(pn.x, pn.y, pn.z, osr);
            H.cx = pn.x; // kernel.cu:26
            H.cy = pn.y;
            H.cz = pn.z;
        }
    }
    if (!(t > 0.0f)) return false;
    const float4 m0 = lds_geom4(L, g, 4), m1 = lds_geom4(L, g, 5), m2 = lds_geom4(L, g, 6), m3 = lds_geom4(L, g, 7);
    const float Px = osr.ox + osr.dx * t, Py = osr.oy + osr.dy * t, Pz = osr.oz + osr.dz * t; // kernel.cu:99 / :16
    H.wx = (m0.x * Px + m1.x * Py) + (m2.x * Pz + m3.x);                                      // kernel.cu:113
    H.wy = (m0.y * Px + m1.y * Py) + (m2.y * Pz + m3.y);
    H.wz = (m0.z * Px + m1.z * Py) + (m2.z * Pz + m3.z);
    const float vx = wr.ox - H.wx, vy = wr.oy - H.wy, vz = wr.oz - H.wz;
    dist = ieee_sqrt((vx * vx + vy * vy) + vz * vz); // kernel.cu:114
    return true;
}

// Resolve the pending candidate exactly and merge it into `best` (kernel.cu:115-121).  Returns true if it became the
// best; then H is its hit point and normal.
template <class LDS>
__device__ __forceinline__ bool resolve_pending(const LDS& L, const TriRecord* __restrict__ tris, const Ray& wr, Pending& pend, BestId& best,
                                                HitPoint& H)
{
    const int g = pend.geom, rec = pend.rec;
    pend.geom = -1;
    pend.dist = kInf;
    float dist;
    int orig_tri;
    if (!exact_hit(L, tris, wr, g, rec, dist, H, orig_tri)) return false;
    bool take = dist < best.dist; // kernel.cu:115
    if (!take && dist == best.dist && best.geom >= 0) {
        // the reference keeps the first hit in (geometry, triangle) iteration order among equal distances
        const int go = lds_geom_i4(L, g, 17).y, bo = lds_geom_i4(L, best.geom, 17).y;
        if (go < bo) take = true;
        else if (go == bo && rec >= 0 && best.rec >= 0) take = orig_tri < tris[best.rec].orig_index;
    }
    if (take) {
        best.dist = dist;
        best.geom = g;
        best.rec = rec;
    }
    return take;
}

// Offer a certain hit at approximate world distance d to the lane's pending slot.  Returns true when the slot holds a
// candidate that is too close to rank approximately: the caller must resolve the held one exactly (resolve_pending) and
// offer this one again.  The exact code is kept OUT of the hot loops on purpose: it runs at wave-loop level, where the
// loops' temporaries are dead, which keeps the kernel within the register budget of 4 waves per SIMD.
__device__ __forceinline__ bool offer(float d, int g, int rec, Pending& pend, const BestId& best)
{
    const float lim = fminf(best.dist, pend.dist);
    if (d > lim * (1.0f + kRel) + kAbs) return false;                             // clearly farther than something already held
    if (pend.geom >= 0 && !(pend.dist > d * (1.0f + kRel) + kAbs)) return true;   // near tie with the held candidate
    pend.dist = d;
    pend.geom = g;
    pend.rec = rec;
    return false;
}

// One plane or sphere against the lane's query.  Planes are screened WITHOUT the IEEE sqrt/divide of kernel.cu:138: the hit
// position on the unit quad does not depend on the length of the object-space direction, so the screen works on the
// un-normalised direction M^-1*d, for which the ray parameter is the world-space parameter.  Anything within the margins
// (quad edges, t ~ 0, |n.d| ~ 1e-7) is decided by the exact reference test at once.
template <bool STATS, class LDS>
__device__ __forceinline__ void screen_analytic(const LDS& L, int g, const TriRecord* __restrict__ tris, const Ray& wr, float wlen, Segment& S, Counters& cnt)
{
    if (STATS) { cnt.planes += 1; probe_round(cnt.plane_rounds); }
    if (g >= L.num_quads) {
        // spheres have no screening form: the exact test runs here and yields the approximate world distance
        Ray osr;
        float len;
        object_space_ray_lds(L, g, wr, osr, len);
        const float tt = sphere_t(lds_geom4(L, g, 11).w, osr);
        const float sdist = tt * wlen * __builtin_amdgcn_rcpf(len);
        if (tt > 0.0f && offer(sdist, g, -1, S.pend, S.best)) {
            HitPoint H;
            resolve_pending(L, tris, wr, S.pend, S.best, H);
            offer(sdist, g, -1, S.pend, S.best);
        }
        return;
    }
    const float4 c0 = lds_geom4(L, g, 0), c1 = lds_geom4(L, g, 1), c2 = lds_geom4(L, g, 2), c3 = lds_geom4(L, g, 3);
    const float4 pn = lds_geom4(L, g, 11);
    // object-space origin and un-normalised direction (screening only: FMA form)
    const float ox = __builtin_fmaf(c0.x, wr.ox, __builtin_fmaf(c1.x, wr.oy, __builtin_fmaf(c2.x, wr.oz, c3.x)));
    const float oy = __builtin_fmaf(c0.y, wr.ox, __builtin_fmaf(c1.y, wr.oy, __builtin_fmaf(c2.y, wr.oz, c3.y)));
    const float oz = __builtin_fmaf(c0.z, wr.ox, __builtin_fmaf(c1.z, wr.oy, __builtin_fmaf(c2.z, wr.oz, c3.z)));
    const float ux = __builtin_fmaf(c0.x, wr.dx, __builtin_fmaf(c1.x, wr.dy, c2.x * wr.dz));
    const float uy = __builtin_fmaf(c0.y, wr.dx, __builtin_fmaf(c1.y, wr.dy, c2.y * wr.dz));
    const float uz = __builtin_fmaf(c0.z, wr.dx, __builtin_fmaf(c1.z, wr.dy, c2.z * wr.dz));
    const float nx = pn.x, ny = pn.y, nz = pn.z;
    const float dn = __builtin_fmaf(nx, ux, __builtin_fmaf(ny, uy, nz * uz));         // n . (M^-1 d)
    const float num = -__builtin_fmaf(nx, ox, __builtin_fmaf(ny, oy, nz * oz));       // -(n . o')
    const float len2 = __builtin_fmaf(ux, ux, __builtin_fmaf(uy, uy, uz * uz));
    // kernel.cu:12 |n.d'| >= 1e-7 with d' = u/len  <=>  dn^2 >= 1e-14 * len2
    const float q = dn * dn, qlim = 1.0e-14f * len2;
    const float ta = num * __builtin_amdgcn_rcpf(dn);                                 // world ray parameter of the plane
    const float Pxa = __builtin_fmaf(ta, ux, ox), Pya = __builtin_fmaf(ta, uy, oy);
    const float omag = fabsf(ox) + fabsf(oy) + fabsf(oz);
    // (an error of the parameter moves the point by that times u / dn: the margin grows with the ray's obliquity to the plane)
    const float delta = 1.0e-5f * (1.0f + omag) * __builtin_fmaf(fabsf(ux) + fabsf(uy), fabsf(__builtin_amdgcn_rcpf(dn)), 1.0f);
    const float ex = fabsf(Pxa), ey = fabsf(Pya);
    const bool front_sure = ta > 0.0f && fabsf(num) > 1.0e-5f * omag * (fabsf(nx) + fabsf(ny) + fabsf(nz));
    bool hit = ex <= 0.5f - delta && ey <= 0.5f - delta && front_sure && q >= qlim * 1.01f;
    float dist = ta * wlen; // approximate world distance
    if (!hit && ex <= 0.5f + delta && ey <= 0.5f + delta && q >= qlim * 0.99f && (front_sure || fabsf(num) <= 1.0e-5f * omag * (fabsf(nx) + fabsf(ny) + fabsf(nz)))) {
        // within a margin: decide with the exact reference test (kernel.cu:138 + :8-32)
        if (STATS) cnt.plane_exact += 1;
        Ray osr;
        float len;
        object_space_ray_lds(L, g, wr, osr, len);
        const float tt = plane_t(nx, ny, nz, osr);
        hit = tt > 0.0f;
        dist = tt * wlen * __builtin_amdgcn_rcpf(len);
    }
    if (hit && offer(dist, g, -1, S.pend, S.best)) {
        // two planes too close to rank approximately (a ray into an edge of the box): settle the held one exactly
        HitPoint H;
        resolve_pending(L, tris, wr, S.pend, S.best, H);
        offer(dist, g, -1, S.pend, S.best);
    }
}

// ---- axis-aligned walls (WallTable) ---------------------------------------------------------------------------------------
//
// One wall normal to world axis k against the calling lanes' rays, in world space: t = (c - o_k) / d_k through the slab
// constants of the ray, the hit point's other two coordinates against the rectangle.  (u, v) are the two other axes in the
// table's order.  Three outcomes per lane: a certain hit (the candidate of the lane if it is clearly the nearest so far), a
// certain miss, or `slow` gets the wall's bit: the per-lane screen of the plane's record decides, with the exact reference test
// where it is close (kernel.cu:8-32).  Certain means: by more than `dl` in the rectangle's plane - a multiple of the rounding
// error of BOTH this form and the reference's object-space arithmetic, which grows with the ray's obliquity to the wall (an
// error of the parameter moves the point by that times d_u / d_k) - and by more than `tt` in the parameter's sign.  NaNs (an
// origin beyond 1e8) compare false everywhere and land in `slow`.
__device__ __forceinline__ void wall_test(const Wall& w, float ixk, float oxk, float ou, float du, float ov, float dv, float dl, float tt, bool steep,
                                          float wlen, float& best_d, int& best_g, bool& tie, unsigned& slow)
{
    const float t = __builtin_fmaf(w.c, ixk, oxk);
    const float pu = __builtin_fmaf(t, du, ou), pv = __builtin_fmaf(t, dv, ov);
    const float m = fmaxf(fabsf(pu - w.cu) - w.hu, fabsf(pv - w.cv) - w.hv); // > 0: outside the rectangle by that much
    const bool hit = m <= -dl && t > tt && steep;
    const bool miss = m > dl || t < -tt;
    // Straight-line selects throughout.  (The two rare cases - a lane inside a margin, a second certain hit that is not clearly
    // nearer - behind wave-uniform branches instead: C2 -3 %, the default camera -4 %.  A branch costs this loop more than the
    // five vector instructions it skips.)
    slow |= (!hit && !miss) ? 1u << w.geom : 0u;
    // offer(): clearly farther than the lane's candidate -> dropped; clearly nearer -> the new candidate; else a near tie
    const float d = t * wlen;
    const bool nearer = hit && best_d > __builtin_fmaf(d, 1.0f + kRel, kAbs);
    const bool farther = d > __builtin_fmaf(best_d, 1.0f + kRel, kAbs);
    tie = tie || (hit && !nearer && !farther);
    best_d = nearer ? d : best_d;
    best_g = nearer ? w.geom : best_g;
}

// An entry that holds two walls with one rectangle, at w.c < w.hi_c (floor and ceiling, left and right wall of a box).  A ray
// that starts between them can reach only the one its direction points at: the rectangle is tested once, at that wall's parameter;
// the other wall is a certain miss when its own parameter is certainly negative (the same criterion as above) and goes to the
// per-lane screen otherwise (an origin outside the pair, or on the wall itself).
__device__ __forceinline__ void wall_test_pair(const Wall& w, float ixk, float oxk, float ou, float du, float ov, float dv, float dl, float tt, bool steep,
                                               float wlen, float& best_d, int& best_g, bool& tie, unsigned& slow)
{
    const float tlo = __builtin_fmaf(w.c, ixk, oxk), thi = __builtin_fmaf(w.hi_c, ixk, oxk);
    const bool up = ixk > 0.0f;
    const float t = up ? thi : tlo, tother = up ? tlo : thi;
    const int g = up ? w.hi_geom1 - 1 : w.geom, gother = up ? w.geom : w.hi_geom1 - 1;
    slow |= !(tother < -tt) ? 1u << gother : 0u;
    const float pu = __builtin_fmaf(t, du, ou), pv = __builtin_fmaf(t, dv, ov);
    const float m = fmaxf(fabsf(pu - w.cu) - w.hu, fabsf(pv - w.cv) - w.hv);
    const bool hit = m <= -dl && t > tt && steep;
    const bool miss = m > dl || t < -tt;
    slow |= (!hit && !miss) ? 1u << g : 0u;
    const float d = t * wlen;
    const bool nearer = hit && best_d > __builtin_fmaf(d, 1.0f + kRel, kAbs);
    const bool farther = d > __builtin_fmaf(best_d, 1.0f + kRel, kAbs);
    tie = tie || (hit && !nearer && !farther);
    best_d = nearer ? d : best_d;
    best_g = nearer ? g : best_g;
}

// All walls of the table against the calling lanes' rays (wave-uniform loops; the table comes through scalar loads).  Must run
// on a query that holds nothing yet (begin_segment).  A lane that met a near tie between two walls gives all of them to the
// per-lane screens, which rank on exact distances.
template <bool STATS>
__device__ __forceinline__ void screen_walls(const WallTable& W, const Ray& wr, const WorldSlab& ws, float wlen, Segment& S, unsigned& slow, Counters& cnt)
{
    const int nx = W.count[0], ny = nx + W.count[1], nz = ny + W.count[2];
    if (nz == 0) return;
    if (STATS) { cnt.planes += (unsigned)nz; probe_round(cnt.wall_rounds); }
    const float D = 2.0e-5f * ((fabsf(wr.ox) + fabsf(wr.oy)) + (fabsf(wr.oz) + W.margin_s));
    const float th = 0.05f * D;
    const float ax = fabsf(wr.dx), ay = fabsf(wr.dy), az = fabsf(wr.dz);
    const float aix = fabsf(ws.ix), aiy = fabsf(ws.iy), aiz = fabsf(ws.iz);
    const float gmin = W.graze * wlen;
    float best_d = kInf;
    int best_g = -1;
    bool tie = false;
    // (the record of the next wall is requested before the current one is tested: a scalar load per iteration would otherwise
    // sit in front of every test)
    Wall cur = W.w[0];
    int i = 0;
    {
        const float dl = D * __builtin_fmaf(fmaxf(ay, az), aix, 1.0f), tt = th * aix;
        const bool steep = ax >= gmin;
        for (; i < nx; ++i) {
            const Wall nxt = W.w[min(i + 1, kMaxWalls - 1)];
            if (cur.hi_geom1) wall_test_pair(cur, ws.ix, ws.ox, wr.oy, wr.dy, wr.oz, wr.dz, dl, tt, steep, wlen, best_d, best_g, tie, slow);
            else wall_test(cur, ws.ix, ws.ox, wr.oy, wr.dy, wr.oz, wr.dz, dl, tt, steep, wlen, best_d, best_g, tie, slow);
            cur = nxt;
        }
    }
    {
        const float dl = D * __builtin_fmaf(fmaxf(az, ax), aiy, 1.0f), tt = th * aiy;
        const bool steep = ay >= gmin;
        for (; i < ny; ++i) {
            const Wall nxt = W.w[min(i + 1, kMaxWalls - 1)];
            if (cur.hi_geom1) wall_test_pair(cur, ws.iy, ws.oy, wr.oz, wr.dz, wr.ox, wr.dx, dl, tt, steep, wlen, best_d, best_g, tie, slow);
            else wall_test(cur, ws.iy, ws.oy, wr.oz, wr.dz, wr.ox, wr.dx, dl, tt, steep, wlen, best_d, best_g, tie, slow);
            cur = nxt;
        }
    }
    {
        const float dl = D * __builtin_fmaf(fmaxf(ax, ay), aiz, 1.0f), tt = th * aiz;
        const bool steep = az >= gmin;
        for (; i < nz; ++i) {
            const Wall nxt = W.w[min(i + 1, kMaxWalls - 1)];
            if (cur.hi_geom1) wall_test_pair(cur, ws.iz, ws.oz, wr.ox, wr.dx, wr.oy, wr.dy, dl, tt, steep, wlen, best_d, best_g, tie, slow);
            else wall_test(cur, ws.iz, ws.oz, wr.ox, wr.dx, wr.oy, wr.dy, dl, tt, steep, wlen, best_d, best_g, tie, slow);
            cur = nxt;
        }
    }
    if (tie) {
        slow |= W.mask;
    } else if (best_g >= 0) {
        S.pend.dist = best_d;
        S.pend.geom = best_g;
        S.pend.rec = -1;
    }
}

// Start a closest-hit query: test every plane (fast form) and remember which meshes the ray can reach.
//
// Planes are pre-filtered by their world boxes in a wave-uniform loop, then screened per lane WITHOUT the IEEE sqrt/divide
// of kernel.cu:138: the hit position on the unit quad does not depend on the length of the object-space direction, so
// the screen works on the un-normalised direction M^-1*d, for which the ray parameter is the world-space parameter.
// Anything within the margins (quad edges, t ~ 0, |n.d| ~ 1e-7) is decided by the exact reference test at once.
// Scenes of up to 32 geometries (the reference has 5): every query screens all planes / spheres and collects its candidate meshes
// in a bit mask (larger scenes walk the geometry tree instead: enter_top / geom_step).
// Bit mask of the records in the query's candidate slots (records 0..31: the analytic records screened before anything else).
__device__ __forceinline__ unsigned holds(const Segment& S)
{
    return (S.pend.geom >= 0 ? 1u << (S.pend.geom & 31) : 0u) | (S.best.geom >= 0 ? 1u << (S.best.geom & 31) : 0u);
}

template <bool STATS, class LDS>
__device__ __forceinline__ bool scan_records(const LDS& L, const WallTable& W, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                           const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Counters& cnt, bool cut, unsigned emitters)
{
    const int prim_end = num_planes;
    const float wlen = __builtin_amdgcn_rcpf(inv_length(wr)); // |world direction| (1 for the integrator's rays)

    // Stage 1, wave-uniform: which quads can the ray reach at all?  The padded world box of a quad is flat, so for the
    // axis-aligned walls of a box scene this conservative slab test already singles out the one wall the ray hits.
    unsigned long long tb0 = 0, tb1 = 0, tb2 = 0;
    if (STATS) tb0 = __builtin_amdgcn_s_memtime();
    const WorldSlab ws = make_world_slab(wr);
    unsigned quads = 0u;
    // Stage 0, wave-uniform: the axis-aligned walls in world space (one multiply-add and two range checks each; the walls of a box
    // scene never reach the per-lane screens below except on their edges)
    screen_walls<STATS>(W, wr, ws, wlen, S, quads, cnt);
    for (int g = 0; g < prim_end; ++g) {
        if ((W.mask >> g) & 1u) continue;
        const float4 bmin = lds_geom4(L, g, 14), bmax = lds_geom4(L, g, 15);
        if (slab_may_hit(bmin.x, bmin.y, bmin.z, bmax.x, bmax.y, bmax.z, ws, kInf)) quads |= 1u << g;
    }
    if (STATS) tb1 = __builtin_amdgcn_s_memtime();
    // Stage 2, per lane: screen the lane's own candidates (records from the LDS copy at per-lane addresses).
    for (int guard = 0; __ballot(quads != 0u) != 0ull && guard < 32; ++guard) {
        if (quads == 0u) continue;
        const int g = __ffs((int)quads) - 1;
        quads &= quads - 1u;
        screen_analytic<STATS>(L, g, tris, wr, wlen, S, cnt);
    }

    if (STATS) tb2 = __builtin_amdgcn_s_memtime();
    // meshes: conservative world-box test against what the planes already found
    S.meshes = 0u;
    const float limit = fminf(S.best.dist, S.pend.dist);
    {
        // (the boxes of the meshes that have a tree come with the table: scalar loads, the next one requested before the test)
        // (one 32-byte scalar load per box: read field by field the compiler issues seven loads and as many address computations)
        typedef unsigned box_words __attribute__((ext_vector_type(8)));
        static_assert(sizeof(WallTable::MeshBox) == 32, "one box, one load");
        box_words cur = *reinterpret_cast<const box_words*>(&W.box[0]);
        for (int i = 0; i < W.num_boxes; ++i) {
            const box_words nxt = *reinterpret_cast<const box_words*>(&W.box[min(i + 1, 31)]);
            S.meshes |= slab_may_hit(__uint_as_float(cur.s0), __uint_as_float(cur.s1), __uint_as_float(cur.s2), __uint_as_float(cur.s4), __uint_as_float(cur.s5),
                                     __uint_as_float(cur.s6), ws, limit) ? 1u << cur.s3 : 0u;
            cur = nxt;
        }
    }
    // A path's last segment adds radiance only if it ends on an emitter.  Every analytic record has been screened: the nearest
    // of them is one of the (at most two) candidates held.  If neither is an emitter, the closest hit of the whole query is a
    // non-emitter or nothing, whatever the meshes hold: the query ends here.  (Returned, and counted by the caller at wave level.)
    const bool over = cut && (holds(S) & emitters) == 0u;
    if (over) S.meshes = 0u;
    if (STATS && S.meshes == 0u) cnt.no_mesh += 1;
    if (STATS) {
        const unsigned long long tb3 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { cnt.t_b1 += tb1 - tb0; cnt.t_b2 += tb2 - tb1; cnt.t_b3 += tb3 - tb2; }
    }
    return over;
}

template <class LDS>
__device__ __forceinline__ void enter_top(const LDS& L, const Ray& wr, Segment& S);

// Big scenes: the planes kept out of the geometry tree (records [0, num_scan); host: count_scan_planes), screened like the planes
// of a small scene - world-box pre-filter in a wave-uniform loop, then every lane screens its own candidates - before the walk
// through the tree starts: the wall the ray ends on bounds that walk from its first node.
template <bool STATS, class LDS>
__device__ __forceinline__ void scan_walls(const LDS& L, const WallTable& W, const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Counters& cnt)
{
    const float wlen = __builtin_amdgcn_rcpf(inv_length(wr));
    const WorldSlab ws = make_world_slab(wr);
    unsigned prims = 0u;
    screen_walls<STATS>(W, wr, ws, wlen, S, prims, cnt);
    for (int g = 0; g < L.num_scan; ++g) {
        if ((W.mask >> g) & 1u) continue;
        const float4 bmin = lds_geom4(L, g, 14), bmax = lds_geom4(L, g, 15);
        if (slab_may_hit(bmin.x, bmin.y, bmin.z, bmax.x, bmax.y, bmax.z, ws, kInf)) prims |= 1u << g;
    }
    for (int guard = 0; __ballot(prims != 0u) != 0ull && guard < 8; ++guard) {
        if (prims == 0u) continue;
        const int g = __ffs((int)prims) - 1;
        prims &= prims - 1u;
        screen_analytic<STATS>(L, g, tris, wr, wlen, S, cnt);
    }
}

// Start a closest-hit query: empty candidate slots, then the geometry records (small scenes) or the root of the geometry tree.
// Returns true for a last-bounce query (`cut`) that is already over (scan_records).
template <bool STATS, class LDS>
__device__ __forceinline__ bool begin_segment(const LDS& L, const WallTable& W, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                              const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Counters& cnt, bool cut = false,
                                              unsigned emitters = 0u)
{
    S.best.dist = kInf; // kernel.cu:131
    S.best.geom = -1;
    S.best.rec = -1;
    S.pend.dist = kInf;
    S.pend.geom = -1;
    S.pend.rec = -1;
    S.cur = kDone;
    S.sp = 0;
    S.tl_sp = 0;
    S.mesh = -1;
    S.resume = 0;
    if constexpr (LDS::big) {
        // big scenes: the query starts at the root of the tree over the geometries, in world space
        S.meshes = 0u;
        if (L.num_scan > 0) scan_walls<STATS>(L, W, tris, wr, S, cnt);
        const bool over = cut && (holds(S) & emitters) == 0u; // (see scan_records: here every emitter is among the scanned planes)
        if (over) return true;
        enter_top(L, wr, S);
        S.cur = 0;
        return false;
    }
    return scan_records<STATS>(L, W, geoms, num_geoms, num_planes, tris, wr, S, cnt, cut, emitters);
}

// Box-pruning bound of the current mesh: refreshed whenever the lane's best/pending distance or its mesh changes, so the
// inner-node step reads one register instead of recomputing it per node.
__device__ __forceinline__ void refresh_tbound(Segment& S)
{
    S.tbound = (fminf(S.best.dist, S.pend.dist) * 1.001f + 1.0e-3f) * S.scale * 1.00001f;
}

// The links of a node are needed only after its box tests.  Left alone, the compiler merges the LDS load and the global load
// of the two branches into one load through a generic pointer placed after the tests: four flat_load_dword.  Pinning the
// loaded value inside each branch keeps them ds_read_b128 / global_load_dwordx4.
#define FF_PIN4(q) asm volatile("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w))

// q[c] for a lane-varying c in 0..3 without control flow (the compiler turns a ?: chain on c into nested branches): two
// sign-extended bit fields as masks and three bit-field inserts.
__device__ __forceinline__ int select_slot(const uint4 q, int c)
{
    const unsigned m0 = (unsigned)((c << 31) >> 31), m1 = (unsigned)((c << 30) >> 31); // all ones where bit 0 / bit 1 of c is set
    const unsigned lo = (q.y & m0) | (q.x & ~m0), hi = (q.w & m0) | (q.z & ~m0);
    return (int)((hi & m1) | (lo & ~m1));
}

// Take the next subtree off the lane's stack.  An entry is a link (the common case: one sibling was pending) or names a
// node and two or three of its slots, nearest first (kPackedEntry | node << 8 | slots << 2 | count): then the node's link
// quarter is read again, the nearest slot becomes the cursor and the entry is rewritten for the rest.  One entry per
// visited node bounds the stack by the depth of the tree.
template <class LDS>
__device__ __forceinline__ void pop_entry(const LDS& L, const uint4* __restrict__ nodes4, Segment& S, int e)
{
    // (`e` is the entry on top of the lane's stack, already read; the caller has checked that the stack is not empty)
    if (e >= 0 && (e & kPackedEntry) != 0) {
        const int node = (e >> 8) & 0x3FFFFF;
        uint4 lk;
        if ((unsigned)node < (unsigned)S.lds_count) {
            lk = ff_smem[S.lds_first + node + 6 * L.node_cap];
            FF_PIN4(lk);
        } else {
            lk = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(nodes4) + ((unsigned)(S.node_base + node) * (unsigned)(kNodeVec4 * 16) + 96u));
            FF_PIN4(lk);
        }
        S.cur = select_slot(lk, (e >> 2) & 3);
        const int rest = (e & 3) == 2 ? select_slot(lk, (e >> 4) & 3)                // one slot left: its link
                                      : ((e & ~0xFF) | (((e >> 4) & 0xF) << 2) | 2); // two left
        stack_push(L, S.sp - 1, rest);
    } else {
        S.cur = e;
        --S.sp;
    }
}

template <class LDS>
__device__ __forceinline__ void pop_subtree(const LDS& L, const uint4* __restrict__ nodes4, Segment& S)
{
    if (S.sp == (LDS::big ? S.tl_sp : 0)) {
        // nothing of the current tree is left; under a mesh of a big scene wait the pending entries of the geometry tree
        S.cur = LDS::big && S.mesh >= 0 ? kMeshDone : kDone;
        return;
    }
    pop_entry(L, nodes4, S, stack_pop(L, S.sp - 1));
}

// The quarters of a node the ray enters through (min planes 0/1/2 or max planes 3/4/5 by the signs of its direction) and leaves
// through, as BYTE offsets into the LDS node image (quarter k of node j: (k * node_cap + j) * 16): the inner step forms each of its
// six addresses with one add.  (Nodes outside LDS: inner_step derives the quarters from the same signs.)
template <class LDS>
__device__ __forceinline__ void set_box_planes(const LDS& L, Segment& S)
{
    const int q = L.node_cap * 16;
    S.bnx = S.ix < 0.0f ? 3 * q : 0;
    S.bny = S.iy < 0.0f ? 4 * q : q;
    S.bnz = S.iz < 0.0f ? 5 * q : 2 * q;
    S.bfx = 3 * q - S.bnx;
    S.bfy = 5 * q - S.bny;
    S.bfz = 7 * q - S.bnz;
}

// Put the lane's cursor on the root of mesh g's tree: object-space ray (kernel.cu:138), slab constants, box planes by the
// signs of the direction.
template <class LDS>
__device__ __forceinline__ void enter_mesh(const LDS& L, int g, const Ray& wr, Segment& S)
{
    const int4 tree = lds_geom_i4(L, g, 17); // bvh_root, orig_index, node4_first, lds_nodes
    if (tree.x < 0) return;
    float len;
    object_space_ray_lds(L, g, wr, S.osr, len);
    S.ix = safe_rcp(S.osr.dx);
    S.iy = safe_rcp(S.osr.dy);
    S.iz = safe_rcp(S.osr.dz);
    S.ox = -S.osr.ox * S.ix;
    S.oy = -S.osr.oy * S.iy;
    S.oz = -S.osr.oz * S.iz;
    set_box_planes(L, S);
    S.scale = len * inv_length(wr); // object-space t per unit of world distance
    refresh_tbound(S);
    S.mesh = g;
    S.node_base = tree.z;
    S.lds_count = tree.w;
    S.lds_first = __float_as_int(lds_geom4(L, g, 14).w);
    S.cur = 0;
    if constexpr (!LDS::big) S.sp = 0;
}

// Idle lane with candidate meshes left: enter the next one.
template <class LDS>
__device__ __forceinline__ void start_next_mesh(const LDS& L, const Ray& wr, Segment& S)
{
    const int g = __ffs((int)S.meshes) - 1;
    S.meshes &= S.meshes - 1u;
    enter_mesh(L, g, wr, S);
}

// Big scenes: the world-space half of the two-level traversal.  The lane's traversal state (ray, slab constants, box planes,
// node range) describes EITHER the geometry tree in world space (S.mesh < 0) OR one mesh in object space; the same inner
// step serves both.
template <class LDS>
__device__ __forceinline__ void enter_top(const LDS& L, const Ray& wr, Segment& S)
{
    S.osr = wr;
    S.ix = safe_rcp(wr.dx);
    S.iy = safe_rcp(wr.dy);
    S.iz = safe_rcp(wr.dz);
    S.ox = -wr.ox * S.ix;
    S.oy = -wr.oy * S.iy;
    S.oz = -wr.oz * S.iz;
    set_box_planes(L, S);
    S.scale = inv_length(wr); // ray parameter per unit of world distance (ff_intersect_rays takes rays of any length)
    refresh_tbound(S);
    S.mesh = -1;
    S.tl_sp = 0;
    S.node_base = L.top_first;
    S.lds_first = L.top_lds_first;
    S.lds_count = L.top_lds_count;
}

// A mesh is exhausted (S.cur == kMeshDone): back to the geometry tree, whose pending entries are on the stack below.
template <class LDS>
__device__ __forceinline__ void leave_mesh(const LDS& L, const uint4* __restrict__ nodes4, const Ray& wr, Segment& S)
{
    enter_top(L, wr, S);
    pop_subtree(L, nodes4, S);
}

// The cursor is on a leaf of the geometry tree: a plane or sphere is screened at once (kernel.cu:157-165 with the margins of
// screen_analytic), a mesh becomes the lane's current tree (kernel.cu:138: its object-space ray).  The slot's box test in
// the inner step has already pruned the geometry against what the lane held then.
template <bool STATS, class LDS>
__device__ __forceinline__ void geom_step(const LDS& L, int num_planes, const TriRecord* __restrict__ tris, const uint4* __restrict__ nodes4, const Ray& wr,
                                          Segment& S, Counters& cnt)
{
    const int g = (~S.cur) & (kGeomLeaf - 1);
    if (g < num_planes) {
        const float wlen = __builtin_amdgcn_rcpf(inv_length(wr));
        screen_analytic<STATS>(L, g, tris, wr, wlen, S, cnt);
        refresh_tbound(S);
        pop_subtree(L, nodes4, S);
    } else {
        const int floor = S.sp;
        if (STATS) cnt.no_mesh += 1; // (big scenes: the counter of plane-only queries counts mesh entries instead)
        enter_mesh(L, g, wr, S); // (leaves the cursor alone for a mesh without a tree)
        if (S.mesh == g) {
            S.tl_sp = floor;
            S.sp = floor;
        } else {
            pop_subtree(L, nodes4, S);
        }
    }
}

// One visit of a 4-wide node: test the four slot boxes, descend into the nearest hit, leave the others on the stack
// (nearest on top), or pop.  Pruning only: FMA + approximate 1/d on padded boxes with inflated bounds.
template <bool STATS, class LDS>
__device__ __forceinline__ void inner_step(const LDS& L, const uint4* __restrict__ nodes4, Segment& S, Counters& cnt)
{
    const int rel = S.cur;
    uint4 nx, ny, nz, fx, fy, fz, lk;
    if ((unsigned)rel < (unsigned)S.lds_count) {
        const char* const nb = reinterpret_cast<const char*>(ff_smem) + (S.lds_first + rel) * 16;
        nx = *reinterpret_cast<const uint4*>(nb + S.bnx);
        ny = *reinterpret_cast<const uint4*>(nb + S.bny);
        nz = *reinterpret_cast<const uint4*>(nb + S.bnz);
        fx = *reinterpret_cast<const uint4*>(nb + S.bfx);
        fy = *reinterpret_cast<const uint4*>(nb + S.bfy);
        fz = *reinterpret_cast<const uint4*>(nb + S.bfz);
        lk = *reinterpret_cast<const uint4*>(nb + 6 * 16 * L.node_cap);
        FF_PIN4(lk);
    } else {
        // (32-bit byte offsets from the array's base - a node index has 22 bits, kPackedEntry - so that the seven loads take the
        // base from a scalar register pair and one add each, instead of 64-bit address arithmetic per quarter)
        const char* const base = reinterpret_cast<const char*>(nodes4);
        static_assert(kNodeVec4 * 16 == 112, "node size");
        // (x 112 as two shifts: the compiler folds them back into the quarter-rate 32-bit multiply unless one is hidden from it)
        const unsigned ni = (unsigned)(S.node_base + rel);
        unsigned nb = ni << 7;
        asm volatile("" : "+v"(nb));
        nb -= ni << 4;
        const unsigned gx = S.ix < 0.0f ? 48u : 0u, gy = S.iy < 0.0f ? 64u : 16u, gz = S.iz < 0.0f ? 80u : 32u; // (set_box_planes)
        nx = *reinterpret_cast<const uint4*>(base + (nb + gx));
        ny = *reinterpret_cast<const uint4*>(base + (nb + gy));
        nz = *reinterpret_cast<const uint4*>(base + (nb + gz));
        fx = *reinterpret_cast<const uint4*>(base + (nb + (48u - gx)));
        fy = *reinterpret_cast<const uint4*>(base + (nb + (80u - gy)));
        fz = *reinterpret_cast<const uint4*>(base + (nb + (112u - gz)));
        lk = *reinterpret_cast<const uint4*>(base + (nb + 96u));
        FF_PIN4(lk);
    }
    if (STATS) { cnt.nodes += 1; probe_round(cnt.inner_rounds); }
    const float tbound = S.tbound;
    // slot c: entry parameter = the latest of the three near planes (and 0), exit = the earliest of the far planes (and the
    // pruning bound).  A hit slot sorts by its entry parameter: the key keeps the parameter's bits (non-negative floats
    // order like unsigned integers) with the slot number in the two low bits; a missed slot gets the largest key.
#define FF_SLOT_KEY(c, id)                                                                                                                       \
    ([&]() -> unsigned {                                                                                                                         \
        const float tn = fmaxf(fmaxf(__builtin_fmaf(__uint_as_float(nx.c), S.ix, S.ox), __builtin_fmaf(__uint_as_float(ny.c), S.iy, S.oy)),      \
                               fmaxf(__builtin_fmaf(__uint_as_float(nz.c), S.iz, S.oz), 0.0f));                                                  \
        const float tf = fminf(fminf(__builtin_fmaf(__uint_as_float(fx.c), S.ix, S.ox), __builtin_fmaf(__uint_as_float(fy.c), S.iy, S.oy)),      \
                               fminf(__builtin_fmaf(__uint_as_float(fz.c), S.iz, S.oz), tbound));                                                \
        return tn <= tf * 1.000002f ? ((__float_as_uint(tn) & ~3u) | (unsigned)(id)) : 0xFFFFFFFFu;                                              \
    }())
    unsigned k0 = FF_SLOT_KEY(x, 0), k1 = FF_SLOT_KEY(y, 1), k2 = FF_SLOT_KEY(z, 2), k3 = FF_SLOT_KEY(w, 3);
#undef FF_SLOT_KEY
    // five-comparator sorting network: k0 <= k1 <= k2 <= k3
    unsigned lo, hi;
    lo = min(k0, k1); hi = max(k0, k1); k0 = lo; k1 = hi;
    lo = min(k2, k3); hi = max(k2, k3); k2 = lo; k3 = hi;
    lo = min(k0, k2); hi = max(k0, k2); k0 = lo; k2 = hi;
    lo = min(k1, k3); hi = max(k1, k3); k1 = lo; k3 = hi;
    lo = min(k1, k2); hi = max(k1, k2); k1 = lo; k2 = hi;
    // Straight-line selects (the lanes of a wave disagree on every one of these cases): the nearest slot's link, and the
    // entry for the siblings to come back to: one -> its link; more -> the node and their slots, nearest first.
    const int near_link = select_slot(lk, (int)(k0 & 3u));
    const int second_link = select_slot(lk, (int)(k1 & 3u));
    const int packed = (int)((unsigned)kPackedEntry | ((unsigned)rel << 8) | ((k3 & 3u) << 6) | ((k2 & 3u) << 4) | ((k1 & 3u) << 2) |
                             (k3 != 0xFFFFFFFFu ? 3u : 2u));
    const int entry = k2 == 0xFFFFFFFFu ? second_link : packed;
    if (k1 != 0xFFFFFFFFu) {
        if (STATS && S.sp >= L.stack_depth) cnt.stack_overflow += 1; // (with a spill area: entries that went there)
        stack_push(L, S.sp, entry);
        ++S.sp;
    }
    if (k0 != 0xFFFFFFFFu) S.cur = near_link;
    else pop_subtree(L, nodes4, S);
}

// One leaf visit: test the leaf's triangles (fast form), then take the next entry off the stack.  On a near tie with the
// pending candidate the leaf is left under the cursor with S.resume set; the caller resolves the pending candidate and
// the leaf continues from the triangle that met the tie.
template <bool STATS, class LDS>
__device__ __forceinline__ void leaf_step(const LDS& L, const TriRecord* __restrict__ tris, const uint4* __restrict__ nodes4, const Ray& wr, Segment& S,
                                          Counters& cnt)
{
    const int ref = ~S.cur;
    const int first = ref >> 3, count = (ref & 7) + 1;
    const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)first * 3;
    const Ray& r = S.osr;
    int k = S.resume > 0 ? S.resume - 1 : 0;
    S.resume = 0;
    if (STATS) probe_round(cnt.leaf_rounds);
    unsigned long long tl_wait = 0, tl_test = 0, tl0 = 0;
    // The next triangle's record is requested before this one is tested (its wait overlaps the arithmetic; the last round asks
    // for its own record again, a hit in the L1).
    float4 An = tp[3 * k], E1n = tp[3 * k + 1], E2n = tp[3 * k + 2];
    for (; k < count; ++k) {
        if (STATS) tl0 = __builtin_amdgcn_s_memtime();
        const float4 A = An, E1 = E1n, E2 = E2n;
        {
            const int kn = min(k + 1, count - 1);
            An = tp[3 * kn]; E1n = tp[3 * kn + 1]; E2n = tp[3 * kn + 2];
        }
        if (STATS) {
            // (instrumented launches only: the wait for THIS triangle's three loads is made explicit so that it can be told from the
            // arithmetic; the three youngest loads - the next triangle's, issued just above - stay in flight as in the real kernel)
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            const unsigned long long tl1 = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) tl_wait += tl1 - tl0; // one lane per round keeps the wave's time
            tl0 = tl1;
        }
        if (STATS) { cnt.tris += 1; probe_round(cnt.tri_rounds); }
        // kernel.cu:44-75: exact up to the division; every accept/reject comparison is the reference's own
        const float e1x = E1.x, e1y = E1.y, e1z = E1.z;
        const float e2x = E2.x, e2y = E2.y, e2z = E2.z;
        const float px = r.dy * e2z - e2y * r.dz, py = r.dz * e2x - e2z * r.dx, pz = r.dx * e2y - e2x * r.dy;
        const float det = dot3(e1x, e1y, e1z, px, py, pz);
        const float tx = r.ox - A.x, ty = r.oy - A.y, tz = r.oz - A.z;
        const float u = dot3(tx, ty, tz, px, py, pz);
        const float qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;
        const float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz);
        const float tn = dot3(e2x, e2y, e2z, qx, qy, qz);
        bool ok = !(det < kTriEpsilon) && !(u < 0.0f || u > det) && !(v < 0.0f || u + v > det);
        if (ok && det < kTriEpsilon + E1.w) {
            // the back-face test of kernel.cu:48-49 could disagree with the sign of det only this close to edge-on
            const float nx = e1y * e2z - e2y * e1z, ny = e1z * e2x - e2z * e1x, nz = e1x * e2y - e2x * e1y;
            ok = !(dot3(r.dx, r.dy, r.dz, nx, ny, nz) > 0.0f);
        }
        if (ok) {
            float ta = tn * __builtin_amdgcn_rcpf(det); // approximate t (kernel.cu:77-79 is exact: 1/det, then multiply)
            if (ta < kTriEpsilon * 1.001f) {
                if (ta > kTriEpsilon * 0.999f) {
                    ta = tn * ieee_rcp(det); // within the margin of the t > EPSILON test: decide exactly (kernel.cu:97)
                    ok = ta > kTriEpsilon;
                } else {
                    ok = false;
                }
            }
            if (ok && offer(ta * __builtin_amdgcn_rcpf(S.scale), S.mesh, first + k, S.pend, S.best)) {
                S.resume = k + 1;
                if (STATS) tl_test += __builtin_amdgcn_s_memtime() - tl0; // (a near tie: rare, the lane's own count)
                break;
            }
        }
        if (STATS) {
            const unsigned long long tl2 = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) tl_test += tl2 - tl0;
        }
    }
    refresh_tbound(S);
    if (STATS) {
        cnt.t_l1 += tl_wait;
        cnt.t_l2 += tl_test;
        tl0 = __builtin_amdgcn_s_memtime();
    }
    if (S.resume > 0) return;
    pop_subtree(L, nodes4, S);
    if (STATS) {
        const unsigned long long tl3 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) cnt.t_l3 += tl3 - tl0;
    }
}

// Settle what is still pending (the common case: the one exact evaluation of the ray, all hitting lanes together) and
// produce the hit point of the winner.
template <class LDS>
__device__ __forceinline__ void finish_segment(const LDS& L, const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Best& best)
{
    bool have_point = false;
    HitPoint H = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    if (S.pend.geom >= 0) have_point = resolve_pending(L, tris, wr, S.pend, S.best, H);
    if (!have_point && S.best.geom >= 0) {
        // the winner was resolved earlier (two candidates had been too close to rank approximately): recompute its point
        float dist;
        int orig_tri;
        exact_hit(L, tris, wr, S.best.geom, S.best.rec, dist, H, orig_tri);
    }
    best.dist = S.best.dist;
    best.geom = S.best.geom;
    best.rec = S.best.rec;
    best.px = H.wx; best.py = H.wy; best.pz = H.wz;
    best.cx = H.cx; best.cy = H.cy; best.cz = H.cz;
}

// Nothing left to do in the query.
__device__ __forceinline__ bool segment_done(const Segment& S) { return S.cur == kDone && S.meshes == 0u && S.resume == 0; }

// Advance the queries of the calling lanes: mesh starts, inner-node phases, leaf phases and near-tie resolutions alternate
// wave-wide until every calling lane is done or `budget` inner-node rounds have been spent (budget <= 0: no limit).
// Unfinished lanes keep their state in S and continue on the next call.
template <bool STATS, class LDS>
__device__ __forceinline__ void traverse_budget(const LDS& L, const TriRecord* __restrict__ tris, const uint4* __restrict__ nodes4, const Ray& wr,
                                                Segment& S, Counters& cnt, int budget, int leaf_threshold, int num_planes = 0)
{
    const int limit = budget > 0 ? budget : kLoopGuard;
    int rounds = 0, guard = 0;
    for (;;) {
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
        if (STATS) ta = __builtin_amdgcn_s_memtime();
        if constexpr (LDS::big) {
            // lanes whose mesh is exhausted resume the geometry tree; lanes on a geometry leaf screen it or enter its mesh.
            // ONE round per iteration: a lane that pops straight into another geometry leaf waits for the next quorum instead of
            // being served with the two or three others in its situation (108 geometries: +5 %, 258: +8 %).
            {
                const bool back = S.cur == kMeshDone, geom = S.cur < 0 && ((~S.cur) & kGeomLeaf) != 0;
                if (__ballot(back || geom) != 0ull) {
                    if (back) leave_mesh(L, nodes4, wr, S);
                    else if (geom) geom_step<STATS>(L, num_planes, tris, nodes4, wr, S, cnt);
                }
            }
        } else {
            // A lane enters its next candidate mesh (object-space ray, slab constants: ~80 instructions) together with the lanes that
            // start their query, at the top of a slice; in mid-slice, where one or two lanes at a time would ask for it, it waits for
            // the next slice - unless no lane of the wave has anything else to traverse.
            if (guard == 0 || budget <= 0 || __ballot(S.cur != kDone) == 0ull)
                while (S.cur == kDone && S.meshes != 0u) start_next_mesh(L, wr, S);
        }
        if (STATS) tb = __builtin_amdgcn_s_memtime();
        if (__ballot(S.cur != kDone) == 0ull) break;
        bool leaves_due = true;
        for (;;) {
            const bool inner = (unsigned)S.cur < (unsigned)kMeshDone;
            if (__ballot(inner) == 0ull) break;
            // enough lanes hold a leaf: test the leaves now instead of idling them until the last lane finds one
            const bool tri_leaf = S.cur < 0 && !(LDS::big && ((~S.cur) & kGeomLeaf) != 0);
            if (__popcll(__ballot(tri_leaf)) >= leaf_threshold) break;
            if (rounds >= limit) break;
            if constexpr (LDS::big) {
                // enough lanes wait on the geometry tree (a geometry leaf, an exhausted mesh): serve them first; the few lanes
                // that hold triangles keep them for a fuller leaf phase
                if (__popcll(__ballot(S.cur == kMeshDone || (S.cur < 0 && !tri_leaf))) >= leaf_threshold) {
                    leaves_due = false;
                    break;
                }
            }
            ++rounds;
            if (inner) inner_step<STATS>(L, nodes4, S, cnt);
        }
        if (STATS) tc = __builtin_amdgcn_s_memtime();
        if (leaves_due && S.cur < 0 && !(LDS::big && ((~S.cur) & kGeomLeaf) != 0)) leaf_step<STATS>(L, tris, nodes4, wr, S, cnt);
        if (STATS) {
            td = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { cnt.t_start += tb - ta; cnt.t_inner += tc - tb; cnt.t_leaf += td - tc; }
        }
        // (a lane can wait with its leaf half tested for several turns in a big scene, where an iteration may serve the geometry
        // tree instead of the leaves: its pending candidate is resolved once, the first time round)
        if (__ballot(S.resume > 0 && S.pend.geom >= 0) != 0ull) {
            if (S.resume > 0 && S.pend.geom >= 0) {
                HitPoint H;
                resolve_pending(L, tris, wr, S.pend, S.best, H);
                refresh_tbound(S);
            }
        }
        if (rounds >= limit) break;
        if (++guard > kLoopGuard) break; // never reached by a well-formed tree; bounds the loop so no wave can spin forever
    }
    // The guard is a bound on a hang, not a way to end a query: a lane it cut short holds a truncated closest hit.  Make that
    // visible in every build (run-to-completion callers: any unfinished lane; time-sliced callers: only the iteration guard).
    if (budget <= 0 || guard > kLoopGuard) cnt.guard_hits |= __ballot(!segment_done(S));
}

// A complete closest-hit query for every calling lane (ray-batch kernel).
template <bool STATS, class LDS>
__device__ __forceinline__ void closest_hit_deferred(const LDS& L, const WallTable& W, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                                     const TriRecord* __restrict__ tris, const uint4* __restrict__ nodes4, const Ray& wr,
                                                     Best& best, Counters& cnt)
{
    Segment S;
    if (STATS) probe_round(cnt.segment_rounds);
    begin_segment<STATS>(L, W, geoms, num_geoms, num_planes, tris, wr, S, cnt);
    traverse_budget<STATS>(L, tris, nodes4, wr, S, cnt, 0, 64, num_planes);
    finish_segment(L, tris, wr, S, best);
    cnt.rays += 1;
}

// Brute-force closest hit: the reference's loop (kernel.cu:133-155) with the triangle array streamed through LDS in
// batches that the whole workgroup stages with coalesced 16-byte loads and then reads at a wave-uniform address.
// Must be called by every thread of the workgroup (it contains barriers); `live` masks lanes without a ray.
template <bool STATS>
__device__ __forceinline__ void closest_hit_brute(const GeomRecord* __restrict__ geoms, int num_geoms, const TriRecord* __restrict__ tris,
                                                  float4* batch, bool live, const Ray& wr, Best& best, Counters& cnt,
                                                  const float4* __restrict__ trinormals = nullptr)
{
    best.dist = kInf;
    best.geom = -1;
    best.rec = -1;
    best.px = best.py = best.pz = 0.0f;
    best.cx = best.cy = 0.0f;
    best.cz = 1.0f;
    for (int g = 0; g < num_geoms; ++g) {
        const GeomRecord& G = geoms[g];
        Ray osr;
        float len;
        object_space_ray(G, wr, osr, len);
        if (G.type == FF_GEOM_TRIANGLEMESH) {
            for (int base = 0; base < G.tri_count; base += kBruteBatchTris) {
                const int nb = min(kBruteBatchTris, G.tri_count - base);
                __syncthreads();
                const float4* src = reinterpret_cast<const float4*>(tris) + (size_t)(G.tri_first + base) * 3;
                for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) batch[i] = src[i];
                __syncthreads();
                if (live) {
                    for (int k = 0; k < nb; ++k) {
                        const float4 a = batch[3 * k], b = batch[3 * k + 1], c = batch[3 * k + 2];
                        const float t = triangle_t(a, b, c, osr);
                        if (t > 0.0f) consider(G, g, G.tri_first + base + k, __float_as_int(a.w), t, osr, wr, geoms, tris, best);
                    }
                    if (STATS) cnt.tris += (unsigned)nb;
                }
            }
        } else if (live) {
            if (STATS) cnt.planes += 1;
            const float t = G.type == FF_GEOM_SPHERE ? sphere_t(G.plane_n[3], osr) : plane_t(G.plane_n[0], G.plane_n[1], G.plane_n[2], osr);
            if (t > 0.0f) consider(G, g, -1, -1, t, osr, wr, geoms, tris, best);
        }
    }
    if (live) {
        fill_object_normal(geoms, tris, best);
        fill_sphere_normal(geoms, wr, best);
        fill_smooth_normal(geoms, tris, trinormals, wr, best);
        cnt.rays += 1;
    }
}

// ---- shading --------------------------------------------------------------------------------------------------------

// What shading needs from the hit geometry's record, fetched piece by piece when it is used (loading the whole record up
// front costs ~25 registers at the kernel's pressure peak).  BVH kernels read the LDS copy, brute-force kernels the
// global one.
struct MaterialRef {
    const GeomRecord* global; // non-null: read the global record
    int geom_base;            // else: the LDS copy (uint4 index of record 0, Lds::geom_base)
    int g;
};

__device__ __forceinline__ float4 mat_f4(const MaterialRef& M, int k)
{
    if (M.global) return reinterpret_cast<const float4*>(M.global)[k];
    return reinterpret_cast<const float4*>(ff_smem)[M.geom_base + M.g * kGeomVec4 + k];
}
__device__ __forceinline__ int mat_bxdf(const MaterialRef& M)
{
    if (M.global) return M.global->bxdf_type;
    return reinterpret_cast<const int4*>(ff_smem)[M.geom_base + M.g * kGeomVec4 + 16].y;
}

// `unit_object_normal`: normalise a triangle's face normal in object space first (kernel.cu:101, what Intersect::m_normal
// and the NORMAL_DEBUG shade carry); the path integrator transforms the raw cross product and normalises once in world space.
__device__ __forceinline__ void world_normal(const MaterialRef& M, const Best& best, bool unit_object_normal, float& nx, float& ny, float& nz)
{
    float ox = best.cx, oy = best.cy, oz = best.cz;
    if (best.rec >= 0 && unit_object_normal) {
        const float inv = ieee_rcp(ieee_sqrt(dot3(ox, oy, oz, ox, oy, oz)));
        ox = ox * inv;
        oy = oy * inv;
        oz = oz * inv;
    }
    const float4 n0 = mat_f4(M, 8), n1 = mat_f4(M, 9), n2 = mat_f4(M, 10); // inverse-transpose columns (w = column3 * 0)
    nx = (n0.x * ox + n1.x * oy) + (n2.x * oz + n0.w);
    ny = (n0.y * ox + n1.y * oy) + (n2.y * oz + n1.w);
    nz = (n0.z * ox + n1.z * oy) + (n2.z * oz + n2.w);
}

// ---- build-defined integrator pieces (DESIGN.md "Integrator"; mirrored by the oracle) ------------------------------

// Philox2x32-10 (Salmon et al., SC'11): counter-based, so a sample's random numbers depend only on
// (global pixel index, sample, bounce, seed) and not on which lane, wave, launch or GPU computes it.
__device__ __forceinline__ void philox2x32_10(unsigned c0, unsigned c1, unsigned key, unsigned& o0, unsigned& o1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) key += 0x9E3779B9u;
        const unsigned long long prod = (unsigned long long)c0 * 0xD256D193ull; // one v_mad_u64_u32 yields both halves
        c0 = (unsigned)(prod >> 32) ^ key ^ c1;
        c1 = (unsigned)prod;
    }
    o0 = c0;
    o1 = c1;
}

// utilities.h:46-55 CosineSampleHemisphere with theta = 2*pi*k24/2^24 reduced exactly to an octant on the integer and
// fixed-order polynomials on [0, pi/4] (bit-identical to the oracle).
__device__ __forceinline__ void cosine_sample(float u1, unsigned k24, float& x, float& y, float& z)
{
    const unsigned oct = k24 >> 21, f = k24 & 0x1FFFFFu;
    const unsigned m = (oct & 1u) ? (0x200000u - f) : f;
    const float a = (float)m * 3.7450704e-07f;
    const float a2 = a * a;
    float sp = -1.9841270e-04f + a2 * 2.7557319e-06f;
    sp = 8.3333333e-03f + a2 * sp;
    sp = -1.6666667e-01f + a2 * sp;
    const float s = a + (a * a2) * sp;
    float cp = -1.3888889e-03f + a2 * 2.4801587e-05f;
    cp = 4.1666667e-02f + a2 * cp;
    cp = -0.5f + a2 * cp;
    const float c = 1.0f + a2 * cp;
    float sn, cs;
    if ((oct + 1u) & 2u) { sn = c; cs = s; } else { sn = s; cs = c; }
    if (oct >= 4u) sn = -sn;
    if (oct >= 2u && oct <= 5u) cs = -cs;
    const float r = ieee_sqrt(u1);
    x = r * cs;
    y = r * sn;
    z = ieee_sqrt(fmaxf(0.0f, 1.0f - u1));
}

__device__ __forceinline__ unsigned char to_u8(float v)
{
    // kernel.cu:214 float -> unsigned char (truncation); out-of-range values are UB there and clamp here
    const float s = v * 255.0f;
    if (!(s > 0.0f)) return 0;
    if (s >= 255.0f) return 255;
    return (unsigned char)s;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned olo = __shfl_xor(lo, off), ohi = __shfl_xor(hi, off);
        const unsigned long long s = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)ohi << 32) | olo);
        lo = (unsigned)s;
        hi = (unsigned)(s >> 32);
    }
    return ((unsigned long long)hi << 32) | lo;
}

// Per-lane path state.
struct Path {
    int item;      // pitem | block << 25 (the pixel's tile-major item number and the sample block whose sum this lane keeps: blocksums[pitem][block]);
                   // negative (kItemTail | pitem): a tail item, whose samples are stored one by one (tail_samples[sample of the block][pitem])
    int send;      // one past the last sample of the block
    unsigned gxy;  // global pixel coordinates x | y << 16; the RNG counter is the pixel index y*W+x (kernel.cu:191)
    int s, b;      // current sample / segment
    float pdx, pdy, pdz; // primary direction of the pixel (every sample starts with the same ray: kernel.cu:200-205 has no jitter)
    Ray ray;       // current world-space ray
    float bx, by, bz; // throughput
    float ax, ay, az; // running sum over samples
};

// kernel.cu:197-205 for global pixel (x, y): origin = camera position, direction through the pixel corner.
__device__ __forceinline__ void primary_ray(const KParams& p, unsigned gxy, Ray& ray)
{
    const int x = (int)(gxy & 0xFFFFu), y = (int)(gxy >> 16);
    const float Px = ((float)x / p.screen_w) * 2.f - 1.f;  // :200
    const float Py = 1.f - ((float)y / p.screen_h) * 2.f;  // :201
    const float v0 = Px * p.far_clip, v1 = Py * p.far_clip, v2 = 1.f * p.far_clip, v3 = 1.f * p.far_clip;
    const float wx = (p.cam_c0[0] * v0 + p.cam_c1[0] * v1) + (p.cam_c2[0] * v2 + p.cam_c3[0] * v3); // :203
    const float wy = (p.cam_c0[1] * v0 + p.cam_c1[1] * v1) + (p.cam_c2[1] * v2 + p.cam_c3[1] * v3);
    const float wz = (p.cam_c0[2] * v0 + p.cam_c1[2] * v1) + (p.cam_c2[2] * v2 + p.cam_c3[2] * v3);
    const float ddx = wx - p.cam_pos[0], ddy = wy - p.cam_pos[1], ddz = wz - p.cam_pos[2];
    const float inv = ieee_rcp(ieee_sqrt(dot3(ddx, ddy, ddz, ddx, ddy, ddz))); // :205
    ray.ox = p.cam_pos[0];
    ray.oy = p.cam_pos[1];
    ray.oz = p.cam_pos[2];
    ray.dx = ddx * inv;
    ray.dy = ddy * inv;
    ray.dz = ddz * inv;
}

__device__ __forceinline__ void start_sample(const KParams& p, Path& P)
{
    P.b = 0;
    P.ray.ox = p.cam_pos[0];
    P.ray.oy = p.cam_pos[1];
    P.ray.oz = p.cam_pos[2];
    P.ray.dx = P.pdx;
    P.ray.dy = P.pdy;
    P.ray.dz = P.pdz;
    P.bx = P.by = P.bz = 1.f;
}

// The part of the work queue a wave owns: items [next, end).  Wave-uniform (scalar registers).
struct WaveQueue {
    unsigned next, end; // in the counter's own numbering (see queue_item)
    unsigned counter;   // which counter the wave draws from
    unsigned owned;     // how many items that counter owns
    unsigned main_end;  // its items [0, main_end) go out in chunks, the last ones [main_end, owned) exactly as asked for (acquire_pixel)
    bool in_tail;       // the chunked part has run dry: the wave draws from the tail counter
    bool dry;           // nothing left at all
};

// The launch's items are dealt to the counters in stripes of kQueueStripe: counter c of n owns the stripes c, c + n, ... so
// every counter covers the whole image evenly and they run dry together.  Number v of counter c is this item:
__device__ __forceinline__ unsigned queue_item(const KParams& p, const WaveQueue& Q, unsigned v)
{
    return ((v / kQueueStripe) * (unsigned)p.queue_counters + Q.counter) * kQueueStripe + (v % kQueueStripe);
}

__device__ __forceinline__ WaveQueue make_wave_queue(const KParams& p)
{
    WaveQueue Q;
    const unsigned n = (unsigned)p.queue_counters, c = blockIdx.x % n;
    const unsigned stripes = (p.total_items + kQueueStripe - 1) / kQueueStripe;
    const unsigned mine = stripes > c ? (stripes - c + n - 1) / n : 0u;
    // the last stripe of the range may be short
    const unsigned cut = mine > 0u && (stripes - 1u) % n == c ? stripes * kQueueStripe - p.total_items : 0u;
    Q.next = Q.end = 0u;
    Q.counter = c;
    Q.owned = mine * kQueueStripe - cut;
    Q.main_end = Q.owned - min(Q.owned, p.queue_tail_items);
    Q.in_tail = Q.main_end == 0u;
    Q.dry = Q.owned == 0u;
    return Q;
}

// Pull the next traceable pixel for every calling lane.  Two levels: a wave takes a CHUNK of consecutive items from the global
// counter (one atomic: what its idle lanes ask for, at least queue_chunk items) and deals them to its lanes with no memory
// traffic; what is left over serves the wave's next requests.  One counter serves about 10^8 atomics a second, each waiting
// behind the others': a wave asking it for every item held 1-spp frames (two million one-path items) to a third of the
// saturated rate.  So the launch spreads its waves over several counters in different memory channels (queue_item: each owns
// an even share of the image; no stealing: they run dry together), and queue_chunk is sized by the host so that a chunk is a
// few dozen samples of work, whatever the item length.  (Chunks that shrink with what is left - guided self-scheduling, up to a
// tile of pixels per wave - were measured: the big early chunks unbalance frames whose cost varies across the image, the
// reference's default camera 191 ms instead of 84.)
// Called by ALL lanes of the wave (Q must stay wave-uniform); `need` marks the lanes that want a pixel.  Returns false for lanes
// that did not ask or saw the end of the queue.
__device__ __forceinline__ bool acquire_pixel(const KParams& p, int lane, Path& P, WaveQueue& Q, bool need, unsigned& rays)
{
    bool got = false, exhausted = !need;
    for (;;) {
        const bool want = !got && !exhausted;
        const unsigned long long m = __ballot(want);
        if (m == 0ull) break;
        if (Q.next >= Q.end) {
            if (Q.dry) {
                exhausted = true;
                continue;
            }
            const int leader = __ffsll((long long)m) - 1;
            if (!Q.in_tail) {
                const unsigned size = max((unsigned)__popcll(m), p.queue_chunk);
                unsigned base = 0u;
                if (lane == leader) base = atomicAdd(p.queue + Q.counter * kQueueStride, size);
                base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
                if (base < Q.main_end) {
                    Q.next = base;
                    Q.end = min(base + size, Q.main_end);
                } else {
                    Q.in_tail = true;
                }
            }
            if (Q.in_tail && Q.next >= Q.end) {
                // The LAST items of the counter's share are handed out exactly as asked for, from a counter of their own.  A chunk is
                // the wave's private stock: it deals it to its own lanes as they fall idle, and a wave that takes 64 sample blocks when
                // two of its lanes are idle works through the other 62 long after every other wave has run dry - the launch's dry end
                // (an eighth of a multi-GPU rank's frame; 40 % of a 1-spp frame: tools/timeline_probe.py).  In the tail zone a wave holds
                // nothing it has no lane for.
                const unsigned size = (unsigned)__popcll(m);
                unsigned base = 0u;
                if (lane == leader) base = atomicAdd(p.queue + Q.counter * kQueueStride + kQueueTailWord, size);
                base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
                if (Q.main_end + base >= Q.owned) {
                    Q.dry = true;
                    Q.next = Q.end = Q.owned;
                    continue;
                }
                Q.next = Q.main_end + base;
                Q.end = min(Q.next + size, Q.owned);
            }
        }
        const unsigned rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        const unsigned avail = Q.end - Q.next, asked = (unsigned)__popcll(m);
        const bool take = want && rank < avail;
        const unsigned item = queue_item(p, Q, Q.next + rank);
        Q.next += min(asked, avail);
        if (take) {
            {
                // an item is one sample block of one pixel; pixels walk 8x8 tiles of the local image (padding items and
                // untraced pixels are consumed and skipped)
                // items [0, tail_first_item): (pixel, whole block), PIXEL-major: a wave's chunk covers the blocks of one or a few
                // pixels, whose sums are neighbours in blocksums[pixel][block]: stored by one wave, they meet in one XCD's L2
                // and leave it as whole lines instead of one masked 64-byte write per 16-byte sum (0.59 instead of 1.93 GB of
                // HBM writes per 1080p 1 024-spp frame); beyond: (pixel, sample group) of the tail block, group-major
                const bool tail = p.tail_block >= 0 && item >= p.tail_first_item;
                const unsigned rel = tail ? item - p.tail_first_item : item;
                unsigned blk, pitem; // tail: blk is the group index
                // (Five divisions by launch constants per item.  Host-computed magic multipliers - mulhi, two shifts, two adds each -
                // were measured: C2 -0.6 %, C3 -1.5 %, the 1-spp frame +-0: v_mul_hi_u32 is a quarter-rate instruction and the compiler's
                // reciprocal-based expansion is not slower.)
                if (tail) {
                    blk = rel / p.pix_items;
                    pitem = rel - blk * p.pix_items;
                } else {
                    pitem = rel / p.whole_blocks;
                    blk = rel - pitem * p.whole_blocks;
                }
                // A camera outside the scene (the reference's default one looks at its box from 12.5 units away: 96 % of the frame is
                // background): cull_mask_kernel has marked the pixels whose primary ray misses the padded box around ALL geometries
                // and zeroed their block sums - kernel.cu:200-205 has no jitter, every sample of the pixel starts with that ray.
                // Their items end here: the rays are counted (each is a closest-hit query with the answer "nothing", as in the
                // brute-force loop, which takes no such shortcut) and the lane asks for the next item.
                if (p.cull_mask != nullptr && !tail && ((p.cull_mask[pitem >> 6] >> (pitem & 63u)) & 1ull) != 0ull) {
                    const int s0 = (p.block_begin + (int)blk) * p.block_spp;
                    rays += (unsigned)(p.shade_mode == FF_SHADE_NORMAL_DEBUG ? 1 : min(p.spp_total, s0 + p.block_spp) - s0);
                    continue;
                }
                const int tile = (int)(pitem >> 6), in = (int)(pitem & 63u);
                const int lx = (tile % p.tiles_per_row) * 8 + (in & 7);
                const int ly = (tile / p.tiles_per_row) * 8 + (in >> 3);
                const int strip = ly / p.strip_rows;
                const int gy = p.y0 + (strip * p.num_parts + p.part) * p.strip_rows + (ly - strip * p.strip_rows);
                const int gx = p.x0 + lx;
                if (lx < p.local_width && gx < p.xlim && ly < p.local_rows && gy < p.ylim) {
                    got = true;
                    P.gxy = (unsigned)gx | ((unsigned)gy << 16);
                    if (tail) {
                        const int first = p.tail_start[blk], past = p.tail_start[blk + 1]; // the group's samples inside the block
                        P.item = (int)(kItemTail | pitem);
                        P.s = p.tail_block * p.block_spp + first;
                        P.send = min(p.spp_total, p.tail_block * p.block_spp + past);
                    } else {
                        const int block = p.block_begin + (int)blk;
                        P.item = (int)(((unsigned)block << kItemBlockShift) | pitem);
                        P.s = block * p.block_spp;
                        P.send = min(p.spp_total, P.s + p.block_spp);
                    }
                    P.ax = P.ay = P.az = 0.f;
                    primary_ray(p, P.gxy, P.ray); // once per (pixel, block); its samples reuse the direction
                    P.pdx = P.ray.dx;
                    P.pdy = P.ray.dy;
                    P.pdz = P.ray.dz;
                    start_sample(p, P);
                    // A camera outside the scene (the reference's default one looks at its box from 12.5 units away: 96 % of the
                    // frame is background): a pixel whose primary ray misses the padded box around ALL geometries has no hit in any
                    // sample - kernel.cu:200-205 has no jitter, every sample starts with the same ray - so the whole item is a sum of
                    // zeros.  It is written at once, its rays are counted (each is a closest-hit query with the answer "nothing", as
                    // in the brute-force loop, which takes no such shortcut), and the lane asks for the next item.
                }
            }
        }
    }
    return got;
}

// Shading comes in two steps so that the trace kernel can run the expensive one once per iteration (trace_bvh_kernel):
//   settle_hit: what the finished segment means for the path - it ends (on an emitter, on nothing, at the last bounce; the sample's
//     radiance goes to the block sum and the next sample or the end of the block follows) or it goes on from this hit;
//   scatter:    the next ray of a path that goes on (normal, random numbers, new direction).
// settle_hit returns kPixelDone (the lane gives the pixel up), kNewSample (the next sample's primary ray is in P.ray) or kGoesOn
// (scatter must follow with the same hit).  SPECULAR = false: the caller guarantees a scene without MIRROR / GLASS surfaces.
enum { kPixelDone = 0, kNewSample = 1, kGoesOn = 2 };
template <bool SPECULAR = true, bool PREPASS = false>
__device__ __forceinline__ int settle_hit(const KParams& p, const Best& best, bool hit, const MaterialRef& M, Path& P)
{
    if constexpr (PREPASS) {
        // The pre-pass of a frame: every pixel's primary ray, traced ONCE (kernel.cu:200-205 sends all samples of a pixel through the
        // pixel's corner: no jitter), its closest hit stored per pixel; the frame's samples start from there (trace_bvh_kernel).
        float4* out = p.primary_hits + ((unsigned)P.item & kItemPixelMask);
        out[0] = make_float4(best.dist, best.px, best.py, best.pz);
        out[p.pix_items] = make_float4(best.cx, best.cy, best.cz, __int_as_float(hit ? best.geom : -1));
        out[2 * (size_t)p.pix_items] = make_float4(__int_as_float(best.rec), 0.f, 0.f, 0.f);
        if (!hit && p.cull_mask_out != nullptr) {
            // Nothing in view: every sample of the pixel adds zero.  Its bit goes into the mask the work queue consults (acquire_pixel:
            // its whole-block items are dropped when they are decoded, their rays counted), its block sums are zeroed and it is counted -
            // exactly what cull_mask_kernel does for the pixels whose ray misses the box around the scene, for those that miss the
            // scene itself (a pixel that pass has marked already is left alone).
            const unsigned pitem = (unsigned)P.item & kItemPixelMask;
            const unsigned long long bit = 1ull << (pitem & 63u);
            const unsigned long long old = atomicOr(&p.cull_mask_out[pitem >> 6], bit);
            if ((old & bit) == 0ull) {
                for (int b = 0; b < p.frame_blocks; ++b) p.blocksums[(size_t)pitem * p.frame_blocks + b] = make_float4(0.f, 0.f, 0.f, 0.f);
                atomicAdd(&p.counters[kCulledPixelsWord + kRaySlotStride * ((pitem >> 6) % kRaySlots)], 1ull);
            }
        }
        return kPixelDone;
    }
    const bool debug_shade = p.shade_mode == FF_SHADE_NORMAL_DEBUG;
    // Radiance of the path: it is zero until the path ends on an emitter (the only light transport here), so it is not
    // carried across segments; "0 + beta*Le" of the integrator is beta*Le bit for bit.
    float Lx = 0.f, Ly = 0.f, Lz = 0.f;
    if (hit) {
        if (debug_shade) {
            // shade(), kernel.cu:178-184
            float nx, ny, nz;
            world_normal(M, best, true, nx, ny, nz);
            Lx = fabsf(nx); Ly = fabsf(ny); Lz = fabsf(nz);
        } else if (mat_bxdf(M) == FF_BXDF_EMITTER) {
            // utilities.h:96-103: two-sided emitter, m_emissiveColor * m_intensity
            const float4 emission = mat_f4(M, 13);
            Lx = 0.f + P.bx * emission.x;
            Ly = 0.f + P.by * emission.y;
            Lz = 0.f + P.bz * emission.z;
        } else {
            // MIRROR: throughput *= m_specularColor (the record's tint slot holds it).  GLASS: the tint depends on the choice between
            // reflection and refraction (scatter).  Everything else is diffuse (utilities.h:109): cosine-weighted sampling, so
            // f*cos/pdf = albedo.
            const bool glass = SPECULAR && mat_bxdf(M) == FF_BXDF_GLASS;
            const float4 albedo = mat_f4(M, 12);
            if (!glass) {
                P.bx = P.bx * albedo.x;
                P.by = P.by * albedo.y;
                P.bz = P.bz * albedo.z;
            }
            if (P.b != p.bounces - 1) return kGoesOn;
        }
    }
    if (P.item < 0) {
        // tail item: every sample is stored on its own (sample-major: [sample of the block][pixel item]); the combine pass adds the
        // block's samples in order
        p.tail_samples[(size_t)(P.s - p.tail_block * p.block_spp) * p.pix_items + ((unsigned)P.item & kItemPixelMask)] = make_float4(Lx, Ly, Lz, 0.f);
    } else {
        P.ax = P.ax + Lx;
        P.ay = P.ay + Ly;
        P.az = P.az + Lz;
    }
    ++P.s;
    if (P.s < P.send && !debug_shade) {
        start_sample(p, P);
        return kNewSample;
    }
    // sample block finished: its sum goes to the block buffer (the combine kernel adds a pixel's blocks in order)
    if (P.item >= 0) p.blocksums[(size_t)((unsigned)P.item & kItemPixelMask) * p.num_blocks + ((unsigned)P.item >> kItemBlockShift)] = make_float4(P.ax, P.ay, P.az, 0.f);
    return kPixelDone;
}

// The next ray of a path that goes on from `best` (settle_hit returned kGoesOn; the throughput already carries the surface's
// albedo, glass excepted).  MIRROR: perfect reflection.  GLASS: smooth dielectric, Fresnel-weighted choice between reflection and
// refraction (oracle/ff_oracle.c is the definition).  Everything else: cosine-weighted direction about the world normal.
template <bool SPECULAR = true>
__device__ __forceinline__ void scatter(const KParams& p, const Best& best, const MaterialRef& M, Path& P)
{
    float nx, ny, nz;
    world_normal(M, best, false, nx, ny, nz);
    const int bxdf = mat_bxdf(M);
    const bool mirror = SPECULAR && bxdf == FF_BXDF_MIRROR, glass = SPECULAR && bxdf == FF_BXDF_GLASS;
    const float4 albedo = mat_f4(M, 12);
    const float ninv = ieee_rcp(ieee_sqrt(dot3(nx, ny, nz, nx, ny, nz)));
    float ux = nx * ninv, uy = ny * ninv, uz = nz * ninv;
    bool flipped = false;
    if (dot3(ux, uy, uz, P.ray.dx, P.ray.dy, P.ray.dz) > 0.0f) { ux = -ux; uy = -uy; uz = -uz; flipped = true; }
    float wox, woy, woz;
    float sx = ux, sy = uy, sz = uz; // the next ray starts on this side of the surface
    if (glass) {
        const float dx = P.ray.dx, dy = P.ray.dy, dz = P.ray.dz;
        const float ior = albedo.w;
        const float eta = flipped ? ior : ieee_rcp(ior);
        const float ci = -dot3(ux, uy, uz, dx, dy, dz);
        const float s2 = (eta * eta) * (1.0f - ci * ci);
        bool reflect = true;
        float ct = 0.f;
        if (s2 < 1.0f) {
            ct = ieee_sqrt(1.0f - s2);
            const float a = eta * ci, bq = eta * ct;
            const float rs = (a - ct) / (a + ct), rp = (ci - bq) / (ci + bq);
            const float F = 0.5f * (rs * rs + rp * rp);
            unsigned r0, r1;
            const unsigned gpix = (P.gxy >> 16) * (unsigned)p.width + (P.gxy & 0xFFFFu);
            philox2x32_10(gpix, ((unsigned)P.s << 8) | ((unsigned)P.b & 0xFFu), p.key, r0, r1);
            const float u1 = (float)(r0 >> 8) * 5.9604644775390625e-08f;
            reflect = u1 < F;
        }
        float tx, ty, tz;
        if (reflect) {
            const float k2 = 2.0f * ci;
            wox = dx + k2 * ux;
            woy = dy + k2 * uy;
            woz = dz + k2 * uz;
            tx = albedo.x; ty = albedo.y; tz = albedo.z;
        } else {
            const float k = eta * ci - ct;
            wox = eta * dx + k * ux;
            woy = eta * dy + k * uy;
            woz = eta * dz + k * uz;
            const float4 tr = mat_f4(M, 13);
            tx = tr.x; ty = tr.y; tz = tr.z;
            sx = -ux; sy = -uy; sz = -uz;
        }
        P.bx = P.bx * tx;
        P.by = P.by * ty;
        P.bz = P.bz * tz;
    } else if (mirror) {
        const float k2 = 2.0f * dot3(ux, uy, uz, P.ray.dx, P.ray.dy, P.ray.dz);
        wox = P.ray.dx - k2 * ux;
        woy = P.ray.dy - k2 * uy;
        woz = P.ray.dz - k2 * uz;
    } else {
        unsigned r0, r1;
        const unsigned gpix = (P.gxy >> 16) * (unsigned)p.width + (P.gxy & 0xFFFFu);
        philox2x32_10(gpix, ((unsigned)P.s << 8) | ((unsigned)P.b & 0xFFu), p.key, r0, r1);
        const float u1 = (float)(r0 >> 8) * 5.9604644775390625e-08f;
        float wlx, wly, wlz;
        cosine_sample(u1, r1 >> 8, wlx, wly, wlz);
        // orthonormal basis (Duff et al. 2017)
        const float sign = copysignf(1.0f, uz);
        const float aa = -ieee_rcp(sign + uz); // -1 / x == -(1 / x)
        const float bb = (ux * uy) * aa;
        const float t0 = 1.0f + ((sign * ux) * ux) * aa, t1 = sign * bb, t2 = -sign * ux;
        const float s0 = bb, s1 = sign + (uy * uy) * aa, s2 = -uy;
        wox = (t0 * wlx + s0 * wly) + ux * wlz;
        woy = (t1 * wlx + s1 * wly) + uy * wlz;
        woz = (t2 * wlx + s2 * wly) + uz * wlz;
    }
    P.ray.ox = best.px + sx * kRayEps;
    P.ray.oy = best.py + sy * kRayEps;
    P.ray.oz = best.pz + sz * kRayEps;
    P.ray.dx = wox; // unit local direction in an orthonormal basis: used as is (|wo| = 1 +- 1e-6)
    P.ray.dy = woy;
    P.ray.dz = woz;
    ++P.b;
}

// Both steps in a row (the brute-force kernel).  Returns true when the lane still owns its pixel (either the path continues with
// a new ray in P.ray, or the next sample's primary ray was generated), false when the pixel is finished.
template <bool SPECULAR = true>
__device__ __forceinline__ bool shade_and_advance(const KParams& p, const Best& best, bool hit, const MaterialRef& M, Path& P)
{
    const int r = settle_hit<SPECULAR>(p, best, hit, M, P);
    if (r == kGoesOn) scatter<SPECULAR>(p, best, M, P);
    return r != kPixelDone;
}

__device__ __forceinline__ void flush_counters(const KParams& p, int lane, const Counters& cnt, bool stats)
{
    // (the pre-pass of a frame traces primary rays that are not path segments of the frame: only a tripped loop guard is reported)
    if (p.shade_mode == kShadePrimaryPass) {
        if (cnt.guard_hits != 0ull && lane == 0) atomicAdd(&p.counters[0], (unsigned long long)__popcll(cnt.guard_hits));
        return;
    }
    // wave-reduced counters, one atomic per wave and counter
    const unsigned long long rays = wave_sum((unsigned long long)cnt.rays);
    // (spread over kRaySlots addresses 128 bytes apart: thousands of waves end within microseconds of each other in a short
    // launch, and atomics on one address are served one after the other, ~10 ns each; the host adds the slots)
    if (lane == 0 && rays) atomicAdd(&p.counters[kRaySlotStride * (kRaySlotFirst + (blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) % kRaySlots)], rays);
    if (lane == 0 && cnt.reused) atomicAdd(&p.counters[kAnsweredWord + kRaySlotStride * ((blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) % kRaySlots)], (unsigned long long)cnt.reused);
    if (lane == 0 && cnt.cut) atomicAdd(&p.counters[kCutShortWord + kRaySlotStride * ((blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) % kRaySlots)], (unsigned long long)cnt.cut);
    if (cnt.guard_hits != 0ull && lane == 0) atomicAdd(&p.counters[0], (unsigned long long)__popcll(cnt.guard_hits)); // (never in a healthy launch)
    if (stats) {
        const unsigned long long n = wave_sum((unsigned long long)cnt.nodes), t = wave_sum((unsigned long long)cnt.tris),
                                 pl = wave_sum((unsigned long long)cnt.planes);
        const unsigned long long r0 = wave_sum((unsigned long long)cnt.inner_rounds), r1 = wave_sum((unsigned long long)cnt.leaf_rounds),
                                 r2 = wave_sum((unsigned long long)cnt.tri_rounds), r3 = wave_sum((unsigned long long)cnt.plane_rounds),
                                 r4 = wave_sum((unsigned long long)cnt.segment_rounds), r5 = wave_sum((unsigned long long)cnt.no_mesh),
                                 r6 = wave_sum((unsigned long long)cnt.plane_exact), r7 = wave_sum((unsigned long long)cnt.stack_overflow),
                                 r8 = wave_sum((unsigned long long)cnt.wall_rounds);
        if (lane == 0) {
            if (n) atomicAdd(&p.counters[1], n);
            if (t) atomicAdd(&p.counters[2], t);
            if (pl) atomicAdd(&p.counters[3], pl);
            atomicAdd(&p.counters[8], r0);
            atomicAdd(&p.counters[9], r1);
            atomicAdd(&p.counters[10], r2);
            atomicAdd(&p.counters[11], r3);
            atomicAdd(&p.counters[12], r4);
            atomicAdd(&p.counters[14], r5);
            atomicAdd(&p.counters[15], r6);
            atomicAdd(&p.counters[26], r7);
            atomicAdd(&p.counters[31], r8);
        }
        {
            const unsigned long long u0 = wave_sum(cnt.t_start), u1 = wave_sum(cnt.t_inner), u2 = wave_sum(cnt.t_leaf);
            if (lane == 0) { atomicAdd(&p.counters[1 + 15], u0); atomicAdd(&p.counters[2 + 15], u1); atomicAdd(&p.counters[3 + 15], u2); }
            const unsigned long long w0 = wave_sum(cnt.t_b1), w1 = wave_sum(cnt.t_b2), w2 = wave_sum(cnt.t_b3);
            if (lane == 0) { atomicAdd(&p.counters[19], w0); atomicAdd(&p.counters[20], w1); atomicAdd(&p.counters[21], w2); }
            const unsigned long long l0 = wave_sum(cnt.t_l1), l1 = wave_sum(cnt.t_l2), l2 = wave_sum(cnt.t_l3);
            if (lane == 0) { atomicAdd(&p.counters[28], l0); atomicAdd(&p.counters[29], l1); atomicAdd(&p.counters[30], l2); }
        }
    }
}

__device__ __forceinline__ void init_path(Path& P)
{
    P.item = 0; P.send = 0; P.gxy = 0; P.s = 0; P.b = 0;
    P.pdx = P.pdy = 0.f; P.pdz = 1.f;
    P.ray = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    P.bx = P.by = P.bz = 1.f;
    P.ax = P.ay = P.az = 0.f;
}

// ---- the BVH mega-kernel -------------------------------------------------------------------------------------------------
//
// Time-sliced: every lane is a small state machine (no query / query in flight / query finished).  One iteration of the main
// loop lets the lanes whose query is finished - or that have none - resolve, shade, fetch work and start their next query
// TOGETHER (the setup block is large: it only pays at good occupancy), then every lane with a query in flight traverses for
// one slice of `setup_threshold` inner-node rounds (traverse_budget).  Lanes whose query outlives the slice keep their
// traversal state in registers and in their LDS stack and simply continue in the next iteration: per-ray traversal cost is
// heavy-tailed, and run to completion the inner-node phase had 11 % of its lanes busy.  Latency is hidden by occupancy: 1024
// threads per workgroup = 4 waves per SIMD.
//
// One iteration, in order (each a pass the whole wave walks through, whatever the number of lanes in it - which is why the
// expensive ones run once per iteration and the cheap one as often as needed):
//   finish_segment  exact evaluation of the finished queries' winners (kernel.cu:110-125)
//   settle_hit      does the path end here?  radiance, next sample, end of block; repeated for the lanes whose next sample starts
//                   from the block's parked primary hit (every sample of a pixel starts with the same ray)
//   scatter         normal, random numbers, next ray: once, for paths that go on from a traced hit and from a parked hit alike
//   acquire_pixel   new (pixel, sample block) items for the lanes without work
//   begin_segment   walls (wall table), other planes / spheres, candidate meshes; a last-bounce query that holds no emitter ends here
//   traverse_budget one time slice of the 4-wide trees (a lane enters its next candidate mesh at the top of a slice)

// EXTRAS = false is the instantiation for scenes made of what the reference itself renders (planes and meshes, diffuse
// and emitting surfaces): the plane/sphere boundary becomes a compile-time "never" and the MIRROR / GLASS branches of the
// shader drop out (together they cost the reference-like scenes 3.5 % otherwise, measured on one box).
// BIG = 1 / 2 (with EXTRAS) are the instantiations for scenes of more than 32 geometries: candidates found by walking the tree
// over the geometries (enter_top / geom_step) instead of scanning all records; 2: records read from global memory.
// PREPASS = true is the instantiation that traces every pixel's primary ray once and stores its hit (settle_hit): the same loop on
// one-ray items; a compile-time switch because its store / mask code inside the shading loop would cost the frame's own kernel
// twenty spilled registers.
template <bool STATS, int BLOCK, bool EXTRAS, int BIG = 0, bool PREPASS = false>
__global__ __launch_bounds__(BLOCK) void trace_bvh_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const LdsT<BIG> L = make_lds<BIG>(p.lds_nodes, p.stack_depth, BLOCK, tid, EXTRAS ? p.num_quads : 0x7fffffff, EXTRAS ? p.trinormals : nullptr,
                                      p.geoms, p.top_first, p.top_lds_first, p.top_lds_count, BIG ? p.num_scan : 0, p.stack_spill);
    const uint4* nodes4 = reinterpret_cast<const uint4*>(p.nodes4);
    if (p.debug_lds_words != 0u) {
        for (unsigned i = tid; i < p.debug_lds_words; i += BLOCK) reinterpret_cast<unsigned*>(ff_smem)[i] = p.debug_lds_pattern;
        __syncthreads();
    }
    stage_scene(L, nodes4, p.geoms, p.num_geoms, p.num_planes, tid, BLOCK);

    Counters cnt = {};
    Path P;
    init_path(P);
    WaveQueue Q = make_wave_queue(p);
    Segment S;
    S.best = { kInf, -1, -1 };
    S.pend = { kInf, -1, -1 };
    S.meshes = 0u;
    S.cur = kDone; S.sp = 0; S.mesh = -1; S.resume = 0;
    S.osr = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    S.ix = S.iy = S.iz = S.ox = S.oy = S.oz = 0.f;
    S.scale = 1.f;
    S.tbound = 0.f;
    S.node_base = 0; S.lds_first = 0; S.lds_count = 0; S.tl_sp = 0;
    S.bnx = S.bny = S.bnz = S.bfx = S.bfy = S.bfz = 0;
    bool active = false, exhausted = false, inflight = false; // inflight: S holds a query of this lane (finished or not)
    // Every sample of a pixel starts with the same ray (kernel.cu:200-205: the pixel's corner, no jitter), so its closest hit is the
    // same hit spp times over.  A pre-pass of the frame (this kernel with shade_mode kShadePrimaryPass: one item per pixel, the hit
    // stored by settle_hit) traces it ONCE per pixel; here every sample starts from the stored hit: no query, no traversal, no
    // resolution for the primary segment - one fifth of the headline frame's path segments.  (Rounds 3 kept the hit per lane and sample
    // block: one traced primary ray per 64 samples and 1.6 GB of parked hits written per 1080p frame.)
    const bool reuse = !PREPASS && p.primary_hits != nullptr; // wave-uniform
    // (the address is formed where it is used - a few instructions - rather than held in registers through the loop)
    auto stored_hit = [&](int k) { return p.primary_hits + (size_t)k * p.pix_items + ((unsigned)P.item & kItemPixelMask); };
    // instrumented launches only: wave cycles per phase (s_memtime), [0] resolve [1] shade [2] acquire [3] begin [4] traverse
    unsigned long long tphase[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long wave_t0 = STATS ? wall_clock64() : 0ull;
    unsigned long long epoch = 0ull;
    if (STATS && p.timeline) {
        if (lane == 0) {
            const unsigned long long seen = atomicCAS(&p.counters[27], 0ull, wave_t0);
            epoch = seen ? seen : wave_t0;
        }
        epoch = __shfl(epoch, 0);
    }
    unsigned* const tl_row = STATS && p.timeline ? p.timeline + (size_t)(blockIdx.x * (BLOCK / kWave) + tid / kWave) * kTimelineBuckets : nullptr;
    int tl_bucket = 0;
    unsigned tl_count = 0u;
    for (;;) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (STATS) t0 = __builtin_amdgcn_s_memtime();
        // Lanes whose query is finished (or that have none) resolve + shade + spawn together; lanes still traversing skip.
        const bool setup = !inflight || segment_done(S);
        if (STATS && p.timeline) {
            // every wave keeps its own row of the histogram (plain stores when the bucket changes: atomics on shared buckets
            // would throttle the launch they are meant to observe); the host adds the rows
            const int finished = __popcll(__ballot(setup && inflight));
            if (finished) {
                const unsigned long long now = wall_clock64();
                const unsigned long long bb = now > epoch ? (now - epoch) / p.timeline_ticks : 0ull;
                const int b = bb < (unsigned long long)kTimelineBuckets ? (int)bb : kTimelineBuckets - 1;
                if (b != tl_bucket) {
                    if (lane == 0 && tl_count) tl_row[tl_bucket] += tl_count;
                    tl_bucket = b;
                    tl_count = 0u;
                }
                tl_count += (unsigned)finished;
            }
        }
        Best best;
        bool hit = false;
        bool shade_now = setup && inflight; // lanes with a finished query
        if (shade_now) {
            finish_segment(L, p.tris, P.ray, S, best);
            hit = best.geom >= 0;
        }
        // A lane that waits with a new sample (it starts from the pixel's stored primary hit, see below) joins this iteration's shading.
        bool from_cache = setup && !inflight && active && P.b == 0 && reuse;
        if (STATS) t1 = __builtin_amdgcn_s_memtime();
        // Shading in two steps (settle_hit / scatter).  A lane whose path ended and whose next sample starts with the parked hit
        // settles again - at once if at least reuse_quorum lanes of the wave are in that position, else together with the next
        // iteration's finished queries - until every settling lane has a path that goes on from a hit, a first-of-block primary ray
        // to trace, a parked hit to wait with, or no work left.  Settling is cheap (a material lookup, a few multiplications); the
        // expensive step - normal, random numbers, new direction - then runs ONCE, for the lanes that go on from the hit they just
        // found and for those that go on from their parked primary hit alike.
        bool settle_now = shade_now, goes_on = false;
        for (;;) { // (every pass ends a sample of each lane in it: at most a block's samples)
            if (from_cache) {
                const float4 c0 = *stored_hit(0), c1 = *stored_hit(1), c2 = *stored_hit(2);
                best.dist = c0.x; best.px = c0.y; best.py = c0.z; best.pz = c0.w;
                best.cx = c1.x; best.cy = c1.y; best.cz = c1.z;
                best.geom = __float_as_int(c1.w);
                best.rec = __float_as_int(c2.x);
                hit = best.geom >= 0;
                cnt.rays += 1; // a path segment like any other, answered without a traversal (counted apart below)
                settle_now = true;
            }
            {
                const unsigned reused_now = (unsigned)__popcll(__ballot(from_cache));
                cnt.reused += reused_now;
                if (STATS && p.timeline) tl_count += reused_now; // (the launch timeline counts every path segment where it completes)
            }
            if (__ballot(settle_now) == 0ull) break;
            bool waiting = false;
            if (settle_now) {
                MaterialRef M;
                M.global = BIG == 2 ? p.geoms + (hit ? best.geom : 0) : nullptr;
                M.geom_base = L.geom_base;
                M.g = best.geom;
                const int r = settle_hit<EXTRAS, PREPASS>(p, best, hit, M, P);
                inflight = false;
                active = r != kPixelDone;
                goes_on = r == kGoesOn;
                waiting = r == kNewSample && reuse; // a new sample: it starts from the pixel's stored primary hit
            }
            settle_now = false;
            from_cache = waiting && __popcll(__ballot(waiting)) >= p.reuse_quorum;
        }
        if (goes_on) {
            MaterialRef M;
            M.global = BIG == 2 ? p.geoms + best.geom : nullptr;
            M.geom_base = L.geom_base;
            M.g = best.geom;
            scatter<EXTRAS>(p, best, M, P);
        }
        if (STATS) t2 = __builtin_amdgcn_s_memtime();
        {
            const bool need = setup && !active && !exhausted;
            if (__ballot(need) != 0ull) {
                const unsigned rays_before = cnt.rays;
                const bool got = acquire_pixel(p, lane, P, Q, need, cnt.rays); // (all lanes call: the wave's chunk of the queue is wave state)
                if (STATS && p.timeline) tl_count += (unsigned)wave_sum((unsigned long long)(cnt.rays - rays_before)); // (the rays of items dropped at the queue)
                if (need) {
                    active = got;
                    exhausted = !got;
                    if (STATS && exhausted) atomicMax(&p.counters[23], ~(unsigned long long)wall_clock64()); // (complemented) first lane to find the queue empty
                }
            }
        }
        if (STATS) t3 = __builtin_amdgcn_s_memtime();
        bool over = false; // a last-bounce query that ended after the planes
        if (setup && active && !(P.b == 0 && reuse)) { // (a lane whose sample starts from the stored primary hit waits for the next shading pass)
            if (STATS) probe_round(cnt.segment_rounds);
            over = begin_segment<STATS>(L, p.walls, p.geoms, p.num_geoms, p.num_planes, p.tris, P.ray, S, cnt, p.cut_last != 0 && P.b == p.bounces - 1, p.emitter_mask);
            cnt.rays += 1;
            inflight = true;
        }
        cnt.cut += (unsigned)__popcll(__ballot(over));
        if (STATS) t4 = __builtin_amdgcn_s_memtime();
        if (__ballot(inflight || (active && P.b == 0 && reuse)) == 0ull) break; // (a lane that waits with a stored hit still has work)
        // Time-sliced traversal: after `setup_threshold` inner-node rounds the finished lanes go and fetch new rays while the
        // long-tail lanes keep their state (per-ray traversal cost is heavy-tailed: a few rays need 10x the mean).
        if (inflight) traverse_budget<STATS>(L, p.tris, nodes4, P.ray, S, cnt, p.setup_threshold, p.leaf_threshold, p.num_planes);
        if (STATS) {
            const unsigned long long t5 = __builtin_amdgcn_s_memtime();
            tphase[0] += t1 - t0; tphase[1] += t2 - t1; tphase[2] += t3 - t2; tphase[3] += t4 - t3; tphase[4] += t5 - t4;
        }
    }
    if (STATS && p.timeline && lane == 0 && tl_count) tl_row[tl_bucket] += tl_count;
    if (STATS && lane == 0) {
        atomicAdd(&p.counters[4], tphase[0]);
        atomicAdd(&p.counters[5], tphase[1]);
        atomicAdd(&p.counters[6], tphase[2]);
        atomicAdd(&p.counters[7], tphase[3]);
        atomicAdd(&p.counters[13], tphase[4]);
        atomicMax(&p.counters[22], tphase[0] + tphase[1] + tphase[2] + tphase[3] + tphase[4]); // slowest wave
        atomicMax(&p.counters[24], ~(unsigned long long)wave_t0); // (complemented) first wave start, 100 MHz wall clock
        atomicMax(&p.counters[25], (unsigned long long)wall_clock64()); // last wave end
    }
    flush_counters(p, lane, cnt, STATS);
}

// ---- the job-pool mega-kernel ------------------------------------------------------------------------------------------------
//
// trace_bvh_kernel keeps a query in the lane that owns its path: when neighbouring rays need very different numbers of node
// visits, the lanes that are through idle until the wave's next setup pass (inner-node phase: a third of the lanes busy).
// Here the traversal of one (ray, mesh) pair is a JOB parked in LDS, and ANY wave of the workgroup executes jobs:
//
//   * a lane still owns its path (the path state never leaves its registers).  In a SETUP pass the lanes of a wave whose query is
//     complete (or that have none) resolve, shade, fetch work and screen the planes of their next ray together, exactly as in
//     trace_bvh_kernel; a lane whose ray can reach a mesh then writes the job - object-space ray, scale, bounds, cursor: 48 bytes
//     in its own slot of the job array - and puts the slot's number in the workgroup's queue (a ring in LDS);
//   * in the TRAVERSE role a wave takes jobs from the queue, 64 at a time, and walks them through the 4-wide trees in slices of a
//     few inner-node rounds; after a slice the lanes whose job is finished hand it back (16 bytes + a flag the owner polls) and,
//     once enough lanes are free, the wave takes as many new jobs: the inner-node phase runs on nearly full waves whatever the
//     spread of work between rays.  The traversal stack of a job lives in LDS under the job's slot, so a job can be put down by
//     one wave (16 bytes written back) and picked up by another;
//   * a wave runs a setup pass when enough of its own lanes are ready for one (pool_quorum) - or when the queue has nothing
//     to offer - and traverses otherwise.
//
// What crosses the hand-over is only what the walk needs; everything exact stays with the owner: a near tie between a triangle
// and the candidate the query holds (offer) ends the job with its leaf under the cursor, the owner resolves the held candidate
// exactly (kernel.cu:110-125) and posts the job again.  Per ray the sequence of box tests, triangle tests, offers and exact
// evaluations is the one trace_bvh_kernel performs: same bits, same ray counts.
//
// Scenes of up to 32 geometries (the reference has five); larger ones run trace_bvh_kernel.

#ifndef FF_POOL_DEBUG
#define FF_POOL_DEBUG 0
#endif
constexpr int kPoolRing = 2048;          // queue entries (16-bit slot numbers): twice the most jobs that can be outstanding, so an
                                         // entry claimed by a consumer is never the one a producer writes
constexpr unsigned kRingEmpty = 0xFFFFu;
constexpr int kPoolSpinLimit = 1 << 20;  // bound on every wait of the pool kernel (a wait that long is a bug: the launch ends with an error instead of hanging)
constexpr int kPendEmpty = -2;           // job's pending tag: the query holds no pending candidate
constexpr int kPendOwner = -3;           // ... holds one whose identity stays with the owner (a plane, a triangle of an earlier mesh)
constexpr unsigned kJobReturned = 0x80000000u; // job word 9 (sp | mesh << 8 | resume << 16): set by the wave that hands the job back
constexpr unsigned kJobTie = 0x40000000u;      // ... because triangle resume - 1 of the leaf under the cursor met a near tie with the pending candidate

// Wave-uniform tallies of the traverse role, kept by every build (scalar registers): how full its rounds run.
struct PoolOccupancy {
    unsigned inner_rounds, inner_lanes; // inner-node rounds of the wave, and the lanes that visited a node in them (= node visits)
    unsigned leaf_rounds, leaf_lanes;   // leaf phases, and the lanes that held a leaf in them
};

struct PoolRef {
    int job_base;  // uint4 index: quarter q of job j at job_base + q * BLOCK + j
    int ring_base; // 16-bit index of ring entry 0
    int ctrl;      // 32-bit index of [head, tail]
};

__device__ __forceinline__ unsigned* pool_u32() { return reinterpret_cast<unsigned*>(ff_smem); }
__device__ __forceinline__ unsigned short* pool_u16() { return reinterpret_cast<unsigned short*>(ff_smem); }

// Put the calling lanes' slots (post) into the queue.  Called by all lanes of the wave.
__device__ __forceinline__ void pool_push(const PoolRef& Q, int lane, int slot, bool post)
{
    const unsigned long long m = __ballot(post);
    if (m == 0ull) return;
    // what the job's slot holds was written by this wave before this point: LDS executes a wave's instructions in order, and
    // the fence keeps the compiler from moving them behind the queue entry
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0u;
    if (lane == leader) base = __hip_atomic_fetch_add(&pool_u32()[Q.ctrl + 1], (unsigned)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    if (post) {
        const unsigned rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        __hip_atomic_store(&pool_u16()[Q.ring_base + (int)((base + rank) & (unsigned)(kPoolRing - 1))], (unsigned short)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Take up to one job per asking lane (want) from the queue: returns the lane's job slot or -1.  Called by all lanes of the wave
// in wave-uniform control flow.
__device__ __forceinline__ int pool_pop(const PoolRef& Q, int lane, bool want, bool& stuck)
{
    const unsigned long long m = __ballot(want);
    const unsigned asked = (unsigned)__popcll(m);
    unsigned h = 0u, n = 0u;
    if (lane == 0) {
        for (int tries = 0; tries < kPoolSpinLimit; ++tries) {
            h = __hip_atomic_load(&pool_u32()[Q.ctrl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned t = __hip_atomic_load(&pool_u32()[Q.ctrl + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            n = min(asked, t - h);
            if (n == 0u) break;
            unsigned expect = h;
            if (__hip_atomic_compare_exchange_strong(&pool_u32()[Q.ctrl], &expect, h + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            n = 0u; // (if the tries run out: nothing taken)
        }
    }
    h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
    n = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
    int slot = -1;
    const unsigned rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (want && rank < n) {
        unsigned short* e = &pool_u16()[Q.ring_base + (int)((h + rank) & (unsigned)(kPoolRing - 1))];
        unsigned v;
        // (the producer reserves its entries with one atomic and writes them straight after: a reserved entry that is still
        // empty is filled within a few instructions of another wave, which nothing here can hold up)
        int spins = 0;
        do { v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (v == kRingEmpty && ++spins < kPoolSpinLimit);
        if (v != kRingEmpty) {
            __hip_atomic_store(e, (unsigned short)kRingEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            slot = (int)v;
        } else {
            stuck = true; // (never in a healthy launch: the caller gives up and the host reports it)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return slot;
}

// Jobs waiting in the queue (a snapshot).
__device__ __forceinline__ unsigned pool_waiting(const PoolRef& Q)
{
    const unsigned h = __hip_atomic_load(&pool_u32()[Q.ctrl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned t = __hip_atomic_load(&pool_u32()[Q.ctrl + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return t - h;
}

// The third quarter of a job: what a traversal changes.  (cursor, sp | mesh << 8 | resume << 16 [| kJobReturned], pending distance, pending tag)
__device__ __forceinline__ uint4 job_state_quarter(const Segment& J, unsigned flags)
{
    return make_uint4((unsigned)J.cur, (unsigned)J.sp | ((unsigned)J.mesh << 8) | ((unsigned)J.resume << 16) | flags, __float_as_uint(J.pend.dist),
                      (unsigned)J.pend.rec);
}

// Write the whole job of the calling lane: the object-space ray and scale of the mesh just entered (enter_mesh), the exact distance
// the query has resolved so far, the state quarter.
template <int BLOCK>
__device__ __forceinline__ void job_write(const PoolRef& Q, int slot, const Segment& J)
{
    ff_smem[Q.job_base + slot] = make_uint4(__float_as_uint(J.osr.ox), __float_as_uint(J.osr.oy), __float_as_uint(J.osr.oz), __float_as_uint(J.scale));
    ff_smem[Q.job_base + BLOCK + slot] = make_uint4(__float_as_uint(J.osr.dx), __float_as_uint(J.osr.dy), __float_as_uint(J.osr.dz), __float_as_uint(J.best.dist));
    ff_smem[Q.job_base + 2 * BLOCK + slot] = job_state_quarter(J, 0u);
}

// Pick a job up: the traversal state of trace_bvh_kernel's Segment, rebuilt from the 48 bytes (slab constants, box planes,
// pruning bound) and the mesh's record.
template <int BLOCK, class LDS>
__device__ __forceinline__ void job_load(const PoolRef& Q, const LDS& L, int slot, Segment& J)
{
    const uint4 a = ff_smem[Q.job_base + slot], b = ff_smem[Q.job_base + BLOCK + slot], c = ff_smem[Q.job_base + 2 * BLOCK + slot];
    J.osr.ox = __uint_as_float(a.x); J.osr.oy = __uint_as_float(a.y); J.osr.oz = __uint_as_float(a.z);
    J.scale = __uint_as_float(a.w);
    J.osr.dx = __uint_as_float(b.x); J.osr.dy = __uint_as_float(b.y); J.osr.dz = __uint_as_float(b.z);
    J.best.dist = __uint_as_float(b.w);
    J.best.geom = -1;
    J.best.rec = -1;
    J.cur = (int)c.x;
    J.sp = (int)(c.y & 0xFFu);
    J.mesh = (int)((c.y >> 8) & 0xFFu);
    J.resume = (int)((c.y >> 16) & 0x3FFFu); // (> 0: the leaf under the cursor continues from triangle resume - 1: leaf_step)
    J.pend.dist = __uint_as_float(c.z);
    J.pend.rec = (int)c.w;
    J.pend.geom = (int)c.w == kPendEmpty ? -1 : J.mesh; // (only its sign matters to offer(); the owner knows whose it is)
    J.meshes = 0u;
    J.tl_sp = 0;
    J.ix = safe_rcp(J.osr.dx);
    J.iy = safe_rcp(J.osr.dy);
    J.iz = safe_rcp(J.osr.dz);
    J.ox = -J.osr.ox * J.ix;
    J.oy = -J.osr.oy * J.iy;
    J.oz = -J.osr.oz * J.iz;
    set_box_planes(L, J);
    refresh_tbound(J);
    const int4 tree = lds_geom_i4(L, J.mesh, 17);
    J.node_base = tree.z;
    J.lds_count = tree.w;
    J.lds_first = __float_as_int(lds_geom4(L, J.mesh, 14).w);
}

// One time slice of the traverse role: inner-node phases and leaf phases alternate wave-wide (traverse_budget without the parts
// that belong to the owner: mesh entry and exact resolution) until `limit` inner rounds have been spent or no lane can go on.
// A lane is busy while its cursor is on a node or a leaf and no near tie is waiting for its owner (`tied`: leaf_step left the leaf
// under the cursor with J.resume set).
template <bool STATS, class LDS>
__device__ __forceinline__ void pool_slice(const LDS& L, const TriRecord* __restrict__ tris, const uint4* __restrict__ nodes4, Segment& J, bool& tied,
                                           Counters& cnt, int limit, int leaf_threshold, PoolOccupancy& occ)
{
    const Ray none = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    int rounds = 0;
    for (int guard = 0; guard < 4 * kWave; ++guard) {
        unsigned long long tb = 0, tc = 0;
        if (STATS) tb = __builtin_amdgcn_s_memtime();
        const bool busy = J.cur != kDone && !tied;
        if (__ballot(busy) == 0ull) break;
        for (;;) {
            const bool inner = !tied && (unsigned)J.cur < (unsigned)kMeshDone;
            if (__ballot(inner) == 0ull) break;
            const bool tri_leaf = !tied && J.cur < 0;
            if (__popcll(__ballot(tri_leaf)) >= leaf_threshold) break;
            if (rounds >= limit) break;
            ++rounds;
            if (!STATS) { occ.inner_rounds += 1u; occ.inner_lanes += (unsigned)__popcll(__ballot(inner)); }
            if (inner) inner_step<STATS>(L, nodes4, J, cnt);
        }
        if (STATS) tc = __builtin_amdgcn_s_memtime();
        if (!STATS) {
            const unsigned at_leaf = (unsigned)__popcll(__ballot(!tied && J.cur < 0));
            occ.leaf_rounds += at_leaf ? 1u : 0u;
            occ.leaf_lanes += at_leaf;
        }
        if (!tied && J.cur < 0) {
            leaf_step<STATS>(L, tris, nodes4, none, J, cnt);
            tied = J.resume > 0;
        }
        if (STATS) {
            const unsigned long long td = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { cnt.t_inner += tc - tb; cnt.t_leaf += td - tc; }
        }
        if (rounds >= limit) break;
    }
}

// The traverse role of trace_pool_kernel: take jobs from the workgroup's queue, walk them in slices, hand finished ones back, take
// new ones as lanes fall free; leave when enough of the wave's own lanes are ready for a setup pass (`waiting`: this lane has a job
// out; `nojob_ready`: it is ready without one) or when there is nothing to walk.  Returns true if a wait ran into the watchdog.
//
// A function of its own, NOT inlined, on purpose.  Walking needs every register a wave has at four waves per SIMD (128); inlined,
// the register allocator treats the kernel's two bodies - setup pass and traversal - as one, holds three dozen values of the one
// through the other and spills inside the loops of both (first build: 154 spills, the frame 41 % SLOWER than the lane-owned
// kernel).  Behind a call the walk gets an allocation of its own; what the kernel holds across the call is a few flags.  Its
// wave-uniform inputs come through a pointer to the kernel arguments and are made scalar on entry.
template <bool STATS, int BLOCK>
__device__ __attribute__((noinline)) bool pool_role(int job_base_in, int tid, bool waiting, bool nojob_ready, Counters* stats)
{
    auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni_ptr = [&](unsigned lo, unsigned hi) { return (void*)(((unsigned long long)uni(hi) << 32) | uni(lo)); };
    const int lane = tid & (kWave - 1);
    PoolRef Q;
    Q.job_base = (int)uni((unsigned)job_base_in);
    Q.ring_base = (Q.job_base + 3 * BLOCK) * 8;
    Q.ctrl = (Q.job_base + 3 * BLOCK) * 4 + kPoolRing / 2;
    // (the kernel left the role's wave-uniform inputs behind the queue's counters: taking the address of the kernel arguments
    // instead would make the compiler copy all 2 KB of them into every lane's scratch memory)
    const uint4 c0 = ff_smem[Q.ctrl / 4 + 1], c1 = ff_smem[Q.ctrl / 4 + 2], c2 = ff_smem[Q.ctrl / 4 + 3];
    const TriRecord* tris = static_cast<const TriRecord*>(uni_ptr(c0.x, c0.y));
    const uint4* nodes4 = static_cast<const uint4*>(uni_ptr(c0.z, c0.w));
    int* spill = static_cast<int*>(uni_ptr(c1.x, c1.y));
    const int slice = (int)uni(c1.z), leaf_threshold = (int)uni(c1.w), refill = (int)uni(c2.x), leave = (int)uni(c2.y), quorum = (int)uni(c2.z);
    const unsigned layout = uni(c2.w); // lds_nodes | stack_depth << 24
    const LdsT<0> L = make_lds<0>((int)(layout & 0xFFFFFFu), (int)(layout >> 24), BLOCK, tid, 0x7fffffff, nullptr, nullptr, 0, 0, 0, 0, spill);
    Counters cnt = {};
    PoolOccupancy occ = { 0u, 0u, 0u, 0u };
    bool stuck = false;
    Segment J;
    J.cur = kDone;
    J.resume = 0;
    J.sp = 0;
    J.mesh = 0;
    J.pend = { kInf, -1, kPendEmpty };
    J.best = { kInf, -1, -1 };
    int jslot = -1;
    bool tied = false;
    LdsT<0> Lj = L;
    bool leaving = false; // enough of this wave's own lanes are ready for a setup pass: no new jobs, the held ones run on for a while
    for (int turns = 0;; ++turns) {
        if (turns > kPoolSpinLimit) { stuck = true; break; } // (a job that never ends: a malformed tree)
        // free lanes take new jobs once there are enough of them (one atomic for all)
        const int held = __popcll(__ballot(jslot >= 0));
        if (leaving && held <= leave) {
            // put the stragglers down (16 bytes each; another wave picks them up together with other waves' stragglers) and go
            const bool down = jslot >= 0;
            if (down) ff_smem[Q.job_base + 2 * BLOCK + jslot] = job_state_quarter(J, 0u);
            pool_push(Q, lane, jslot, down);
            break;
        }
        if (!leaving && (kWave - held >= refill || held == 0) && (held == 0 || uni(pool_waiting(Q)) != 0u)) {
            const int got = pool_pop(Q, lane, jslot < 0, stuck);
            if (got >= 0) {
                jslot = got;
                Lj.stack_base = L.node_cap * (kNodeVec4 * 4) + got;
                Lj.stack_slot = got;
                job_load<BLOCK>(Q, Lj, got, J);
            }
        }
        if (__ballot(stuck) != 0ull) { stuck = true; break; }
        if (__ballot(jslot >= 0) == 0ull) break; // nothing held, nothing queued
        pool_slice<STATS>(Lj, tris, nodes4, J, tied, cnt, slice, leaf_threshold, occ);
        // finished jobs (and jobs that met a near tie) go back to their owners
        const bool back = jslot >= 0 && (J.cur == kDone || tied);
        if (__ballot(back) != 0ull) {
            if (back) {
                // (the state first, then - in an instruction of its own - the flag its owner polls)
                const uint4 st = job_state_quarter(J, tied ? kJobTie : 0u);
                ff_smem[Q.job_base + 2 * BLOCK + jslot] = st;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&pool_u32()[(Q.job_base + 2 * BLOCK + jslot) * 4 + 1], st.y | kJobReturned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                jslot = -1;
                tied = false;
                J.cur = kDone;
                J.resume = 0;
            }
        }
        // enough of this wave's own lanes ready for a setup pass?
        if (!leaving) {
            bool ret2 = false;
            if (waiting) ret2 = (__hip_atomic_load(&pool_u32()[(Q.job_base + 2 * BLOCK + tid) * 4 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & kJobReturned) != 0u;
            leaving = __popcll(__ballot(ret2 || nojob_ready)) >= quorum;
        }
    }
    if (!STATS && lane == 0) {
        // (the workgroup's tallies sit behind the role's inputs; thread 0 adds them to the launch's counters when the workgroup ends)
        unsigned long long* t = reinterpret_cast<unsigned long long*>(&pool_u32()[Q.ctrl + 16]);
        __hip_atomic_fetch_add(&t[0], (unsigned long long)occ.inner_rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&t[1], (unsigned long long)occ.inner_lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&t[2], (unsigned long long)occ.leaf_rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&t[3], (unsigned long long)occ.leaf_lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (STATS && stats != nullptr) {
        stats->nodes += cnt.nodes; stats->tris += cnt.tris;
        stats->inner_rounds += cnt.inner_rounds; stats->leaf_rounds += cnt.leaf_rounds; stats->tri_rounds += cnt.tri_rounds;
        stats->stack_overflow += cnt.stack_overflow;
        stats->t_inner += cnt.t_inner; stats->t_leaf += cnt.t_leaf;
        stats->t_l1 += cnt.t_l1; stats->t_l2 += cnt.t_l2; stats->t_l3 += cnt.t_l3;
    }
    return __ballot(stuck) != 0ull;
}

template <bool STATS, int BLOCK, bool EXTRAS>
__global__ __launch_bounds__(BLOCK) void trace_pool_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const LdsT<0> L = make_lds<0>(p.lds_nodes, p.stack_depth, BLOCK, tid, EXTRAS ? p.num_quads : 0x7fffffff, EXTRAS ? p.trinormals : nullptr, p.geoms, 0, 0, 0,
                                  0, p.stack_spill);
    const uint4* nodes4 = reinterpret_cast<const uint4*>(p.nodes4);
    if (p.debug_lds_words != 0u) {
        for (unsigned i = tid; i < p.debug_lds_words; i += BLOCK) reinterpret_cast<unsigned*>(ff_smem)[i] = p.debug_lds_pattern;
        __syncthreads();
    }
    // the pool sits behind the geometry records: [jobs: 3 quarters x BLOCK x 16 B][ring: kPoolRing x 2 B][head, tail, -, -]
    PoolRef Q;
    Q.job_base = L.geom_base + p.num_geoms * kGeomVec4;
    Q.ring_base = (Q.job_base + 3 * BLOCK) * 8;
    Q.ctrl = (Q.job_base + 3 * BLOCK) * 4 + kPoolRing / 2;
    for (int i = tid; i < kPoolRing / 2; i += BLOCK) pool_u32()[Q.ring_base / 2 + i] = 0xFFFFFFFFu;
    if (tid < 4) pool_u32()[Q.ctrl + tid] = 0u;
    if (tid < 16) pool_u32()[Q.ctrl + 16 + tid] = 0u; // the workgroup's tallies (pool_role, and the waves' time split below)
    if (tid == 0) {
        // the traverse role's wave-uniform inputs (pool_role reads them back from here)
        const unsigned long long a = (unsigned long long)p.tris, b = (unsigned long long)p.nodes4, c = (unsigned long long)p.stack_spill;
        ff_smem[Q.ctrl / 4 + 1] = make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
        ff_smem[Q.ctrl / 4 + 2] = make_uint4((unsigned)c, (unsigned)(c >> 32), (unsigned)p.pool_slice, (unsigned)p.leaf_threshold);
        ff_smem[Q.ctrl / 4 + 3] = make_uint4((unsigned)p.pool_refill, (unsigned)p.pool_leave, (unsigned)p.pool_quorum, (unsigned)p.lds_nodes | ((unsigned)p.stack_depth << 24));
    }
    ff_smem[Q.job_base + 2 * BLOCK + tid] = make_uint4((unsigned)kDone, 0u, 0u, (unsigned)kPendEmpty);
    stage_scene(L, nodes4, p.geoms, p.num_geoms, p.num_planes, tid, BLOCK); // (ends with the workgroup's barrier)

    Counters cnt = {};
    WaveQueue WQ = make_wave_queue(p);
    // The lane's own path (Path) and what it keeps of its query while a job is out (the exactly resolved winner so far, the pending
    // candidate, the meshes to come; its ray count) live in the lane's slot of a global array (KParams::park, 7 x 16 bytes, lane-
    // strided) BETWEEN setup passes: a pass loads them, works, and stores them back.  While the lane walks other lanes' jobs they
    // would be 30 registers of dead weight, and walking needs every register the wave has (four waves per SIMD: 128); carried
    // through the loop they were spilled and re-loaded piecemeal in every phase (first build: 154 spills, -41 %).  What IS carried
    // from iteration to iteration is a handful of flags.
    // (`me` is the thread's index behind an optimisation barrier, taken afresh in every pass: formed from `tid` the seven slot
    // addresses - and a dozen others - are loop invariants that the compiler computes once and then holds in registers through
    // BOTH roles: 80 VGPRs that neither could spare)
    auto park_slot = [&](int k, int me) { return p.park + ((size_t)k * gridDim.x + blockIdx.x) * (size_t)BLOCK + (size_t)me; };
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto park_store = [&](int me, const Path& P, const BestId& obest, const Pending& opend, unsigned omeshes, unsigned rays) {
        *park_slot(0, me) = make_float4(__int_as_float(P.item), __int_as_float(P.send), __uint_as_float(P.gxy), __int_as_float(P.s));
        *park_slot(1, me) = make_float4(__int_as_float(P.b), P.pdx, P.pdy, P.pdz);
        *park_slot(2, me) = make_float4(P.ray.ox, P.ray.oy, P.ray.oz, P.ray.dx);
        *park_slot(3, me) = make_float4(P.ray.dy, P.ray.dz, P.bx, P.by);
        *park_slot(4, me) = make_float4(P.bz, P.ax, P.ay, P.az);
        *park_slot(5, me) = make_float4(obest.dist, __int_as_float(obest.geom), __int_as_float(obest.rec), opend.dist);
        *park_slot(6, me) = make_float4(__int_as_float(opend.geom), __int_as_float(opend.rec), __uint_as_float(omeshes), __uint_as_float(rays));
    };
    {
        Path P0;
        init_path(P0);
        park_store(tid, P0, BestId{ kInf, -1, -1 }, Pending{ kInf, -1, -1 }, 0u, 0u);
    }
    bool more_meshes = false; // the lane's query has candidate meshes left (omeshes != 0)
    bool active = false, exhausted = false, inflight = false; // inflight: the lane has a query (begun, not yet shaded)
    bool waiting = false;                                     // ... and a job of it is out (queued, being walked, or handed back and not yet read)
    const bool reuse = p.primary_hits != nullptr; // every sample starts from its pixel's stored primary hit (trace_bvh_kernel)
    unsigned long long tphase[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long wave_t0 = STATS ? wall_clock64() : 0ull;
    const int quorum = p.pool_quorum, qmin = p.pool_quorum_min;
    int starved = 0; // consecutive polls that found nothing to do
    // Watchdog: every loop of this kernel is bounded.  A wave that sleeps kPoolSpinLimit times in a row, or waits that long for a
    // queue entry, gives up: it reports through the guard counter (the host turns it into an error, as for the traversal loop
    // guard) and leaves, so a scheduling bug is a failed frame, not a hung GPU.
    bool stuck = false;
    int idle_polls = 0;
    // where the wave's time goes (every build: three stamps per turn of the loop, scalar): setup passes, the traverse role, the rest
    // (deciding, waiting); raw counters [16] [17] [18] and the number of setup passes / role visits in [11] [12]
    unsigned long long tw_setup = 0ull, tw_role = 0ull, tw_rest = 0ull, tw_mark = __builtin_amdgcn_s_memtime();
    unsigned n_setup = 0u, n_role = 0u;
    for (unsigned turns = 0;; ++turns) {
        if (stuck || idle_polls > kPoolSpinLimit || (FF_POOL_DEBUG && turns > (1u << 16))) {
            cnt.guard_hits |= 1ull;
            if (FF_POOL_DEBUG) {
                const unsigned long long bw = __ballot(waiting), bi = __ballot(inflight), ba = __ballot(active), be = __ballot(exhausted);
                if (lane == 0)
                    printf("pool watchdog: block %d wave %d stuck %d idle %d turns %u waiting %llx inflight %llx active %llx exhausted %llx head %u tail %u\n", (int)blockIdx.x,
                           tid / kWave, (int)stuck, idle_polls, turns, bw, bi, ba, be, pool_u32()[Q.ctrl], pool_u32()[Q.ctrl + 1]);
            }
            break;
        }
        // ---- what can this wave do? ----
        const int me_d = opaque(tid);
        bool returned = false;
        if (waiting) returned = (__hip_atomic_load(&pool_u32()[(Q.job_base + 2 * BLOCK + me_d) * 4 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & kJobReturned) != 0u;
        const bool mine = returned || (!waiting && (inflight || active || !exhausted)); // lanes a setup pass would serve
        const int nready = __popcll(__ballot(mine));
        const unsigned queued = (unsigned)__builtin_amdgcn_readfirstlane((int)pool_waiting(Q));
        const bool any_waiting = __ballot(waiting) != 0ull;
        // What to do next.  A setup pass when enough lanes are ready for one; else a visit to the queue when it holds a batch worth
        // taking (pool_batch_min jobs: a wave that takes a handful walks them on a handful of lanes - and the inner-node phase is
        // back where the lane-owned kernel had it); else a setup pass for a smaller company (pool_quorum_min); else wait a moment
        // for the other waves to post or hand back - and after a few empty looks take whatever there is, ready lanes first, so that
        // nothing waits for ever.
        bool do_setup = nready >= quorum || (nready > 0 && !any_waiting && queued == 0u);
        bool do_role = !do_setup && queued >= (unsigned)p.pool_batch_min;
        if (!do_setup && !do_role) {
            if (nready == 0 && !any_waiting && queued == 0u) break; // nothing left for this wave: no work, no query, no job out, nothing to walk
            if (nready >= qmin || (nready > 0 && starved >= 4)) do_setup = true;
            else if (queued > 0u && starved >= 4) do_role = true;
            else {
                ++starved;
                ++idle_polls;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
        }
        starved = 0;
        idle_polls = 0;
#ifdef FF_EXP_NO_SETUP
        if (do_setup) { active = false; exhausted = true; waiting = false; continue; }
#endif
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tw_rest += now - tw_mark;
            tw_mark = now;
        }
        if (do_setup) {
            unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
            if (STATS) t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int me = opaque(tid);
            Path P;
            BestId obest;
            Pending opend;
            unsigned omeshes;
            {
                const float4 a0 = *park_slot(0, me), a1 = *park_slot(1, me), a2 = *park_slot(2, me), a3 = *park_slot(3, me), a4 = *park_slot(4, me), a5 = *park_slot(5, me),
                             a6 = *park_slot(6, me);
                P.item = __float_as_int(a0.x); P.send = __float_as_int(a0.y); P.gxy = __float_as_uint(a0.z); P.s = __float_as_int(a0.w);
                P.b = __float_as_int(a1.x); P.pdx = a1.y; P.pdy = a1.z; P.pdz = a1.w;
                P.ray.ox = a2.x; P.ray.oy = a2.y; P.ray.oz = a2.z; P.ray.dx = a2.w;
                P.ray.dy = a3.x; P.ray.dz = a3.y; P.bx = a3.z; P.by = a3.w;
                P.bz = a4.x; P.ax = a4.y; P.ay = a4.z; P.az = a4.w;
                obest.dist = a5.x; obest.geom = __float_as_int(a5.y); obest.rec = __float_as_int(a5.z); opend.dist = a5.w;
                opend.geom = __float_as_int(a6.x); opend.rec = __float_as_int(a6.y); omeshes = __float_as_uint(a6.z); cnt.rays = __float_as_uint(a6.w);
            }
            // -- jobs that came back: merge what they found; a near tie is resolved exactly and the job goes out again --
            bool repost = false;
            if (returned) {
                const uint4 c = ff_smem[Q.job_base + 2 * BLOCK + me];
                const int mesh = (int)((c.y >> 8) & 0xFFu);
                if ((int)c.w >= 0) {
                    opend.dist = __uint_as_float(c.z);
                    opend.geom = mesh;
                    opend.rec = (int)c.w;
                }
                waiting = false;
                repost = (c.y & kJobTie) != 0u;
            }
            if (__ballot(repost) != 0ull) {
                if (repost) {
                    HitPoint H;
                    resolve_pending(L, p.tris, P.ray, opend, obest, H);
                    // (the job's ray, cursor, stack height and leaf position are where the walk left them; the pending slot is empty now)
                    pool_u32()[(Q.job_base + BLOCK + me) * 4 + 3] = __float_as_uint(obest.dist);
                    unsigned* st = &pool_u32()[(Q.job_base + 2 * BLOCK + me) * 4];
                    st[1] = st[1] & ~(kJobReturned | kJobTie);
                    st[2] = __float_as_uint(kInf);
                    st[3] = (unsigned)kPendEmpty;
                    waiting = true;
                }
            }
            // Lanes whose query is complete (or that have none) resolve + shade + spawn together.
            const bool setup = mine && !waiting && !(inflight && omeshes != 0u);
            (void)more_meshes;
            Best best;
            bool hit = false;
            bool shade_now = setup && inflight;
            if (shade_now) {
                Segment S;
                S.best = obest;
                S.pend = opend;
                finish_segment(L, p.tris, P.ray, S, best);
                hit = best.geom >= 0;
            }
            bool from_cache = setup && !inflight && active && P.b == 0 && reuse;
            if (STATS) t1 = __builtin_amdgcn_s_memtime();
            bool settle_now = shade_now, goes_on = false;
            for (;;) {
                if (from_cache) {
                    const unsigned off = ((unsigned)P.item & kItemPixelMask) * 16u, plane = p.pix_items * 16u;
                    const char* stored = reinterpret_cast<const char*>(p.primary_hits);
                    const float4 c0 = *reinterpret_cast<const float4*>(stored + off), c1 = *reinterpret_cast<const float4*>(stored + (off + plane)),
                                 c2 = *reinterpret_cast<const float4*>(stored + (off + 2u * plane));
                    best.dist = c0.x; best.px = c0.y; best.py = c0.z; best.pz = c0.w;
                    best.cx = c1.x; best.cy = c1.y; best.cz = c1.z;
                    best.geom = __float_as_int(c1.w);
                    best.rec = __float_as_int(c2.x);
                    hit = best.geom >= 0;
                    cnt.rays += 1;
                    settle_now = true;
                }
                cnt.reused += (unsigned)__popcll(__ballot(from_cache));
                if (__ballot(settle_now) == 0ull) break;
                bool again = false;
                if (settle_now) {
                    MaterialRef M;
                    M.global = nullptr;
                    M.geom_base = L.geom_base;
                    M.g = best.geom;
                    const int r = settle_hit<EXTRAS>(p, best, hit, M, P);
                    inflight = false;
                    active = r != kPixelDone;
                    goes_on = r == kGoesOn;
                    again = r == kNewSample && reuse;
                }
                settle_now = false;
                from_cache = again && __popcll(__ballot(again)) >= p.reuse_quorum;
            }
            if (goes_on) {
                MaterialRef M;
                M.global = nullptr;
                M.geom_base = L.geom_base;
                M.g = best.geom;
                scatter<EXTRAS>(p, best, M, P);
            }
            if (STATS) t2 = __builtin_amdgcn_s_memtime();
            {
                const bool need = setup && !active && !exhausted;
                if (__ballot(need) != 0ull) {
                    const bool got = acquire_pixel(p, lane, P, WQ, need, cnt.rays);
                    if (need) {
                        active = got;
                        exhausted = !got;
                    }
                }
            }
            if (STATS) t3 = __builtin_amdgcn_s_memtime();
            // -- the next ray's query: planes first; a lane whose ray can reach a mesh enters it and posts the job --
            Segment S;
            S.cur = kDone;
            bool over = false;
            const bool begin = setup && active && !(P.b == 0 && reuse);
            if (begin) {
                if (STATS) probe_round(cnt.segment_rounds);
                over = begin_segment<STATS>(L, p.walls, p.geoms, p.num_geoms, p.num_planes, p.tris, P.ray, S, cnt, p.cut_last != 0 && P.b == p.bounces - 1, p.emitter_mask);
                cnt.rays += 1;
                inflight = true;
                obest = S.best;
                opend = S.pend;
                omeshes = S.meshes;
            }
            cnt.cut += (unsigned)__popcll(__ballot(over));
            // (a query that is going on - its last job came back without a tie - enters its next candidate mesh here as well)
            const bool enter = mine && inflight && !waiting && omeshes != 0u;
            bool post = false;
            if (enter) {
                S.best = obest;
                S.pend = opend;
                S.meshes = omeshes;
                S.cur = kDone;
                S.resume = 0;
                while (S.cur == kDone && S.meshes != 0u) start_next_mesh(L, P.ray, S);
                omeshes = S.meshes;
                if (S.cur != kDone) {
                    // (the pending candidate's identity stays here: the job only needs to know that there is one, and how far)
                    S.pend.rec = opend.geom >= 0 ? kPendOwner : kPendEmpty;
                    S.sp = 0;
                    job_write<BLOCK>(Q, me, S);
                    post = true;
                    waiting = true;
                }
            }
            pool_push(Q, lane, me, post || repost);
            more_meshes = omeshes != 0u;
            park_store(me, P, obest, opend, omeshes, cnt.rays);
            if (STATS) {
                t4 = __builtin_amdgcn_s_memtime();
                tphase[0] += t1 - t0; tphase[1] += t2 - t1; tphase[2] += t3 - t2; tphase[3] += t4 - t3;
            }
            {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                tw_setup += now - tw_mark;
                tw_mark = now;
                n_setup += 1u;
            }
            continue;
        }

        // ---- the traverse role (a function of its own: see pool_role) ----
        {
            unsigned long long t4 = 0;
            if (STATS) t4 = __builtin_amdgcn_s_memtime();
            const int me_r = opaque(tid);
            const bool nojob_ready = !waiting && (inflight || active || !exhausted);
            if (pool_role<STATS, BLOCK>(Q.job_base, me_r, waiting, nojob_ready, STATS ? &cnt : nullptr)) stuck = true;
            if (STATS) tphase[4] += __builtin_amdgcn_s_memtime() - t4;
            {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                tw_role += now - tw_mark;
                tw_mark = now;
                n_role += 1u;
            }
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&p.counters[4], tphase[0]);
        atomicAdd(&p.counters[5], tphase[1]);
        atomicAdd(&p.counters[6], tphase[2]);
        atomicAdd(&p.counters[7], tphase[3]);
        atomicAdd(&p.counters[13], tphase[4]);
        atomicMax(&p.counters[22], tphase[0] + tphase[1] + tphase[2] + tphase[3] + tphase[4]);
        atomicMax(&p.counters[24], ~(unsigned long long)wave_t0);
        atomicMax(&p.counters[25], (unsigned long long)wall_clock64());
    }
    flush_counters(p, lane, cnt, STATS);
    if (!STATS && lane == 0) {
        unsigned long long* t = reinterpret_cast<unsigned long long*>(&pool_u32()[Q.ctrl + 16]);
        __hip_atomic_fetch_add(&t[4], tw_setup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&t[5], tw_role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&t[6], tw_rest + (__builtin_amdgcn_s_memtime() - tw_mark), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&t[7], (unsigned long long)n_setup | ((unsigned long long)n_role << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (!STATS) {
        // node visits and round counts of the production kernel (the instrumented build counts them per lane): raw counters [1], [8],
        // [9] and, for the lanes that held a leaf in a leaf phase, [14]
        __syncthreads();
        if (tid == 0) {
            const unsigned long long* t = reinterpret_cast<const unsigned long long*>(&pool_u32()[Q.ctrl + 16]);
            atomicAdd(&p.counters[8], t[0]);
            atomicAdd(&p.counters[1], t[1]);
            atomicAdd(&p.counters[9], t[2]);
            atomicAdd(&p.counters[14], t[3]);
            atomicAdd(&p.counters[16], t[4]);
            atomicAdd(&p.counters[17], t[5]);
            atomicAdd(&p.counters[18], t[6]);
            atomicAdd(&p.counters[11], t[7] & 0xFFFFFFFFull);
            atomicAdd(&p.counters[12], t[7] >> 32);
        }
    }
}

// ---- the brute-force mega-kernel (reference loop, validation path) ---------------------------------------------------

template <bool STATS>
__global__ __launch_bounds__(kBlockThreads) void trace_brute_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    float4* batch = reinterpret_cast<float4*>(ff_smem); // triangle batch buffer
    Counters cnt = {};
    Path P;
    init_path(P);
    WaveQueue Q = make_wave_queue(p);
    bool active = false, exhausted = false;
    for (;;) {
        {
            const bool need = !active && !exhausted;
            if (__ballot(need) != 0ull) {
                const bool got = acquire_pixel(p, lane, P, Q, need, cnt.rays);
                if (need) {
                    active = got;
                    exhausted = !got;
                }
            }
        }
        // every thread of the workgroup takes part in staging the triangle batches
        if (__syncthreads_or(active ? 1 : 0) == 0) break;
        Best best;
        closest_hit_brute<STATS>(p.geoms, p.num_geoms, p.tris, batch, active, P.ray, best, cnt, p.trinormals);
        if (!active) continue;
        const bool hit = best.geom >= 0;
        MaterialRef M;
        M.global = hit ? &p.geoms[best.geom] : p.geoms;
        M.geom_base = 0;
        M.g = 0;
        active = shade_and_advance(p, best, hit, M, P);
    }
    flush_counters(p, lane, cnt, STATS);
}

// Batch closest-hit query: intersectRays (kernel.cu:127-176) for caller-supplied rays, one thread per ray.
template <int MODE, int BIG = 0>
__global__ __launch_bounds__(kBlockThreads) void ray_batch_kernel(const RayBatchParams p)
{
    const int tid = threadIdx.x;
    const LdsT<BIG> L = make_lds<BIG>(p.lds_nodes, p.stack_depth, kBlockThreads, tid, p.num_quads, nullptr, p.geoms, p.top_first, p.top_lds_first,
                                      p.top_lds_count, BIG ? p.num_scan : 0, p.stack_spill);
    const uint4* nodes4 = reinterpret_cast<const uint4*>(p.nodes4);
    if (MODE == FF_TRACE_BVH) stage_scene(L, nodes4, p.geoms, p.num_geoms, p.num_planes, tid, kBlockThreads);
    float4* batch = reinterpret_cast<float4*>(ff_smem);
    const int i = blockIdx.x * kBlockThreads + tid;
    const bool live = i < p.n;
    Ray wr = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    if (live) {
        const FfRay r = p.rays[i];
        wr.ox = r.m_origin.x; wr.oy = r.m_origin.y; wr.oz = r.m_origin.z;
        wr.dx = r.m_direction.x; wr.dy = r.m_direction.y; wr.dz = r.m_direction.z;
    }
    Best best;
    best.dist = kInf; best.geom = -1; best.rec = -1; best.px = best.py = best.pz = 0.f; best.cx = best.cy = 0.f; best.cz = 1.f;
    Counters cnt = {};
    if (MODE == FF_TRACE_BRUTE_FORCE) closest_hit_brute<false>(p.geoms, p.num_geoms, p.tris, batch, live, wr, best, cnt);
    else if (live) closest_hit_deferred<false>(L, p.walls, p.geoms, p.num_geoms, p.num_planes, p.tris, nodes4, wr, best, cnt);
    if (live && ((cnt.guard_hits >> (threadIdx.x & 63)) & 1ull) != 0ull && p.guard_hits) atomicAdd(p.guard_hits, 1ull);
    if (!live) return;
    FfIntersect out;
    out.m_intersectionPoint.x = 0.f; out.m_intersectionPoint.y = 0.f; out.m_intersectionPoint.z = 0.f;
    out.m_normal.x = 0.f; out.m_normal.y = 0.f; out.m_normal.z = 0.f;
    out.m_t = 0.f;          // utilities.h:62
    out.m_hit = 0;          // :63
    out._pad[0] = out._pad[1] = out._pad[2] = 0;
    out.geometryIndex = -1; // :64
    out.triangleIndex = -1; // :65
    if (best.geom >= 0) {
        const GeomRecord& G = p.geoms[best.geom];
        MaterialRef M;
        M.global = &G;
        M.geom_base = L.geom_base;
        M.g = 0;
        float nx, ny, nz;
        world_normal(M, best, true, nx, ny, nz);
        out.m_intersectionPoint.x = best.px; out.m_intersectionPoint.y = best.py; out.m_intersectionPoint.z = best.pz;
        out.m_normal.x = nx; out.m_normal.y = ny; out.m_normal.z = nz;
        out.m_t = best.dist;   // kernel.cu:119: the world distance
        out.m_hit = 1;
        out.geometryIndex = G.orig_index;
        out.triangleIndex = best.rec >= 0 ? p.tris[best.rec].orig_index : -1;
    }
    p.out[i] = out;
}

// Which pixels of the local image can the camera not see anything in?  One thread per pixel item (tile-major, like the work
// queue's): primary ray (kernel.cu:197-205) against the padded box around all geometries - conservative like slab_may_hit:
// approximate reciprocals, inflated exit, NaN counts as "may hit" - one mask word per 64 items, and the block sums of a culled
// pixel zeroed for every block of the frame (the combine pass reads them all).
__global__ void cull_mask_kernel(const KParams p, unsigned long long* mask)
{
    const unsigned pitem = blockIdx.x * blockDim.x + threadIdx.x;
    bool culled = false;
    if (pitem < p.pix_items) {
        const int tile = (int)(pitem >> 6), in = (int)(pitem & 63u);
        const int lx = (tile % p.tiles_per_row) * 8 + (in & 7);
        const int ly = (tile / p.tiles_per_row) * 8 + (in >> 3);
        const int strip = ly / p.strip_rows;
        const int gy = p.y0 + (strip * p.num_parts + p.part) * p.strip_rows + (ly - strip * p.strip_rows);
        const int gx = p.x0 + lx;
        if (lx < p.local_width && gx < p.xlim && ly < p.local_rows && gy < p.ylim) {
            Ray r;
            primary_ray(p, (unsigned)gx | ((unsigned)gy << 16), r);
            culled = !slab_may_hit(p.scene_min[0], p.scene_min[1], p.scene_min[2], p.scene_max[0], p.scene_max[1], p.scene_max[2], make_world_slab(r), kInf);
            if (culled)
                for (int b = 0; b < p.num_blocks; ++b) p.blocksums[(size_t)pitem * p.num_blocks + b] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const unsigned long long word = __ballot(culled);
    if ((threadIdx.x & 63) == 0 && pitem < ((p.pix_items + 63u) & ~63u)) {
        mask[pitem >> 6] = word;
        // (the host adds the slots and turns pixels into rays; on ONE address the 32 000 atomics of a 1080p frame queue up for
        // 0.3 ms - five times the reference's whole 1-spp frame)
        if (word) atomicAdd(&p.counters[kCulledPixelsWord + kRaySlotStride * ((pitem >> 6) % kRaySlots)], (unsigned long long)__popcll(word));
    }
}

// Final pass of a frame: add every pixel's sample-block sums in block order, scale by 1/spp (kernel.cu:214 stores the
// colour as 8 bits; the float radiance is kept next to it), write rows coalesced.  Untraced pixels keep the cleared 0.
__global__ void combine_kernel(const KParams p)
{
    const int lx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ly = blockIdx.y;
    if (lx >= p.local_width || ly >= p.local_rows) return;
    const int strip = ly / p.strip_rows;
    const int gy = p.y0 + (strip * p.num_parts + p.part) * p.strip_rows + (ly - strip * p.strip_rows);
    if (p.x0 + lx >= p.xlim || gy >= p.ylim) return;
    const unsigned pitem = (unsigned)(((ly >> 3) * p.tiles_per_row + (lx >> 3)) * 64 + ((ly & 7) * 8 + (lx & 7)));
    float ax = 0.f, ay = 0.f, az = 0.f;
    // (a pixel that sees nothing is black whatever its sums hold: a frame that takes the mask over from the last one - same camera, same
    // scene - runs no pass that zeroes them, and its dropped items wrote none)
    const bool culled = p.cull_mask != nullptr && ((p.cull_mask[pitem >> 6] >> (pitem & 63u)) & 1ull) != 0ull;
    for (int b = 0; b < (culled ? 0 : p.num_blocks); ++b) {
        float4 v;
        if (p.tail_block >= 0 && b >= p.tail_block) {
            // this block was traced sample by sample (the frame's last block, or its last two): the sequential sum a lane would have
            // kept in registers.  The stored samples are numbered from the first of those blocks on.
            float bx = 0.f, by = 0.f, bz = 0.f;
            const float4* sp = p.tail_samples + pitem; // sample-major: neighbouring threads read neighbouring values
            const int first = (b - p.tail_block) * p.block_spp, past = min(p.tail_samples_in_block, first + p.block_spp);
            for (int i = first; i < past; ++i) {
                const float4 l = sp[(size_t)i * p.pix_items];
                bx = bx + l.x;
                by = by + l.y;
                bz = bz + l.z;
            }
            v = make_float4(bx, by, bz, 0.f);
        } else {
            v = p.blocksums[(size_t)pitem * p.num_blocks + b];
        }
        ax = ax + v.x;
        ay = ay + v.y;
        az = az + v.z;
    }
    float rx = ax, ry = ay, rz = az;
    if (p.shade_mode != FF_SHADE_NORMAL_DEBUG) {
        const float inv = 1.0f / (float)p.spp_total;
        rx = ax * inv; ry = ay * inv; rz = az * inv;
    }
    const size_t lpix = (size_t)ly * (size_t)p.local_width + (size_t)lx;
    if (p.radiance) {
        p.radiance[3 * lpix] = rx;
        p.radiance[3 * lpix + 1] = ry;
        p.radiance[3 * lpix + 2] = rz;
    }
    if (p.rgb8) {
        p.rgb8[3 * lpix] = to_u8(rx);
        p.rgb8[3 * lpix + 1] = to_u8(ry);
        p.rgb8[3 * lpix + 2] = to_u8(rz);
    }
}

// Exhaustive self-check of ieee_rcp / ieee_sqrt against the compiler's IEEE expansions: every float bit pattern.
__global__ void ieee_check_kernel(unsigned long long* mismatches)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad_rcp = 0, bad_sqrt = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const float a = ieee_rcp(x), ra = 1.0f / x;
        const float b = ieee_sqrt(x), rb = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(ra) && !(a != a && ra != ra)) ++bad_rcp;
        if (__float_as_uint(b) != __float_as_uint(rb) && !(b != b && rb != rb)) ++bad_sqrt;
    }
    if (bad_rcp) atomicAdd(&mismatches[0], bad_rcp);
    if (bad_sqrt) atomicAdd(&mismatches[1], bad_sqrt);
}

// Progressive accumulation (ff_render_progressive): running sum of whole frames, output = sum * (1 / frames).
__global__ void accumulate_kernel(float* __restrict__ sum, const float* __restrict__ frame, float* __restrict__ mean, unsigned char* __restrict__ rgb8,
                                  size_t values, int first_frame, float inv_frames)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= values) return;
    const float acc = first_frame ? frame[i] : sum[i] + frame[i];
    sum[i] = acc;
    const float m = acc * inv_frames;
    if (mean) mean[i] = m;
    if (rgb8) rgb8[i] = to_u8(m);
}

// Strip de-interleave after the framebuffer gather: src = parts' compact row blocks back to back, dst = image order.
__global__ void deinterleave_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int width, int height,
                                    int strip_rows, int num_parts, int elem_bytes)
{
    const size_t row_bytes = (size_t)width * (size_t)elem_bytes;
    const int y = blockIdx.y;
    if (y >= height) return;
    const int strip = y / strip_rows, part = strip % num_parts, local_strip = strip / num_parts;
    // rows owned by parts before `part`
    size_t rows_before = 0;
    const int nstrips = (height + strip_rows - 1) / strip_rows;
    for (int q = 0; q < part; ++q) {
        const int owned = (nstrips - q + num_parts - 1) / num_parts; // strips q, q+P, ...
        size_t rows = (size_t)owned * (size_t)strip_rows;
        // the last strip of the image may be short
        const int last = nstrips - 1;
        if (owned > 0 && last % num_parts == q) rows -= (size_t)(nstrips * strip_rows - height);
        rows_before += rows;
    }
    const size_t local_row = (size_t)local_strip * (size_t)strip_rows + (size_t)(y - strip * strip_rows);
    const unsigned char* s = src + (rows_before + local_row) * row_bytes;
    unsigned char* d = dst + (size_t)y * row_bytes;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// Multi-GPU gather, last step (ff_dist.cpp): `src` holds every part's packed strips, part after part, each part as
// [rows x width float3 radiance][rows x width rgb8], both sections padded to 16 bytes; one pass scatters all rows of both
// framebuffers to image order.  Row y belongs to strip y / strip_rows, which part (strip % num_parts) rendered as its
// local strip strip / num_parts.
__global__ void unpack_strips_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ rgb8, float* __restrict__ radiance, int width,
                                     int height, int strip_rows, int num_parts)
{
    const int y = blockIdx.y;
    if (y >= height) return;
    const int nstrips = (height + strip_rows - 1) / strip_rows;
    const int strip = y / strip_rows, part = strip % num_parts, local_strip = strip / num_parts;
    auto part_rows = [&](int q) {
        const int owned = (nstrips - q + num_parts - 1) / num_parts; // strips q, q + P, ...
        size_t rows = (size_t)owned * (size_t)strip_rows;
        if (owned > 0 && (nstrips - 1) % num_parts == q) rows -= (size_t)(nstrips * strip_rows - height); // the image's last strip may be short
        return rows;
    };
    auto pad16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t base = 0;
    for (int q = 0; q < part; ++q) base += pad16(part_rows(q) * (size_t)width * 12) + pad16(part_rows(q) * (size_t)width * 3);
    const size_t rows = part_rows(part);
    const size_t local_row = (size_t)local_strip * (size_t)strip_rows + (size_t)(y - strip * strip_rows);
    const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (radiance) {
        const float* s = reinterpret_cast<const float*>(src + base) + local_row * (size_t)width * 3;
        float* d = radiance + (size_t)y * (size_t)width * 3;
        for (size_t i = first; i < (size_t)width * 3; i += stride) d[i] = s[i];
    }
    if (rgb8) {
        const unsigned char* s = src + base + pad16(rows * (size_t)width * 12) + local_row * (size_t)width * 3;
        unsigned char* d = rgb8 + (size_t)y * (size_t)width * 3;
        for (size_t i = first; i < (size_t)width * 3; i += stride) d[i] = s[i];
    }
}

} // namespace

#ifdef FF_PROBE
// Register-pressure probes (tools/diag/probe_kernel.sh): compile ONE instantiation to ISA in a few seconds.
namespace {
template __global__ void FF_PROBE(const KParams);
}
#else
// (num_geoms: the records cached in LDS; 0 for scenes of more than 32 geometries, whose records stay in global memory)
size_t bvh_lds_bytes(int lds_nodes, int stack_depth, int block_threads, int num_geoms)
{
    return (size_t)lds_nodes * sizeof(Bvh4Node) + (size_t)stack_depth * (size_t)block_threads * sizeof(unsigned) +
           (size_t)num_geoms * sizeof(GeomRecord);
}

int max_lds_nodes(int stack_depth, int block_threads, int num_geoms, size_t reserve)
{
    const long avail = (long)kLdsBudgetBytes - (long)reserve - (long)stack_depth * (long)block_threads * (long)sizeof(unsigned) -
                       (long)num_geoms * (long)sizeof(GeomRecord);
    return avail > 0 ? (int)(avail / (long)sizeof(Bvh4Node)) : 0;
}

hipError_t prepare_kernels()
{
    hipError_t e;
#define FF_SET_LDS(K)                                                                                                     \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudgetBytes); \
    if (e != hipSuccess) return e;
    // Only the BVH kernels go past the 64 KiB default (node cache + stacks + geometry records); the brute-force kernels
    // use a 48 KiB batch buffer plus a little static LDS, and asking for 160 KiB on top of static LDS is rejected.
    FF_SET_LDS((trace_bvh_kernel<false, 512, false>))
    FF_SET_LDS((trace_bvh_kernel<false, 512, true>))
    FF_SET_LDS((trace_bvh_kernel<true, 512, false>))
    FF_SET_LDS((trace_bvh_kernel<true, 512, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, false>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, true>))
    FF_SET_LDS((trace_bvh_kernel<true, 768, false>))
    FF_SET_LDS((trace_bvh_kernel<true, 768, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, false>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, true>))
    FF_SET_LDS((trace_bvh_kernel<true, 1024, false>))
    FF_SET_LDS((trace_bvh_kernel<true, 1024, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 512, true, 1>))
    FF_SET_LDS((trace_bvh_kernel<true, 512, true, 1>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, true, 1>))
    FF_SET_LDS((trace_bvh_kernel<true, 768, true, 1>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, true, 1>))
    FF_SET_LDS((trace_bvh_kernel<true, 1024, true, 1>))
    FF_SET_LDS((trace_bvh_kernel<false, 512, true, 2>))
    FF_SET_LDS((trace_bvh_kernel<true, 512, true, 2>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, true, 2>))
    FF_SET_LDS((trace_bvh_kernel<true, 768, true, 2>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, true, 2>))
    FF_SET_LDS((trace_bvh_kernel<true, 1024, true, 2>))
    FF_SET_LDS((trace_bvh_kernel<false, 512, true, 0, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, true, 0, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, true, 0, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 512, true, 1, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, true, 1, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, true, 1, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 512, true, 2, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 768, true, 2, true>))
    FF_SET_LDS((trace_bvh_kernel<false, 1024, true, 2, true>))
    FF_SET_LDS((trace_pool_kernel<false, 1024, false>))
    FF_SET_LDS((trace_pool_kernel<false, 1024, true>))
    FF_SET_LDS((trace_pool_kernel<true, 1024, false>))
    FF_SET_LDS((trace_pool_kernel<true, 1024, true>))
    FF_SET_LDS((ray_batch_kernel<FF_TRACE_BVH>))
    FF_SET_LDS((ray_batch_kernel<FF_TRACE_BVH, 1>))
    FF_SET_LDS((ray_batch_kernel<FF_TRACE_BVH, 2>))
#undef FF_SET_LDS
    return hipSuccess;
}

size_t pool_lds_bytes(int block_threads) { return (size_t)block_threads * 48 + (size_t)kPoolRing * 2 + 128; } // jobs, ring, [head, tail, -, -], the role's inputs, the workgroup's tallies

hipError_t launch_trace(const KParams& p, int trace_mode, bool collect_stats, int grid_blocks, int block_threads, hipStream_t stream,
                        const char** kernel_name, bool pool, bool prepass)
{
    const dim3 grid(grid_blocks);
    const char* name = "";
    if (prepass) {
        // the pre-pass of a frame (KParams::primary_hits): one instantiation per workgroup size and scene size, the general one
        const dim3 block(block_threads);
        const int big = p.num_geoms <= kChunkGeometries ? 0 : (p.num_geoms <= kMaxLdsRecords ? 1 : 2);
        const size_t lds = bvh_lds_bytes(p.lds_nodes, p.stack_depth, block_threads, big == 2 ? 0 : p.num_geoms);
#define FF_LAUNCH_PRE(B)                                                                                                  \
    do {                                                                                                                  \
        if (big == 0) hipLaunchKernelGGL((trace_bvh_kernel<false, B, true, 0, true>), grid, block, lds, stream, p);      \
        else if (big == 1) hipLaunchKernelGGL((trace_bvh_kernel<false, B, true, 1, true>), grid, block, lds, stream, p); \
        else hipLaunchKernelGGL((trace_bvh_kernel<false, B, true, 2, true>), grid, block, lds, stream, p);               \
    } while (0)
        if (block_threads == 1024) FF_LAUNCH_PRE(1024);
        else if (block_threads == 768) FF_LAUNCH_PRE(768);
        else FF_LAUNCH_PRE(512);
#undef FF_LAUNCH_PRE
        if (kernel_name) *kernel_name = "trace_bvh_kernel<false, B, true, big, true>"; // (the frame's own launches name their kernel exactly; nobody asks for this one)
        return hipGetLastError();
    }
    if (trace_mode == FF_TRACE_BVH && pool && p.num_geoms <= kChunkGeometries && block_threads == 1024) {
        const dim3 block(block_threads);
        const size_t lds = bvh_lds_bytes(p.lds_nodes, p.stack_depth, block_threads, p.num_geoms) + pool_lds_bytes(block_threads);
        const bool extras = p.num_planes > p.num_quads || p.has_specular != 0 || p.trinormals != nullptr;
#define FF_LAUNCH_POOL(B)                                                                                                 \
    do {                                                                                                                  \
        if (collect_stats) {                                                                                              \
            if (extras) { hipLaunchKernelGGL((trace_pool_kernel<true, B, true>), grid, block, lds, stream, p); name = "trace_pool_kernel<true, " #B ", true>"; } \
            else { hipLaunchKernelGGL((trace_pool_kernel<true, B, false>), grid, block, lds, stream, p); name = "trace_pool_kernel<true, " #B ", false>"; } \
        } else {                                                                                                          \
            if (extras) { hipLaunchKernelGGL((trace_pool_kernel<false, B, true>), grid, block, lds, stream, p); name = "trace_pool_kernel<false, " #B ", true>"; } \
            else { hipLaunchKernelGGL((trace_pool_kernel<false, B, false>), grid, block, lds, stream, p); name = "trace_pool_kernel<false, " #B ", false>"; } \
        }                                                                                                                 \
    } while (0)
        FF_LAUNCH_POOL(1024); // (an experiment: only the workgroup size the production kernel runs is instantiated)
#undef FF_LAUNCH_POOL
    } else if (trace_mode == FF_TRACE_BVH) {
        const dim3 block(block_threads);
        const int big = p.num_geoms <= kChunkGeometries ? 0 : (p.num_geoms <= kMaxLdsRecords ? 1 : 2);
        const size_t lds = bvh_lds_bytes(p.lds_nodes, p.stack_depth, block_threads, big == 2 ? 0 : p.num_geoms);
        const bool spheres = p.num_planes > p.num_quads || p.has_specular != 0 || p.trinormals != nullptr; // any extra: the full kernel
#define FF_LAUNCH_BVH(B)                                                                                                  \
    do {                                                                                                                  \
        if (big == 1) {                                                                                                   \
            if (collect_stats) { hipLaunchKernelGGL((trace_bvh_kernel<true, B, true, 1>), grid, block, lds, stream, p); name = "trace_bvh_kernel<true, " #B ", true, 1, false>"; } \
            else { hipLaunchKernelGGL((trace_bvh_kernel<false, B, true, 1>), grid, block, lds, stream, p); name = "trace_bvh_kernel<false, " #B ", true, 1, false>"; } \
        } else if (big == 2) {                                                                                            \
            if (collect_stats) { hipLaunchKernelGGL((trace_bvh_kernel<true, B, true, 2>), grid, block, lds, stream, p); name = "trace_bvh_kernel<true, " #B ", true, 2, false>"; } \
            else { hipLaunchKernelGGL((trace_bvh_kernel<false, B, true, 2>), grid, block, lds, stream, p); name = "trace_bvh_kernel<false, " #B ", true, 2, false>"; } \
        } else if (collect_stats) {                                                                                       \
            if (spheres) { hipLaunchKernelGGL((trace_bvh_kernel<true, B, true>), grid, block, lds, stream, p); name = "trace_bvh_kernel<true, " #B ", true, 0, false>"; } \
            else { hipLaunchKernelGGL((trace_bvh_kernel<true, B, false>), grid, block, lds, stream, p); name = "trace_bvh_kernel<true, " #B ", false, 0, false>"; } \
        } else {                                                                                                          \
            if (spheres) { hipLaunchKernelGGL((trace_bvh_kernel<false, B, true>), grid, block, lds, stream, p); name = "trace_bvh_kernel<false, " #B ", true, 0, false>"; } \
            else { hipLaunchKernelGGL((trace_bvh_kernel<false, B, false>), grid, block, lds, stream, p); name = "trace_bvh_kernel<false, " #B ", false, 0, false>"; } \
        }                                                                                                                 \
    } while (0)
        if (block_threads == 1024) FF_LAUNCH_BVH(1024);
        else if (block_threads == 768) FF_LAUNCH_BVH(768);
        else FF_LAUNCH_BVH(512);
#undef FF_LAUNCH_BVH
    } else {
        const size_t lds = (size_t)kBruteBatchTris * sizeof(TriRecord);
        const dim3 block(kBlockThreads);
        if (collect_stats) hipLaunchKernelGGL((trace_brute_kernel<true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((trace_brute_kernel<false>), grid, block, lds, stream, p);
        name = collect_stats ? "trace_brute_kernel<true>" : "trace_brute_kernel<false>";
    }
    if (kernel_name) *kernel_name = name;
    return hipGetLastError();
}

hipError_t launch_combine(const KParams& p, hipStream_t stream)
{
    if (p.local_width <= 0 || p.local_rows <= 0) return hipSuccess;
    const dim3 block(256), grid((p.local_width + 255) / 256, p.local_rows);
    hipLaunchKernelGGL(combine_kernel, grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_cull_mask(const KParams& p, unsigned long long* mask, hipStream_t stream)
{
    if (p.pix_items == 0) return hipSuccess;
    hipLaunchKernelGGL(cull_mask_kernel, dim3((p.pix_items + 255) / 256), dim3(256), 0, stream, p, mask);
    return hipGetLastError();
}

hipError_t launch_ieee_check(unsigned long long* mismatches2, hipStream_t stream)
{
    hipLaunchKernelGGL(ieee_check_kernel, dim3(256 * 8), dim3(256), 0, stream, mismatches2);
    return hipGetLastError();
}

hipError_t launch_accumulate(float* sum, const float* frame, float* mean, unsigned char* rgb8, size_t values, int first_frame, float inv_frames,
                             hipStream_t stream)
{
    if (values == 0) return hipSuccess;
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((values + 255) / 256)), dim3(256), 0, stream, sum, frame, mean, rgb8, values, first_frame,
                       inv_frames);
    return hipGetLastError();
}

hipError_t launch_ray_batch(const RayBatchParams& p, int trace_mode, hipStream_t stream)
{
    if (p.n <= 0) return hipSuccess;
    const int big = p.num_geoms <= kChunkGeometries ? 0 : (p.num_geoms <= kMaxLdsRecords ? 1 : 2);
    const size_t lds = trace_mode == FF_TRACE_BVH ? bvh_lds_bytes(p.lds_nodes, p.stack_depth, kBlockThreads, big == 2 ? 0 : p.num_geoms)
                                                  : (size_t)kBruteBatchTris * sizeof(TriRecord);
    const dim3 grid((p.n + kBlockThreads - 1) / kBlockThreads), block(kBlockThreads);
    if (trace_mode == FF_TRACE_BVH && big == 1) hipLaunchKernelGGL((ray_batch_kernel<FF_TRACE_BVH, 1>), grid, block, lds, stream, p);
    else if (trace_mode == FF_TRACE_BVH && big == 2) hipLaunchKernelGGL((ray_batch_kernel<FF_TRACE_BVH, 2>), grid, block, lds, stream, p);
    else if (trace_mode == FF_TRACE_BVH) hipLaunchKernelGGL((ray_batch_kernel<FF_TRACE_BVH>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((ray_batch_kernel<FF_TRACE_BRUTE_FORCE>), grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const void* src, void* dst, int width, int height, int strip_rows, int num_parts, int elem_bytes,
                               hipStream_t stream)
{
    if (width <= 0 || height <= 0) return hipSuccess;
    const dim3 grid(4, height), block(256);
    hipLaunchKernelGGL(deinterleave_kernel, grid, block, 0, stream, (const unsigned char*)src, (unsigned char*)dst, width, height,
                       strip_rows, num_parts, elem_bytes);
    return hipGetLastError();
}

hipError_t launch_unpack_strips(const void* src, unsigned char* rgb8, float* radiance, int width, int height, int strip_rows, int num_parts,
                                hipStream_t stream)
{
    if (width <= 0 || height <= 0 || (!rgb8 && !radiance)) return hipSuccess;
    const dim3 grid(std::max(1, std::min(8, (width * 3 + 255) / 256)), height), block(256);
    hipLaunchKernelGGL(unpack_strips_kernel, grid, block, 0, stream, (const unsigned char*)src, rgb8, radiance, width, height, strip_rows, num_parts);
    return hipGetLastError();
}

#endif // FF_PROBE
} // namespace ff
