// ff_kernels.hip — the gfx950 (CDNA4, wave64) trace kernels.
//
// What the reference runs per pixel (kernel.cu:186-221: primary ray, brute-force closest hit over all geometries and
// triangles, shade, 8-bit store) is restructured here as a persistent mega-kernel:
//
//   * one workgroup per CU; the top of the BVH node array and all geometry records are staged ONCE per workgroup into
//     LDS (64-byte nodes with both child boxes, kept as four planes of 16-byte quarters so that a wave's reads spread over
//     all banks) and every lane keeps its traversal stack in LDS (lane-strided: pushes / pops are bank-conflict free);
//   * lanes pull (pixel, sample block) work items from one global counter with a wave-wide ballot + prefix compaction
//     (several short items per fetch when a frame has few samples per pixel); a lane sums its block's samples in order,
//     terminated paths regenerate in place, and a combine pass adds a pixel's blocks in order;
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
    // occupancy probes (instrumented launches only): wave-level rounds of each phase.  The active-lane totals of the
    // phases are the counters above (nodes = inner-step lanes, tris = triangle-test lanes, planes, rays).
    unsigned inner_rounds, leaf_rounds, tri_rounds, plane_rounds, segment_rounds;
    unsigned no_mesh; // queries that needed no mesh traversal (planes only)
    unsigned plane_exact; // plane tests that fell inside a screening margin and ran the exact reference test
    unsigned long long t_start, t_inner, t_leaf; // instrumented launches: wave cycles in mesh starts / inner phases / leaf phases
    unsigned long long t_b1, t_b2, t_b3;         // ... and in the three parts of begin_segment (quad boxes / quad screens / mesh boxes)
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
//   [ nodes: 4 planes of lds_nodes x 16 B ][ traversal stacks: stack_depth x BLOCK x 4 B, lane-strided ][ geometry records: G x 288 B ]
//
// ff_smem is indexed directly (never through a generic pointer) so that every access compiles to ds_read/ds_write.
extern __shared__ uint4 ff_smem[];

constexpr int kGeomVec4 = (int)(sizeof(GeomRecord) / 16); // 18 float4 per geometry record
constexpr int kDone = 0x7fffffff;                         // traversal cursor of a lane with nothing left to visit

struct Lds {
    int node_count; // nodes [0, node_count) live in LDS
    int stack_base; // uint index of this lane's stack slot 0 (in units of 4 bytes from ff_smem)
    int stride;     // uints between consecutive stack entries of one lane (= block size)
    int geom_base;  // uint4 index of geometry record 0
    int num_quads;  // geometry records [0, num_quads) are planes; [num_quads, num_planes) spheres; meshes follow
    const float4* smooth_normals; // non-null: triangle hits carry the interpolated vertex normal (FF_SHADE_DIFFUSE_PATH_SMOOTH)
    int last_base;  // first record of the last chunk of 32 geometry records: ((num_geoms - 1) / 32) * 32
};

__device__ __forceinline__ Lds make_lds(int lds_nodes, int stack_depth, int block, int tid, int num_quads, const float4* smooth_normals = nullptr,
                                        int last_base = 0)
{
    Lds L;
    L.num_quads = num_quads;
    L.smooth_normals = smooth_normals;
    L.last_base = last_base;
    L.node_count = lds_nodes;
    L.stride = block;
    L.stack_base = lds_nodes * 16 + tid;
    L.geom_base = lds_nodes * 4 + (stack_depth * block) / 4;
    return L;
}

// Stage the top of the BVH and the geometry records: coalesced 16-byte loads, 1 KiB per wave-instruction.  Persistent
// workgroups pay this once per launch, not per ray.
__device__ __forceinline__ void stage_scene(const Lds& L, const BvhNode* __restrict__ nodes, const GeomRecord* __restrict__ geoms, int num_geoms,
                                            int tid, int block)
{
    // Nodes are stored as four planes of 16-byte quarters (quarter k of node i at uint4 index k * node_count + i): lanes
    // fetch quarter k of unrelated nodes with one ds_read_b128, and in this layout those addresses spread over all LDS
    // banks, whereas whole 64-byte nodes would put every lane's quarter k on the same quarter of the banks.
    const uint4* src = reinterpret_cast<const uint4*>(nodes);
    for (int i = tid; i < L.node_count * 4; i += block) ff_smem[(i & 3) * L.node_count + (i >> 2)] = src[i];
    const uint4* gsrc = reinterpret_cast<const uint4*>(geoms);
    for (int i = tid; i < num_geoms * kGeomVec4; i += block) ff_smem[L.geom_base + i] = gsrc[i];
    __syncthreads();
}

__device__ __forceinline__ float4 lds_geom4(const Lds& L, int g, int k)
{
    return reinterpret_cast<const float4*>(ff_smem)[L.geom_base + g * kGeomVec4 + k];
}
__device__ __forceinline__ int4 lds_geom_i4(const Lds& L, int g, int k)
{
    return reinterpret_cast<const int4*>(ff_smem)[L.geom_base + g * kGeomVec4 + k];
}
__device__ __forceinline__ void stack_push(const Lds& L, int sp, int v) { reinterpret_cast<int*>(ff_smem)[L.stack_base + sp * L.stride] = v; }
__device__ __forceinline__ int stack_pop(const Lds& L, int sp) { return reinterpret_cast<const int*>(ff_smem)[L.stack_base + sp * L.stride]; }

__device__ __forceinline__ void fetch_node(const Lds& L, const BvhNode* __restrict__ nodes, int cur, uint4& q0, uint4& q1, uint4& q2, uint4& q3)
{
    if (cur < L.node_count) {
        q0 = ff_smem[cur];
        q1 = ff_smem[cur + L.node_count];
        q2 = ff_smem[cur + 2 * L.node_count];
        q3 = ff_smem[cur + 3 * L.node_count];
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(nodes) + (size_t)cur * 4;
        q0 = p[0];
        q1 = p[1];
        q2 = p[2];
        q3 = p[3];
    }
}

// kernel.cu:138 with the geometry record gathered from LDS by a lane-varying index (same arithmetic as object_space_ray).
__device__ __forceinline__ void object_space_ray_lds(const Lds& L, int g, const Ray& r, Ray& o, float& len)
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
    unsigned meshes;           // candidate meshes not started yet (bit = record index - base)
    int base;                  // first geometry record of the chunk of 32 the query is working on (0 unless the scene has > 32)
    int cur, sp, mesh;         // traversal cursor (inner >= 0, leaf < 0, kDone), stack height, record index of the current mesh
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

__device__ __forceinline__ bool exact_hit(const Lds& L, const TriRecord* __restrict__ tris, const Ray& wr, int g, int rec, float& dist, HitPoint& H,
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
__device__ __forceinline__ bool resolve_pending(const Lds& L, const TriRecord* __restrict__ tris, const Ray& wr, Pending& pend, BestId& best,
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

// Start a closest-hit query: test every plane (fast form) and remember which meshes the ray can reach.
//
// Planes are screened in a wave-uniform loop (records through scalar loads, all lanes busy) WITHOUT the IEEE sqrt/divide
// of kernel.cu:138: the hit position on the unit quad does not depend on the length of the object-space direction, so
// the screen works on the un-normalised direction M^-1*d, for which the ray parameter is the world-space parameter.
// Anything within the margins (quad edges, t ~ 0, |n.d| ~ 1e-7) is decided by the exact reference test at once.
// One chunk of up to 32 geometry records starting at S.base: screen its planes / spheres and collect its candidate meshes.
// Scenes of up to 32 geometries (the reference has 5) are a single chunk; larger scenes are worked through chunk by chunk,
// each query carrying its best / pending candidate across chunks.
template <bool STATS>
__device__ __forceinline__ void scan_chunk(const Lds& L, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                           const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Counters& cnt)
{
    const int base = S.base;
    const int prim_end = min(num_planes, base + 32), geom_end = min(num_geoms, base + 32);
    const float wlen = __builtin_amdgcn_rcpf(inv_length(wr)); // |world direction| (1 for the integrator's rays)

    // Stage 1, wave-uniform: which quads can the ray reach at all?  The padded world box of a quad is flat, so for the
    // axis-aligned walls of a box scene this conservative slab test already singles out the one wall the ray hits.
    unsigned long long tb0 = 0, tb1 = 0, tb2 = 0;
    if (STATS) tb0 = __builtin_amdgcn_s_memtime();
    const WorldSlab ws = make_world_slab(wr);
    unsigned quads = 0u;
    for (int g = base; g < prim_end; ++g) {
        const float4 bmin = lds_geom4(L, g, 14), bmax = lds_geom4(L, g, 15);
        if (slab_may_hit(bmin.x, bmin.y, bmin.z, bmax.x, bmax.y, bmax.z, ws, kInf)) quads |= 1u << (g - base);
    }
    if (STATS) tb1 = __builtin_amdgcn_s_memtime();
    // Stage 2, per lane: screen the lane's own candidates (records from the LDS copy at per-lane addresses).
    for (int guard = 0; __ballot(quads != 0u) != 0ull && guard < 32; ++guard) {
        if (quads == 0u) continue;
        const int g = base + __ffs((int)quads) - 1;
        quads &= quads - 1u;
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
            continue;
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
        const float delta = 1.0e-5f * (1.0f + omag);
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

    if (STATS) tb2 = __builtin_amdgcn_s_memtime();
    // meshes: conservative world-box test against what the planes already found
    S.meshes = 0u;
    const float limit = fminf(S.best.dist, S.pend.dist);
    for (int g = max(num_planes, base); g < geom_end; ++g) {
        const float4 bmin = lds_geom4(L, g, 14), bmax = lds_geom4(L, g, 15);
        if (lds_geom_i4(L, g, 17).x >= 0 && slab_may_hit(bmin.x, bmin.y, bmin.z, bmax.x, bmax.y, bmax.z, ws, limit)) S.meshes |= 1u << (g - base);
    }
    if (STATS && S.meshes == 0u) cnt.no_mesh += 1;
    if (STATS) {
        const unsigned long long tb3 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { cnt.t_b1 += tb1 - tb0; cnt.t_b2 += tb2 - tb1; cnt.t_b3 += tb3 - tb2; }
    }
}

// Start a closest-hit query: empty candidate slots, then the first chunk of geometry records.
template <bool STATS>
__device__ __forceinline__ void begin_segment(const Lds& L, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                              const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Counters& cnt)
{
    S.best.dist = kInf; // kernel.cu:131
    S.best.geom = -1;
    S.best.rec = -1;
    S.pend.dist = kInf;
    S.pend.geom = -1;
    S.pend.rec = -1;
    S.cur = kDone;
    S.sp = 0;
    S.mesh = -1;
    S.resume = 0;
    S.base = 0;
    scan_chunk<STATS>(L, geoms, num_geoms, num_planes, tris, wr, S, cnt);
}

// Box-pruning bound of the current mesh: refreshed whenever the lane's best/pending distance or its mesh changes, so the
// inner-node step reads one register instead of recomputing it per node.
__device__ __forceinline__ void refresh_tbound(Segment& S)
{
    S.tbound = (fminf(S.best.dist, S.pend.dist) * 1.001f + 1.0e-3f) * S.scale * 1.00001f;
}

// Take the next subtree off the lane's stack.
__device__ __forceinline__ void pop_subtree(const Lds& L, Segment& S)
{
    if (S.sp > 0) {
        --S.sp;
        S.cur = stack_pop(L, S.sp);
    } else {
        S.cur = kDone;
    }
}

// Idle lane with candidate meshes left: enter the next one.
__device__ __forceinline__ void start_next_mesh(const Lds& L, const Ray& wr, Segment& S)
{
    const int g = S.base + __ffs((int)S.meshes) - 1;
    S.meshes &= S.meshes - 1u;
    const int root = lds_geom_i4(L, g, 17).x;
    if (root < 0) return;
    float len;
    object_space_ray_lds(L, g, wr, S.osr, len);
    S.ix = safe_rcp(S.osr.dx);
    S.iy = safe_rcp(S.osr.dy);
    S.iz = safe_rcp(S.osr.dz);
    S.ox = -S.osr.ox * S.ix;
    S.oy = -S.osr.oy * S.iy;
    S.oz = -S.osr.oz * S.iz;
    S.scale = len * inv_length(wr); // object-space t per unit of world distance
    refresh_tbound(S);
    S.mesh = g;
    S.cur = root;
    S.sp = 0;
}

// One inner-node visit: test both child boxes, descend to the nearer hit child, push the other, or pop.
template <bool STATS>
__device__ __forceinline__ void inner_step(const Lds& L, const BvhNode* __restrict__ nodes, Segment& S, Counters& cnt)
{
    uint4 q0, q1, q2, q3;
    fetch_node(L, nodes, S.cur, q0, q1, q2, q3);
    if (STATS) { cnt.nodes += 1; probe_round(cnt.inner_rounds); }
    const float tbound = S.tbound;
    // left box: q0.xyz = min, q1.xyz = max; right box: q2.xyz = min, q3.xyz = max (pruning only: FMA + approximate 1/d)
    float a0 = __builtin_fmaf(__uint_as_float(q0.x), S.ix, S.ox), a1 = __builtin_fmaf(__uint_as_float(q1.x), S.ix, S.ox);
    float b0 = __builtin_fmaf(__uint_as_float(q0.y), S.iy, S.oy), b1 = __builtin_fmaf(__uint_as_float(q1.y), S.iy, S.oy);
    float c0 = __builtin_fmaf(__uint_as_float(q0.z), S.iz, S.oz), c1 = __builtin_fmaf(__uint_as_float(q1.z), S.iz, S.oz);
    const float ln = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
    const float lf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tbound));
    a0 = __builtin_fmaf(__uint_as_float(q2.x), S.ix, S.ox); a1 = __builtin_fmaf(__uint_as_float(q3.x), S.ix, S.ox);
    b0 = __builtin_fmaf(__uint_as_float(q2.y), S.iy, S.oy); b1 = __builtin_fmaf(__uint_as_float(q3.y), S.iy, S.oy);
    c0 = __builtin_fmaf(__uint_as_float(q2.z), S.iz, S.oz); c1 = __builtin_fmaf(__uint_as_float(q3.z), S.iz, S.oz);
    const float rn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
    const float rf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tbound));
    const bool hl = ln <= lf * 1.000002f, hr = rn <= rf * 1.000002f;
    const int left = (int)q0.w, right = (int)q1.w;
    // Two short predicated regions instead of a four-way branch: the next node is a select; only the push (both hit) and
    // the pop (none hit) touch LDS.
    const bool both = hl && hr, swap = both && rn < ln;
    const int next = (hl && !swap) ? left : right; // left if it is hit and not the farther of two hits, else right
    if (both) {
        stack_push(L, S.sp, swap ? left : right);
        ++S.sp;
    }
    if (hl || hr) S.cur = next;
    else pop_subtree(L, S);
}

// One leaf visit: test the leaf's triangles (fast form), then take the next entry off the stack.  On a near tie with the
// pending candidate the leaf is left under the cursor with S.resume set; the caller resolves the pending candidate and
// the leaf continues from the triangle that met the tie.
template <bool STATS>
__device__ __forceinline__ void leaf_step(const Lds& L, const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Counters& cnt)
{
    const int ref = ~S.cur;
    const int first = ref >> 3, count = (ref & 7) + 1;
    const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)first * 3;
    const Ray& r = S.osr;
    int k = S.resume > 0 ? S.resume - 1 : 0;
    S.resume = 0;
    if (STATS) probe_round(cnt.leaf_rounds);
    for (; k < count; ++k) {
        const float4 A = tp[3 * k], E1 = tp[3 * k + 1], E2 = tp[3 * k + 2];
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
                break;
            }
        }
    }
    refresh_tbound(S);
    if (S.resume > 0) return;
    pop_subtree(L, S);
}

// Settle what is still pending (the common case: the one exact evaluation of the ray, all hitting lanes together) and
// produce the hit point of the winner.
__device__ __forceinline__ void finish_segment(const Lds& L, const TriRecord* __restrict__ tris, const Ray& wr, Segment& S, Best& best)
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

// Nothing left to do in the current chunk of geometry records / in the whole query.
__device__ __forceinline__ bool chunk_done(const Segment& S) { return S.cur == kDone && S.meshes == 0u && S.resume == 0; }
__device__ __forceinline__ bool segment_done(const Lds& L, const Segment& S) { return chunk_done(S) && S.base >= L.last_base; }

// Advance the queries of the calling lanes: mesh starts, inner-node phases, leaf phases and near-tie resolutions alternate
// wave-wide until every calling lane is done or `budget` inner-node rounds have been spent (budget <= 0: no limit).
// Unfinished lanes keep their state in S and continue on the next call.
template <bool STATS>
__device__ __forceinline__ void traverse_budget(const Lds& L, const TriRecord* __restrict__ tris, const BvhNode* __restrict__ nodes, const Ray& wr,
                                                Segment& S, Counters& cnt, int budget, int leaf_threshold)
{
    const int limit = budget > 0 ? budget : kLoopGuard;
    int rounds = 0, guard = 0;
    for (;;) {
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
        if (STATS) ta = __builtin_amdgcn_s_memtime();
        while (S.cur == kDone && S.meshes != 0u) start_next_mesh(L, wr, S);
        if (STATS) tb = __builtin_amdgcn_s_memtime();
        if (__ballot(S.cur != kDone) == 0ull) break;
        for (;;) {
            const bool inner = S.cur >= 0 && S.cur != kDone;
            if (__ballot(inner) == 0ull) break;
            // enough lanes hold a leaf: test the leaves now instead of idling them until the last lane finds one
            if (__popcll(__ballot(S.cur < 0)) >= leaf_threshold) break;
            if (rounds >= limit) break;
            ++rounds;
            if (inner) inner_step<STATS>(L, nodes, S, cnt);
        }
        if (STATS) tc = __builtin_amdgcn_s_memtime();
        if (S.cur < 0) leaf_step<STATS>(L, tris, wr, S, cnt);
        if (STATS) {
            td = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { cnt.t_start += tb - ta; cnt.t_inner += tc - tb; cnt.t_leaf += td - tc; }
        }
        if (__ballot(S.resume > 0) != 0ull) {
            if (S.resume > 0) {
                HitPoint H;
                resolve_pending(L, tris, wr, S.pend, S.best, H);
                refresh_tbound(S);
            }
        }
        if (rounds >= limit) break;
        if (++guard > kLoopGuard) break; // never reached by a well-formed tree; bounds the loop so no wave can spin forever
    }
}

// A complete closest-hit query for every calling lane (ray-batch kernel).
template <bool STATS>
__device__ __forceinline__ void closest_hit_deferred(const Lds& L, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                                     const TriRecord* __restrict__ tris, const BvhNode* __restrict__ nodes, const Ray& wr,
                                                     Best& best, Counters& cnt)
{
    Segment S;
    if (STATS) probe_round(cnt.segment_rounds);
    begin_segment<STATS>(L, geoms, num_geoms, num_planes, tris, wr, S, cnt);
    traverse_budget<STATS>(L, tris, nodes, wr, S, cnt, 0, 64);
    while (S.base < L.last_base) { // > 32 geometries: the remaining chunks of records (uniform: every lane walks all chunks)
        S.base += 32;
        scan_chunk<STATS>(L, geoms, num_geoms, num_planes, tris, wr, S, cnt);
        traverse_budget<STATS>(L, tris, nodes, wr, S, cnt, 0, 64);
    }
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
    Lds lds;                  // else: LDS copy
    int g;
};

__device__ __forceinline__ float4 mat_f4(const MaterialRef& M, int k)
{
    if (M.global) return reinterpret_cast<const float4*>(M.global)[k];
    return lds_geom4(M.lds, M.g, k);
}
__device__ __forceinline__ int mat_bxdf(const MaterialRef& M)
{
    if (M.global) return M.global->bxdf_type;
    return lds_geom_i4(M.lds, M.g, 16).y;
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
    int bitem;     // where this (pixel, sample block)'s sum goes: block * pix_items + tile-major pixel item
    int send;      // one past the last sample of the block
    unsigned gxy;  // global pixel coordinates x | y << 16; the RNG counter is the pixel index y*W+x (kernel.cu:191)
    int s, b;      // current sample / segment
    float pdx, pdy, pdz; // primary direction of the pixel (every sample starts with the same ray: kernel.cu:200-205 has no jitter)
    Ray ray;       // current world-space ray
    float bx, by, bz; // throughput
    float ax, ay, az; // running sum over samples
    unsigned run_next, run_stride; // work items this lane still owns from its last queue fetch: run_next, run_next + run_stride, ...
    int run_left;
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

// Pull the next traceable pixel from the global queue for every calling lane (wave-wide ballot + prefix compaction
// among the lanes that execute the call: one atomic per wave and round).  Returns false for lanes that saw the end of
// the queue.
__device__ __forceinline__ bool acquire_pixel(const KParams& p, int lane, Path& P)
{
    bool got = false, exhausted = false;
    for (;;) {
        const bool want = !got && !exhausted;
        if (__ballot(want) == 0ull) break;
        // Lanes that used up their run fetch a new one: one atomic per wave for items_per_fetch items per asking lane,
        // dealt so that at every step the lanes of the fetch hold consecutive items (base + step * lanes + rank).  With
        // short items (few samples per pixel) a fetch per item would saturate the counter's memory channel.
        const bool fetch = want && P.run_left == 0;
        const unsigned long long m = __ballot(fetch);
        if (m != 0ull) {
            const unsigned cnt = (unsigned)__popcll(m);
            unsigned base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(p.queue, cnt * (unsigned)p.items_per_fetch);
            base = __shfl(base, leader);
            if (fetch) {
                P.run_next = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                P.run_stride = cnt;
                P.run_left = p.items_per_fetch;
            }
        }
        if (want) {
            const unsigned item = P.run_next;
            P.run_next += P.run_stride;
            --P.run_left;
            if (item >= p.total_items) {
                exhausted = true;
                P.run_left = 0;
            } else {
                // an item is one sample block of one pixel; pixels walk 8x8 tiles of the local image (padding items and
                // untraced pixels are consumed and skipped), blocks are the slow index
                // items [0, tail_first_item): (pixel, whole block); beyond: (pixel, sample group) of the tail block, group-major
                const bool tail = p.tail_block >= 0 && item >= p.tail_first_item;
                const unsigned rel = tail ? item - p.tail_first_item : item;
                const unsigned blk = rel / p.pix_items, pitem = rel - blk * p.pix_items; // tail: blk is the group index
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
                        P.bitem = ~(first * (int)p.pix_items + (int)pitem); // negative: slot of the next sample in tail_samples (sample-major)
                        P.s = p.tail_block * p.block_spp + first;
                        P.send = min(p.spp_total, p.tail_block * p.block_spp + past);
                    } else {
                        const int block = p.block_begin + (int)blk;
                        P.bitem = block * (int)p.pix_items + (int)pitem;
                        P.s = block * p.block_spp;
                        P.send = min(p.spp_total, P.s + p.block_spp);
                    }
                    P.ax = P.ay = P.az = 0.f;
                    primary_ray(p, P.gxy, P.ray); // once per (pixel, block); its samples reuse the direction
                    P.pdx = P.ray.dx;
                    P.pdy = P.ray.dy;
                    P.pdz = P.ray.dz;
                    start_sample(p, P);
                }
            }
        }
    }
    return got;
}

// Shade the finished segment and advance the path.  Returns true when the lane still owns its pixel (either the path
// continues with a new ray in P.ray, or the next sample's primary ray was generated), false when the pixel is finished.
// SPECULAR = false: the caller guarantees a scene without MIRROR / GLASS surfaces and their code drops out.
template <bool SPECULAR = true>
__device__ __forceinline__ bool shade_and_advance(const KParams& p, const Best& best, bool hit, const MaterialRef& M, Path& P)
{
    const bool debug_shade = p.shade_mode == FF_SHADE_NORMAL_DEBUG;
    bool path_done = true;
    // Radiance of the path: it is zero until the path ends on an emitter (the only light transport here), so it is not
    // carried across segments; "0 + beta*Le" of the integrator is beta*Le bit for bit.
    float Lx = 0.f, Ly = 0.f, Lz = 0.f;
    if (hit) {
        float nx, ny, nz;
        world_normal(M, best, debug_shade, nx, ny, nz);
        if (debug_shade) {
            // shade(), kernel.cu:178-184
            Lx = fabsf(nx); Ly = fabsf(ny); Lz = fabsf(nz);
        } else if (mat_bxdf(M) == FF_BXDF_EMITTER) {
            // utilities.h:96-103: two-sided emitter, m_emissiveColor * m_intensity
            const float4 emission = mat_f4(M, 13);
            Lx = 0.f + P.bx * emission.x;
            Ly = 0.f + P.by * emission.y;
            Lz = 0.f + P.bz * emission.z;
        } else {
            // MIRROR: perfect reflection, throughput *= m_specularColor (the record's tint slot holds it).  GLASS: smooth
            // dielectric, Fresnel-weighted choice between reflection and refraction (oracle/ff_oracle.c is the definition).
            // Everything else is diffuse (utilities.h:109): cosine-weighted sampling, so f*cos/pdf = albedo.
            const int bxdf = mat_bxdf(M);
            const bool mirror = SPECULAR && bxdf == FF_BXDF_MIRROR, glass = SPECULAR && bxdf == FF_BXDF_GLASS;
            const float4 albedo = mat_f4(M, 12);
            if (!glass) {
                P.bx = P.bx * albedo.x;
                P.by = P.by * albedo.y;
                P.bz = P.bz * albedo.z;
            }
            if (P.b != p.bounces - 1) {
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
                path_done = false;
            }
        }
    }
    if (!path_done) return true;
    if (P.bitem < 0) {
        // tail item: every sample is stored on its own; the combine pass adds the block's samples in order
        p.tail_samples[~P.bitem] = make_float4(Lx, Ly, Lz, 0.f);
        P.bitem -= (int)p.pix_items; // ~(slot + pix_items): the pixel's next sample
    } else {
        P.ax = P.ax + Lx;
        P.ay = P.ay + Ly;
        P.az = P.az + Lz;
    }
    ++P.s;
    if (P.s < P.send && !debug_shade) {
        start_sample(p, P);
        return true;
    }
    // sample block finished: its sum goes to the block buffer (the combine kernel adds a pixel's blocks in order)
    if (P.bitem >= 0) p.blocksums[P.bitem] = make_float4(P.ax, P.ay, P.az, 0.f);
    return false;
}

__device__ __forceinline__ void flush_counters(const KParams& p, int lane, const Counters& cnt, bool stats)
{
    // wave-reduced counters, one atomic per wave and counter
    const unsigned long long rays = wave_sum((unsigned long long)cnt.rays);
    if (lane == 0 && rays) atomicAdd(&p.counters[0], rays);
    if (stats) {
        const unsigned long long n = wave_sum((unsigned long long)cnt.nodes), t = wave_sum((unsigned long long)cnt.tris),
                                 pl = wave_sum((unsigned long long)cnt.planes);
        const unsigned long long r0 = wave_sum((unsigned long long)cnt.inner_rounds), r1 = wave_sum((unsigned long long)cnt.leaf_rounds),
                                 r2 = wave_sum((unsigned long long)cnt.tri_rounds), r3 = wave_sum((unsigned long long)cnt.plane_rounds),
                                 r4 = wave_sum((unsigned long long)cnt.segment_rounds), r5 = wave_sum((unsigned long long)cnt.no_mesh),
                                 r6 = wave_sum((unsigned long long)cnt.plane_exact);
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
        }
        {
            const unsigned long long u0 = wave_sum(cnt.t_start), u1 = wave_sum(cnt.t_inner), u2 = wave_sum(cnt.t_leaf);
            if (lane == 0) { atomicAdd(&p.counters[1 + 15], u0); atomicAdd(&p.counters[2 + 15], u1); atomicAdd(&p.counters[3 + 15], u2); }
            const unsigned long long w0 = wave_sum(cnt.t_b1), w1 = wave_sum(cnt.t_b2), w2 = wave_sum(cnt.t_b3);
            if (lane == 0) { atomicAdd(&p.counters[19], w0); atomicAdd(&p.counters[20], w1); atomicAdd(&p.counters[21], w2); }
        }
    }
}

__device__ __forceinline__ void init_path(Path& P)
{
    P.bitem = 0; P.send = 0; P.gxy = 0; P.s = 0; P.b = 0;
    P.pdx = P.pdy = 0.f; P.pdz = 1.f;
    P.ray = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    P.run_next = 0u;
    P.run_stride = 1u;
    P.run_left = 0;
    P.bx = P.by = P.bz = 1.f;
    P.ax = P.ay = P.az = 0.f;
}

// ---- the BVH mega-kernel -------------------------------------------------------------------------------------------------
//
// Segment-synchronous: in every round each live lane of the wave runs one complete closest-hit query and then
// resolves/shades/spawns together with its neighbours.  (A per-lane state machine that let finished lanes wait for a
// quorum while others kept traversing was measured slower on MI355X: the setup block is too large to run at partial
// occupancy, see DESIGN.md.)  Latency is hidden by occupancy: 1024 threads per workgroup = 4 waves per SIMD.

// EXTRAS = false is the instantiation for scenes made of what the reference itself renders (planes and meshes, diffuse
// and emitting surfaces): the plane/sphere boundary becomes a compile-time "never" and the MIRROR / GLASS branches of the
// shader drop out (together they cost the reference-like scenes 3.5 % otherwise, measured on one box).
template <bool STATS, int BLOCK, bool EXTRAS>
__global__ __launch_bounds__(BLOCK) void trace_bvh_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const Lds L = make_lds(p.lds_nodes, p.stack_depth, BLOCK, tid, EXTRAS ? p.num_quads : 0x7fffffff, EXTRAS ? p.trinormals : nullptr,
                           EXTRAS ? ((p.num_geoms - 1) >> 5) << 5 : 0);
    stage_scene(L, p.nodes, p.geoms, p.num_geoms, tid, BLOCK);

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    Path P;
    init_path(P);
    Segment S;
    S.best = { kInf, -1, -1 };
    S.pend = { kInf, -1, -1 };
    S.meshes = 0u;
    S.cur = kDone; S.sp = 0; S.mesh = -1; S.resume = 0;
    S.osr = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    S.ix = S.iy = S.iz = S.ox = S.oy = S.oz = 0.f;
    S.scale = 1.f;
    S.tbound = 0.f;
    S.base = 0;
    bool active = false, exhausted = false, inflight = false; // inflight: S holds a query of this lane (finished or not)
    // instrumented launches only: wave cycles per phase (s_memtime), [0] resolve [1] shade [2] acquire [3] begin [4] traverse
    unsigned long long tphase[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long wave_t0 = STATS ? wall_clock64() : 0ull;
    for (;;) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (STATS) t0 = __builtin_amdgcn_s_memtime();
        // Lanes whose query is finished (or that have none) resolve + shade + spawn together; lanes still traversing skip.
        if (EXTRAS && L.last_base > 0) {
            // scenes of more than 32 geometries: queries that finished a chunk of records move on to the next one
            const bool advance = inflight && chunk_done(S) && S.base < L.last_base;
            if (__ballot(advance) != 0ull && advance) {
                S.base += 32;
                scan_chunk<STATS>(L, p.geoms, p.num_geoms, p.num_planes, p.tris, P.ray, S, cnt);
            }
        }
        const bool setup = !inflight || segment_done(L, S);
        Best best;
        bool hit = false;
        if (setup && inflight) {
            finish_segment(L, p.tris, P.ray, S, best);
            hit = best.geom >= 0;
        }
        if (STATS) t1 = __builtin_amdgcn_s_memtime();
        if (setup && inflight) {
            MaterialRef M;
            M.global = nullptr;
            M.lds = L;
            M.g = best.geom;
            active = shade_and_advance<EXTRAS>(p, best, hit, M, P);
            inflight = false;
        }
        if (STATS) t2 = __builtin_amdgcn_s_memtime();
        if (setup && !active && !exhausted) {
            active = acquire_pixel(p, lane, P);
            exhausted = !active;
            if (STATS && exhausted) atomicMax(&p.counters[23], ~(unsigned long long)wall_clock64()); // (complemented) first lane to find the queue empty
        }
        if (STATS) t3 = __builtin_amdgcn_s_memtime();
        if (setup && active) {
            if (STATS) probe_round(cnt.segment_rounds);
            begin_segment<STATS>(L, p.geoms, p.num_geoms, p.num_planes, p.tris, P.ray, S, cnt);
            cnt.rays += 1;
            inflight = true;
        }
        if (STATS) t4 = __builtin_amdgcn_s_memtime();
        if (__ballot(inflight) == 0ull) break;
        // Time-sliced traversal: after `setup_threshold` inner-node rounds the finished lanes go and fetch new rays while the
        // long-tail lanes keep their state (per-ray traversal cost is heavy-tailed: a few rays need 10x the mean).
        if (inflight) traverse_budget<STATS>(L, p.tris, p.nodes, P.ray, S, cnt, p.setup_threshold, p.leaf_threshold);
        if (STATS) {
            const unsigned long long t5 = __builtin_amdgcn_s_memtime();
            tphase[0] += t1 - t0; tphase[1] += t2 - t1; tphase[2] += t3 - t2; tphase[3] += t4 - t3; tphase[4] += t5 - t4;
        }
    }
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

// ---- the BVH mega-kernel with a wave-private path pool ---------------------------------------------------------------
//
// The time-sliced kernel above still runs its two halves at partial occupancy: after a slice only the finished lanes
// shade (~50 %), and during a slice the finished lanes idle (~30 %).  Here every wave owns a POOL of P path slots
// (P = 3 x 64) whose state lives in a wave-private global workspace (112 B per slot, slot-major so that 64 consecutive
// slots are one coalesced access; it stays L2 / Infinity-Cache resident) and works in two kinds of steps:
//
//   S (setup)     64 lanes take 64 FINISHED slots: resolve the hit exactly, shade, spawn the next ray (or fetch a new
//                 pixel), screen the planes and the mesh boxes, and file the slot as READY (needs BVH traversal) or
//                 FINISHED again (planes only).  All lanes busy.
//   T (traverse)  lanes run BVH traversal for the slot they hold; a lane whose query completes writes the outcome to
//                 its slot, files it as FINISHED and immediately takes another READY slot, so the traversal loops stay
//                 populated although per-ray traversal cost is heavy-tailed.  A lane in the middle of a long query
//                 simply keeps its registers and LDS stack across S steps.
//
// The slot lists (READY / FINISHED) are wave-private arrays in LDS; only the owning wave touches its pool, so no atomics
// or barriers are involved.  Results are unchanged: a slot owns its pixel for all samples (sequential accumulation), the
// RNG is counter-based, and every hit is resolved with the exact reference arithmetic.

enum PoolWord : int {
    kRayO = 0, kRayD = 3, kBeta = 6, kAcc = 9, kBitem = 12, kGxy = 13, kSb = 14,
    kPendDist = 15, kPendGeom = 16, kPendRec = 17, kMeshes = 18, kBestDist = 19, kBestGeom = 20, kBestRec = 21, kFlags = 22, kSend = 23, kPrimD = 24,
    kPoolWords = 28
};
constexpr unsigned kSlotHasQuery = 1u, kSlotAlive = 2u;

struct Pool {
    unsigned* base; // this wave's workspace: word k of slot j at base[k * slots + j]
    int slots;
};
__device__ __forceinline__ float pool_f(const Pool& W, int j, int k) { return __uint_as_float(W.base[k * W.slots + j]); }
__device__ __forceinline__ unsigned pool_u(const Pool& W, int j, int k) { return W.base[k * W.slots + j]; }
__device__ __forceinline__ void pool_set_f(const Pool& W, int j, int k, float v) { W.base[k * W.slots + j] = __float_as_uint(v); }
__device__ __forceinline__ void pool_set_u(const Pool& W, int j, int k, unsigned v) { W.base[k * W.slots + j] = v; }

template <bool STATS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void trace_pool_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const Lds L = make_lds(p.lds_nodes, p.stack_depth, BLOCK, tid, p.num_quads, p.trinormals);
    stage_scene(L, p.nodes, p.geoms, p.num_geoms, tid, BLOCK);

    const int P = p.pool_slots;
    Pool W;
    W.slots = P;
    W.base = p.pool + ((size_t)blockIdx.x * (BLOCK / kWave) + (size_t)wave) * (size_t)P * kPoolWords;
    // wave-private slot lists in LDS, after the geometry records: READY then FINISHED, 16-bit slot ids
    unsigned short* lists = reinterpret_cast<unsigned short*>(ff_smem + L.geom_base + p.num_geoms * kGeomVec4) + (size_t)wave * 2 * P;
    unsigned short* ready = lists;
    unsigned short* finished = lists + P;
    for (int j = lane; j < P; j += kWave) {
        finished[j] = (unsigned short)j;
        pool_set_u(W, j, kFlags, 0u);
    }
    int n_ready = 0, n_finished = P; // wave-uniform

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int my_slot = -1; // slot whose query this lane is traversing
    Ray ray = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    Segment S;
    S.best = { kInf, -1, -1 };
    S.pend = { kInf, -1, -1 };
    S.meshes = 0u;
    S.cur = kDone; S.sp = 0; S.mesh = -1; S.resume = 0;
    S.osr = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    S.ix = S.iy = S.iz = S.ox = S.oy = S.oz = 0.f;
    S.scale = 1.f;
    S.tbound = 0.f;
    S.base = 0;

    for (int guard = 0; guard < (1 << 28); ++guard) {
        const int running = __popcll(__ballot(my_slot >= 0));
        // ---- S step: a full batch of finished slots is waiting, or nothing else can make progress -----------------------
        if (n_finished >= kWave || (n_finished > 0 && n_ready == 0 && running < p.pool_low)) {
            const int n = n_finished < kWave ? n_finished : kWave;
            n_finished -= n;
            bool to_ready = false, to_finished = false;
            int j = 0;
            if (lane < n) {
                j = finished[n_finished + lane];
                const unsigned flags = pool_u(W, j, kFlags);
                Path Q;
                init_path(Q);
                bool alive = (flags & kSlotAlive) != 0u;
                if (alive) {
                    Q.ray.ox = pool_f(W, j, kRayO); Q.ray.oy = pool_f(W, j, kRayO + 1); Q.ray.oz = pool_f(W, j, kRayO + 2);
                    Q.ray.dx = pool_f(W, j, kRayD); Q.ray.dy = pool_f(W, j, kRayD + 1); Q.ray.dz = pool_f(W, j, kRayD + 2);
                    Q.bx = pool_f(W, j, kBeta); Q.by = pool_f(W, j, kBeta + 1); Q.bz = pool_f(W, j, kBeta + 2);
                    Q.ax = pool_f(W, j, kAcc); Q.ay = pool_f(W, j, kAcc + 1); Q.az = pool_f(W, j, kAcc + 2);
                    Q.pdx = pool_f(W, j, kPrimD); Q.pdy = pool_f(W, j, kPrimD + 1); Q.pdz = pool_f(W, j, kPrimD + 2);
                    Q.bitem = (int)pool_u(W, j, kBitem);
                    Q.send = (int)pool_u(W, j, kSend);
                    Q.gxy = pool_u(W, j, kGxy);
                    const unsigned sb = pool_u(W, j, kSb);
                    Q.s = (int)(sb & 0xFFFFFFu);
                    Q.b = (int)(sb >> 24);
                }
                if (alive && (flags & kSlotHasQuery) != 0u) {
                    Segment R;
                    R.best.dist = pool_f(W, j, kBestDist); R.best.geom = (int)pool_u(W, j, kBestGeom); R.best.rec = (int)pool_u(W, j, kBestRec);
                    R.pend.dist = pool_f(W, j, kPendDist); R.pend.geom = (int)pool_u(W, j, kPendGeom); R.pend.rec = (int)pool_u(W, j, kPendRec);
                    Best best;
                    finish_segment(L, p.tris, Q.ray, R, best);
                    const bool hit = best.geom >= 0;
                    MaterialRef M;
                    M.global = nullptr;
                    M.lds = L;
                    M.g = best.geom;
                    alive = shade_and_advance(p, best, hit, M, Q);
                }
                if (!alive) alive = acquire_pixel(p, lane, Q);
                if (alive) {
                    Segment R;
                    if (STATS) probe_round(cnt.segment_rounds);
                    begin_segment<STATS>(L, p.geoms, p.num_geoms, p.num_planes, p.tris, Q.ray, R, cnt);
                    cnt.rays += 1;
                    pool_set_f(W, j, kRayO, Q.ray.ox); pool_set_f(W, j, kRayO + 1, Q.ray.oy); pool_set_f(W, j, kRayO + 2, Q.ray.oz);
                    pool_set_f(W, j, kRayD, Q.ray.dx); pool_set_f(W, j, kRayD + 1, Q.ray.dy); pool_set_f(W, j, kRayD + 2, Q.ray.dz);
                    pool_set_f(W, j, kBeta, Q.bx); pool_set_f(W, j, kBeta + 1, Q.by); pool_set_f(W, j, kBeta + 2, Q.bz);
                    pool_set_f(W, j, kAcc, Q.ax); pool_set_f(W, j, kAcc + 1, Q.ay); pool_set_f(W, j, kAcc + 2, Q.az);
                    pool_set_f(W, j, kPrimD, Q.pdx); pool_set_f(W, j, kPrimD + 1, Q.pdy); pool_set_f(W, j, kPrimD + 2, Q.pdz);
                    pool_set_u(W, j, kBitem, (unsigned)Q.bitem);
                    pool_set_u(W, j, kSend, (unsigned)Q.send);
                    pool_set_u(W, j, kGxy, Q.gxy);
                    pool_set_u(W, j, kSb, (unsigned)Q.s | ((unsigned)Q.b << 24));
                    pool_set_f(W, j, kPendDist, R.pend.dist); pool_set_u(W, j, kPendGeom, (unsigned)R.pend.geom); pool_set_u(W, j, kPendRec, (unsigned)R.pend.rec);
                    pool_set_f(W, j, kBestDist, R.best.dist); pool_set_u(W, j, kBestGeom, (unsigned)R.best.geom); pool_set_u(W, j, kBestRec, (unsigned)R.best.rec);
                    pool_set_u(W, j, kMeshes, R.meshes);
                    pool_set_u(W, j, kFlags, kSlotHasQuery | kSlotAlive);
                    to_ready = R.meshes != 0u;
                    to_finished = !to_ready;
                } else {
                    pool_set_u(W, j, kFlags, 0u); // queue drained: the slot retires
                }
            }
            // file the slots (wave-wide prefix compaction into the LDS lists)
            const unsigned long long m_r = __ballot(to_ready), m_f = __ballot(to_finished);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (to_ready) ready[n_ready + __popcll(m_r & below)] = (unsigned short)j;
            if (to_finished) finished[n_finished + __popcll(m_f & below)] = (unsigned short)j;
            n_ready += __popcll(m_r);
            n_finished += __popcll(m_f);
            continue;
        }
        if (running == 0 && n_ready == 0) break; // n_finished == 0 here: every slot has retired

        // ---- T step: refill idle lanes, traverse a slice, retire completed queries ------------------------------------
        {
            const bool idle = my_slot < 0;
            const unsigned long long m_idle = __ballot(idle);
            const int n_idle = __popcll(m_idle);
            if (n_ready > 0 && (n_idle >= p.pool_refill || running == 0)) {
                const int k = __popcll(m_idle & ((1ull << lane) - 1ull));
                if (idle && k < n_ready) {
                    const int j = ready[n_ready - 1 - k];
                    my_slot = j;
                    ray.ox = pool_f(W, j, kRayO); ray.oy = pool_f(W, j, kRayO + 1); ray.oz = pool_f(W, j, kRayO + 2);
                    ray.dx = pool_f(W, j, kRayD); ray.dy = pool_f(W, j, kRayD + 1); ray.dz = pool_f(W, j, kRayD + 2);
                    S.best.dist = pool_f(W, j, kBestDist); S.best.geom = (int)pool_u(W, j, kBestGeom); S.best.rec = (int)pool_u(W, j, kBestRec);
                    S.pend.dist = pool_f(W, j, kPendDist); S.pend.geom = (int)pool_u(W, j, kPendGeom); S.pend.rec = (int)pool_u(W, j, kPendRec);
                    S.meshes = pool_u(W, j, kMeshes);
                    S.cur = kDone; S.sp = 0; S.mesh = -1; S.resume = 0;
                }
                n_ready -= n_idle < n_ready ? n_idle : n_ready;
            }
        }
        if (my_slot >= 0) {
            traverse_budget<STATS>(L, p.tris, p.nodes, ray, S, cnt, p.setup_threshold, p.leaf_threshold);
            if (segment_done(L, S)) {
                const int j = my_slot;
                pool_set_f(W, j, kBestDist, S.best.dist); pool_set_u(W, j, kBestGeom, (unsigned)S.best.geom); pool_set_u(W, j, kBestRec, (unsigned)S.best.rec);
                pool_set_f(W, j, kPendDist, S.pend.dist); pool_set_u(W, j, kPendGeom, (unsigned)S.pend.geom); pool_set_u(W, j, kPendRec, (unsigned)S.pend.rec);
            }
        }
        {
            const bool done = my_slot >= 0 && segment_done(L, S);
            const unsigned long long m_done = __ballot(done);
            if (done) {
                finished[n_finished + __popcll(m_done & ((1ull << lane) - 1ull))] = (unsigned short)my_slot;
                my_slot = -1;
            }
            n_finished += __popcll(m_done);
        }
    }
    flush_counters(p, lane, cnt, STATS);
}

// ---- the brute-force mega-kernel (reference loop, validation path) ---------------------------------------------------

template <bool STATS>
__global__ __launch_bounds__(kBlockThreads) void trace_brute_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    float4* batch = reinterpret_cast<float4*>(ff_smem); // triangle batch buffer
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    Path P;
    init_path(P);
    bool active = false, exhausted = false;
    for (;;) {
        if (!active && !exhausted) {
            active = acquire_pixel(p, lane, P);
            exhausted = !active;
        }
        // every thread of the workgroup takes part in staging the triangle batches
        if (__syncthreads_or(active ? 1 : 0) == 0) break;
        Best best;
        closest_hit_brute<STATS>(p.geoms, p.num_geoms, p.tris, batch, active, P.ray, best, cnt, p.trinormals);
        if (!active) continue;
        const bool hit = best.geom >= 0;
        MaterialRef M;
        M.global = hit ? &p.geoms[best.geom] : p.geoms;
        M.lds = make_lds(0, 0, kBlockThreads, tid, 0);
        M.g = 0;
        active = shade_and_advance(p, best, hit, M, P);
    }
    flush_counters(p, lane, cnt, STATS);
}

// Batch closest-hit query: intersectRays (kernel.cu:127-176) for caller-supplied rays, one thread per ray.
template <int MODE>
__global__ __launch_bounds__(kBlockThreads) void ray_batch_kernel(const RayBatchParams p)
{
    const int tid = threadIdx.x;
    const Lds L = make_lds(p.lds_nodes, p.stack_depth, kBlockThreads, tid, p.num_quads, nullptr, ((p.num_geoms - 1) >> 5) << 5);
    if (MODE == FF_TRACE_BVH) stage_scene(L, p.nodes, p.geoms, p.num_geoms, tid, kBlockThreads);
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
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    if (MODE == FF_TRACE_BRUTE_FORCE) closest_hit_brute<false>(p.geoms, p.num_geoms, p.tris, batch, live, wr, best, cnt);
    else if (live) closest_hit_deferred<false>(L, p.geoms, p.num_geoms, p.num_planes, p.tris, p.nodes, wr, best, cnt);
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
        M.lds = L;
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
    for (int b = 0; b < p.num_blocks; ++b) {
        float4 v;
        if (b == p.tail_block) {
            // this block was traced sample by sample: the sequential sum a lane would have kept in registers
            float bx = 0.f, by = 0.f, bz = 0.f;
            const float4* sp = p.tail_samples + pitem; // sample-major: neighbouring threads read neighbouring values
            for (int i = 0; i < p.tail_samples_in_block; ++i) {
                const float4 l = sp[(size_t)i * p.pix_items];
                bx = bx + l.x;
                by = by + l.y;
                bz = bz + l.z;
            }
            v = make_float4(bx, by, bz, 0.f);
        } else {
            v = p.blocksums[(size_t)b * p.pix_items + pitem];
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

size_t bvh_lds_bytes(int lds_nodes, int stack_depth, int block_threads, int num_geoms)
{
    return (size_t)lds_nodes * sizeof(BvhNode) + (size_t)stack_depth * (size_t)block_threads * sizeof(unsigned) +
           (size_t)num_geoms * sizeof(GeomRecord);
}

size_t pool_list_bytes(int pool_slots, int block_threads) { return (size_t)(block_threads / 64) * 2 * (size_t)pool_slots * sizeof(unsigned short); }

size_t pool_workspace_bytes(int pool_slots, int grid_blocks, int block_threads)
{
    return (size_t)grid_blocks * (size_t)(block_threads / 64) * (size_t)pool_slots * 28 * sizeof(unsigned);
}

int max_lds_nodes(int stack_depth, int block_threads, int num_geoms)
{
    const long avail = (long)kLdsBudgetBytes - (long)stack_depth * (long)block_threads * (long)sizeof(unsigned) -
                       (long)num_geoms * (long)sizeof(GeomRecord);
    return avail > 0 ? (int)(avail / (long)sizeof(BvhNode)) : 0;
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
    FF_SET_LDS((trace_pool_kernel<false, 1024>))
    FF_SET_LDS((trace_pool_kernel<true, 1024>))
    FF_SET_LDS((ray_batch_kernel<FF_TRACE_BVH>))
#undef FF_SET_LDS
    return hipSuccess;
}

hipError_t launch_trace(const KParams& p, int trace_mode, bool collect_stats, int grid_blocks, int block_threads, hipStream_t stream,
                        const char** kernel_name)
{
    const dim3 grid(grid_blocks);
    const char* name = "";
    if (trace_mode == FF_TRACE_BVH) {
        const dim3 block(block_threads);
        if (p.pool != nullptr) {
            const size_t lds = bvh_lds_bytes(p.lds_nodes, p.stack_depth, 1024, p.num_geoms) + pool_list_bytes(p.pool_slots, 1024);
            if (collect_stats) hipLaunchKernelGGL((trace_pool_kernel<true, 1024>), grid, dim3(1024), lds, stream, p);
            else hipLaunchKernelGGL((trace_pool_kernel<false, 1024>), grid, dim3(1024), lds, stream, p);
            if (kernel_name) *kernel_name = collect_stats ? "trace_pool_kernel<true, 1024>" : "trace_pool_kernel<false, 1024>";
            return hipGetLastError();
        }
        const size_t lds = bvh_lds_bytes(p.lds_nodes, p.stack_depth, block_threads, p.num_geoms);
        const bool spheres = p.num_planes > p.num_quads || p.has_specular != 0 || p.trinormals != nullptr || p.num_geoms > 32; // any extra: the full kernel
#define FF_LAUNCH_BVH(B)                                                                                                  \
    do {                                                                                                                  \
        if (collect_stats) {                                                                                              \
            if (spheres) { hipLaunchKernelGGL((trace_bvh_kernel<true, B, true>), grid, block, lds, stream, p); name = "trace_bvh_kernel<true, " #B ", true>"; } \
            else { hipLaunchKernelGGL((trace_bvh_kernel<true, B, false>), grid, block, lds, stream, p); name = "trace_bvh_kernel<true, " #B ", false>"; } \
        } else {                                                                                                          \
            if (spheres) { hipLaunchKernelGGL((trace_bvh_kernel<false, B, true>), grid, block, lds, stream, p); name = "trace_bvh_kernel<false, " #B ", true>"; } \
            else { hipLaunchKernelGGL((trace_bvh_kernel<false, B, false>), grid, block, lds, stream, p); name = "trace_bvh_kernel<false, " #B ", false>"; } \
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
    const size_t lds = trace_mode == FF_TRACE_BVH ? bvh_lds_bytes(p.lds_nodes, p.stack_depth, kBlockThreads, p.num_geoms)
                                                  : (size_t)kBruteBatchTris * sizeof(TriRecord);
    const dim3 grid((p.n + kBlockThreads - 1) / kBlockThreads), block(kBlockThreads);
    if (trace_mode == FF_TRACE_BVH) hipLaunchKernelGGL((ray_batch_kernel<FF_TRACE_BVH>), grid, block, lds, stream, p);
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

} // namespace ff
