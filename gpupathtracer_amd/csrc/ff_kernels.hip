// ff_kernels.hip — the gfx950 (CDNA4, wave64) trace kernels.
//
// What the reference runs per pixel (kernel.cu:186-221: primary ray, brute-force closest hit over all geometries and
// triangles, shade, 8-bit store) is restructured here as a persistent mega-kernel:
//
//   * one workgroup of 512 threads per CU; the top of the BVH node array is staged ONCE per workgroup into LDS
//     (64-byte nodes, both child boxes per node) and every lane keeps its traversal stack in LDS (lane-strided, so
//     stack pushes/pops are bank-conflict free);
//   * lanes pull (pixel) work items from one global counter with a wave-wide ballot + prefix compaction, so a lane
//     whose paths have all terminated is refilled immediately instead of idling until its wave finishes;
//   * each loop iteration advances every live lane by one path segment (closest-hit query + shading); terminated
//     paths regenerate in place (next sample of the same pixel), which keeps the 64 lanes busy across bounces;
//   * the per-pixel camera matrix work of kernel.cu:203 is hoisted to the host; the per-hit 4x4 inverse of
//     kernel.cu:117 is hoisted to the scene compiler.
//
// Numerics: the file is compiled with -ffp-contract=off and IEEE-correct sqrt/divide.  Every value that decides or
// becomes part of a hit (object-space ray, Möller-Trumbore, world point, world distance, normal) is computed with
// the reference's / glm's exact operation order, so hits are bit-identical to the brute-force reference loop.  Only
// the BVH box tests use fused multiply-adds and an approximate reciprocal: they prune conservatively and never feed
// a result.
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

// Closest hit so far.  rec = TriRecord index for triangles, -1 for planes.
struct Best {
    float dist;
    int geom;
    int rec;
    float px, py, pz;
};

struct Counters {
    unsigned long long rays, nodes, tris, planes;
};

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    // glm dot(vec3): (x + y) + z  (GLM/detail/func_geometric.inl:52-53)
    const float px = ax * bx, py = ay * by, pz = az * bz;
    return (px + py) + pz;
}

// kernel.cu:138 — Ray(invM * vec4(o,1), normalize(invM * vec4(d,0))).  `len` is |invM*d| before normalisation: an
// object-space parameter t corresponds to the world distance t / len (for a unit world direction).
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
    len = sqrtf(dd);
    const float inv = 1.0f / len; // glm inversesqrt = 1 / sqrt
    o.dx = tx * inv;
    o.dy = ty * inv;
    o.dz = tz * inv;
}

// kernel.cu:110-125 on a candidate at object-space parameter t.  Ties on the world distance resolve like the
// reference's iteration order (lowest geometry index, then lowest triangle index), independent of the visiting order.
// `scale` = len / |world direction| converts a world distance into this geometry's object-space t (pruning only).
__device__ __forceinline__ void consider(const GeomRecord& G, int g, int rec, int orig_tri, float t, const Ray& osr, const Ray& wr,
                                         float scale, const GeomRecord* __restrict__ geoms, const TriRecord* __restrict__ tris, Best& best,
                                         float& tbound)
{
    const float Px = osr.ox + osr.dx * t, Py = osr.oy + osr.dy * t, Pz = osr.oz + osr.dz * t; // kernel.cu:99 / :16
    const float wx = (G.mod_c0[0] * Px + G.mod_c1[0] * Py) + (G.mod_c2[0] * Pz + G.mod_c3[0]); // kernel.cu:113
    const float wy = (G.mod_c0[1] * Px + G.mod_c1[1] * Py) + (G.mod_c2[1] * Pz + G.mod_c3[1]);
    const float wz = (G.mod_c0[2] * Px + G.mod_c1[2] * Py) + (G.mod_c2[2] * Pz + G.mod_c3[2]);
    const float vx = wr.ox - wx, vy = wr.oy - wy, vz = wr.oz - wz;
    const float d2 = (vx * vx + vy * vy) + vz * vz;
    // sqrt is monotonic: a squared distance clearly above the best one cannot win or tie; skip the IEEE sqrt for it
    if (d2 > best.dist * best.dist * 1.00001f) return;
    const float dist = sqrtf(d2); // glm distance, kernel.cu:114
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
        tbound = (dist * 1.001f + 1.0e-3f) * scale; // conservative object-space bound for pruning only
    }
}

// kernel.cu:35-108 (Möller-Trumbore, division deferred, back faces culled).  Returns the object-space t or -1.
__device__ __forceinline__ float triangle_t(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z, float v2x, float v2y,
                                            float v2z, const Ray& r)
{
    const float e1x = v1x - v0x, e1y = v1y - v0y, e1z = v1z - v0z; // :44
    const float e2x = v2x - v0x, e2y = v2y - v0y, e2z = v2z - v0z; // :45
    const float nx = e1y * e2z - e2y * e1z, ny = e1z * e2x - e2z * e1x, nz = e1x * e2y - e2x * e1y; // :48 glm cross
    if (dot3(r.dx, r.dy, r.dz, nx, ny, nz) > 0.0f) return -1.0f;                                        // :49
    const float px = r.dy * e2z - e2y * r.dz, py = r.dz * e2x - e2z * r.dx, pz = r.dx * e2y - e2x * r.dy; // :53
    const float det = dot3(e1x, e1y, e1z, px, py, pz);                                                   // :54
    if (det < kTriEpsilon) return -1.0f;                                                                 // :57
    const float tx = r.ox - v0x, ty = r.oy - v0y, tz = r.oz - v0z;                                       // :61
    const float u = dot3(tx, ty, tz, px, py, pz);                                                        // :62
    if (u < 0.0f || u > det) return -1.0f;                                                               // :64
    const float qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;            // :68
    const float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz);                                                  // :70
    if (v < 0.0f || u + v > det) return -1.0f;                                                           // :71
    float t = dot3(e2x, e2y, e2z, qx, qy, qz);                                                           // :75
    const float invDet = 1.0f / det; // :77 (a double division narrowed to float == the float division)
    t = t * invDet;                  // :79
    return t > kTriEpsilon ? t : -1.0f; // :97
}

__device__ __forceinline__ float triangle_t(const float4 a, const float4 b, const float4 c, const Ray& r)
{
    return triangle_t(a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, r);
}

// kernel.cu:8-32.  Returns t or -1.
__device__ __forceinline__ float plane_t(const GeomRecord& G, const Ray& r)
{
    const float nx = G.plane_n[0], ny = G.plane_n[1], nz = G.plane_n[2];
    const float denom = dot3(nx, ny, nz, r.dx, r.dy, r.dz); // :11
    if (!(fabsf(denom) >= kPlaneDenomMin)) return -1.0f;    // :12
    const float t = dot3(-r.ox, -r.oy, -r.oz, nx, ny, nz) / denom; // :14-15
    const float Px = r.ox + t * r.dx, Py = r.oy + t * r.dy;        // :16
    if (!(Px >= -0.5f && Px <= 0.5f && Py >= -0.5f && Py <= 0.5f)) return -1.0f; // :18
    return t > 0.0f ? t : -1.0f;                                   // :23
}

// ---- LDS layout of the BVH kernels -----------------------------------------------------------------------------------
//
//   [ nodes: lds_nodes x 64 B ][ traversal stacks: stack_depth x BLOCK x 4 B, lane-strided ][ geometry records: G x 288 B ]
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
};

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
        q0 = ff_smem[cur * 4];
        q1 = ff_smem[cur * 4 + 1];
        q2 = ff_smem[cur * 4 + 2];
        q3 = ff_smem[cur * 4 + 3];
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(nodes) + (size_t)cur * 4;
        q0 = p[0];
        q1 = p[1];
        q2 = p[2];
        q3 = p[3];
    }
}

// Geometry record gathered by a lane-varying index from LDS (the uniform-index path reads the global copy through
// scalar loads instead).
struct GeomXf {
    float i0x, i0y, i0z, z0, i1x, i1y, i1z, z1, i2x, i2y, i2z, z2, i3x, i3y, i3z; // inverse model columns + signed zeros
    float m0x, m0y, m0z, m1x, m1y, m1z, m2x, m2y, m2z, m3x, m3y, m3z;             // model columns
};

__device__ __forceinline__ void load_inverse(const Lds& L, int g, GeomXf& X)
{
    const float4 a = lds_geom4(L, g, 0), b = lds_geom4(L, g, 1), c = lds_geom4(L, g, 2), d = lds_geom4(L, g, 3);
    X.i0x = a.x; X.i0y = a.y; X.i0z = a.z; X.z0 = a.w;
    X.i1x = b.x; X.i1y = b.y; X.i1z = b.z; X.z1 = b.w;
    X.i2x = c.x; X.i2y = c.y; X.i2z = c.z; X.z2 = c.w;
    X.i3x = d.x; X.i3y = d.y; X.i3z = d.z;
}

// kernel.cu:138 with a lane-varying geometry (same arithmetic as object_space_ray).
__device__ __forceinline__ void object_space_ray_x(const GeomXf& X, const Ray& r, Ray& o, float& len)
{
    o.ox = (X.i0x * r.ox + X.i1x * r.oy) + (X.i2x * r.oz + X.i3x);
    o.oy = (X.i0y * r.ox + X.i1y * r.oy) + (X.i2y * r.oz + X.i3y);
    o.oz = (X.i0z * r.ox + X.i1z * r.oy) + (X.i2z * r.oz + X.i3z);
    const float tx = (X.i0x * r.dx + X.i1x * r.dy) + (X.i2x * r.dz + X.z0);
    const float ty = (X.i0y * r.dx + X.i1y * r.dy) + (X.i2y * r.dz + X.z1);
    const float tz = (X.i0z * r.dx + X.i1z * r.dy) + (X.i2z * r.dz + X.z2);
    const float dd = (tx * tx + ty * ty) + tz * tz;
    len = sqrtf(dd);
    const float inv = 1.0f / len;
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
__device__ __forceinline__ bool slab_may_hit(const float* wmin, const float* wmax, const WorldSlab& w, float limit)
{
    const float a0 = __builtin_fmaf(wmin[0], w.ix, w.ox), a1 = __builtin_fmaf(wmax[0], w.ix, w.ox);
    const float b0 = __builtin_fmaf(wmin[1], w.iy, w.oy), b1 = __builtin_fmaf(wmax[1], w.iy, w.oy);
    const float c0 = __builtin_fmaf(wmin[2], w.iz, w.oz), c1 = __builtin_fmaf(wmax[2], w.iz, w.oz);
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
//   * BVH traversal alternates wave-wide between an inner-node phase and a leaf phase, so each phase runs with most
//     lanes active instead of interleaving node visits and triangle tests lane by lane.

constexpr float kRel = 1.0e-4f, kAbs = 1.0e-4f; // screening margins, far above the rounding error of the fast forms

struct Pending {
    float dist; // approximate world distance, +inf when empty
    int geom;   // record index, -1 when empty
    int rec;    // TriRecord index, -1 for a plane
};

// Resolve the pending candidate with the exact reference arithmetic (kernel.cu:35-125) and merge it into `best`.
__device__ __forceinline__ void resolve_pending(const Lds& L, const GeomRecord* __restrict__ geoms, const TriRecord* __restrict__ tris,
                                                const Ray& wr, Pending& pend, Best& best)
{
    const int g = pend.geom, rec = pend.rec;
    pend.geom = -1;
    pend.dist = kInf;
    GeomXf X;
    load_inverse(L, g, X);
    Ray osr;
    float len;
    object_space_ray_x(X, wr, osr, len);
    float t;
    int orig_tri = -1;
    if (rec >= 0) {
        const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)rec * 3;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        orig_tri = __float_as_int(a.w);
        t = triangle_t(a, b, c, osr);
    } else {
        const float4 pn = lds_geom4(L, g, 11);
        // plane_t on gathered data (kernel.cu:8-32)
        const float denom = dot3(pn.x, pn.y, pn.z, osr.dx, osr.dy, osr.dz);
        t = -1.0f;
        if (fabsf(denom) >= kPlaneDenomMin) {
            const float tt = dot3(-osr.ox, -osr.oy, -osr.oz, pn.x, pn.y, pn.z) / denom;
            const float Px = osr.ox + tt * osr.dx, Py = osr.oy + tt * osr.dy;
            if (Px >= -0.5f && Px <= 0.5f && Py >= -0.5f && Py <= 0.5f && tt > 0.0f) t = tt;
        }
    }
    if (!(t > 0.0f)) return; // cannot happen for a screened candidate; kept so a wrong margin could not corrupt a result
    const float4 m0 = lds_geom4(L, g, 4), m1 = lds_geom4(L, g, 5), m2 = lds_geom4(L, g, 6), m3 = lds_geom4(L, g, 7);
    const float Px = osr.ox + osr.dx * t, Py = osr.oy + osr.dy * t, Pz = osr.oz + osr.dz * t; // kernel.cu:99 / :16
    const float wx = (m0.x * Px + m1.x * Py) + (m2.x * Pz + m3.x);                            // kernel.cu:113
    const float wy = (m0.y * Px + m1.y * Py) + (m2.y * Pz + m3.y);
    const float wz = (m0.z * Px + m1.z * Py) + (m2.z * Pz + m3.z);
    const float vx = wr.ox - wx, vy = wr.oy - wy, vz = wr.oz - wz;
    const float dist = sqrtf((vx * vx + vy * vy) + vz * vz); // kernel.cu:114
    bool take = dist < best.dist;                             // kernel.cu:115
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
        best.px = wx;
        best.py = wy;
        best.pz = wz;
    }
}

// Offer a certain hit at approximate world distance d to the lane's pending slot.
__device__ __forceinline__ void offer(const Lds& L, const GeomRecord* __restrict__ geoms, const TriRecord* __restrict__ tris, const Ray& wr,
                                      float d, int g, int rec, Pending& pend, Best& best)
{
    const float lim = fminf(best.dist, pend.dist);
    if (d > lim * (1.0f + kRel) + kAbs) return;                       // clearly farther than something already held
    if (pend.geom >= 0 && !(pend.dist > d * (1.0f + kRel) + kAbs))    // too close to rank approximately:
        resolve_pending(L, geoms, tris, wr, pend, best);              //   settle the held one exactly first
    pend.dist = d;
    pend.geom = g;
    pend.rec = rec;
}

template <bool STATS>
__device__ __forceinline__ void closest_hit_deferred(const Lds& L, const GeomRecord* __restrict__ geoms, int num_geoms, int num_planes,
                                                     const TriRecord* __restrict__ tris, const BvhNode* __restrict__ nodes, const Ray& wr,
                                                     Best& best, Counters& cnt)
{
    best.dist = kInf; // kernel.cu:131
    best.geom = -1;
    best.rec = -1;
    best.px = best.py = best.pz = 0.0f;
    Pending pend = { kInf, -1, -1 };
    const WorldSlab ws = make_world_slab(wr);

    // Geometries are handled in groups of 64 (one candidate bit per geometry); the reference's scenes have 5.
    for (int gbase = 0; gbase < num_geoms; gbase += 64) {
    const int gcount = min(64, num_geoms - gbase);

    // 1. screen every geometry's world box (records are stored planes first, then meshes)
    unsigned long long cand = 0ull;
    for (int j = 0; j < gcount; ++j) {
        const GeomRecord& G = geoms[gbase + j];
        if (slab_may_hit(G.wmin, G.wmax, ws, fminf(best.dist, pend.dist))) cand |= 1ull << j;
    }
    const int planes_here = max(0, min(64, num_planes - gbase));
    const unsigned long long plane_bits = planes_here >= 64 ? ~0ull : ((1ull << planes_here) - 1ull);

    // 2. planes: each lane walks its own candidates
    unsigned long long pm = cand & plane_bits;
    while (__ballot(pm != 0ull) != 0ull) {
        if (pm != 0ull) {
            const int g = gbase + __ffsll((long long)pm) - 1;
            pm &= pm - 1ull;
            const float4 bmin = lds_geom4(L, g, 14), bmax = lds_geom4(L, g, 15);
            const float wmin[3] = { bmin.x, bmin.y, bmin.z }, wmax[3] = { bmax.x, bmax.y, bmax.z };
            if (slab_may_hit(wmin, wmax, ws, fminf(best.dist, pend.dist))) {
                if (STATS) cnt.planes += 1;
                GeomXf X;
                load_inverse(L, g, X);
                Ray osr;
                float len;
                object_space_ray_x(X, wr, osr, len);
                const float4 pn = lds_geom4(L, g, 11);
                const float denom = dot3(pn.x, pn.y, pn.z, osr.dx, osr.dy, osr.dz);     // kernel.cu:11 (exact)
                if (fabsf(denom) >= kPlaneDenomMin) {                                    // kernel.cu:12 (exact)
                    const float num = dot3(-osr.ox, -osr.oy, -osr.oz, pn.x, pn.y, pn.z); // kernel.cu:14-15 numerator (exact)
                    float ta = num * __builtin_amdgcn_rcpf(denom);                       // approximate t
                    const float Pxa = __builtin_fmaf(ta, osr.dx, osr.ox), Pya = __builtin_fmaf(ta, osr.dy, osr.oy);
                    const float delta = 1.0e-5f * fmaxf(1.0f, fabsf(ta));
                    const float ex = fabsf(Pxa), ey = fabsf(Pya);
                    bool hit = false;
                    if (ex <= 0.5f - delta && ey <= 0.5f - delta && ta > 1.0e-30f) {
                        hit = true; // clearly inside the quad and in front of the origin
                    } else if (ex <= 0.5f + delta && ey <= 0.5f + delta && ta > -1.0e-30f) {
                        // within the margin of an edge (or t ~ 0): decide with the exact reference test
                        const float tt = num / denom;
                        const float Px = osr.ox + tt * osr.dx, Py = osr.oy + tt * osr.dy;
                        hit = Px >= -0.5f && Px <= 0.5f && Py >= -0.5f && Py <= 0.5f && tt > 0.0f;
                        ta = tt;
                    }
                    if (hit) offer(L, geoms, tris, wr, ta * __builtin_amdgcn_rcpf(len * ws.inv_len), g, -1, pend, best);
                }
            }
        }
    }

    // 3. meshes: lanes traverse the BVHs of their own candidate meshes; the wave alternates inner-node and leaf phases
    unsigned long long mm = cand & ~plane_bits;
    int cur = kDone, sp = 0, mg = -1;
    Ray osr = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    float ix = 0.f, iy = 0.f, iz = 0.f, ox = 0.f, oy = 0.f, oz = 0.f, scale = 1.f, wscale = 1.f;
    for (;;) {
        // (a) idle lanes start their next candidate mesh
        while (__ballot(cur == kDone && mm != 0ull) != 0ull) {
            if (cur == kDone && mm != 0ull) {
                const int g = gbase + __ffsll((long long)mm) - 1;
                mm &= mm - 1ull;
                const float4 bmin = lds_geom4(L, g, 14), bmax = lds_geom4(L, g, 15);
                const float wmin[3] = { bmin.x, bmin.y, bmin.z }, wmax[3] = { bmax.x, bmax.y, bmax.z };
                const int root = lds_geom_i4(L, g, 17).x;
                if (root >= 0 && slab_may_hit(wmin, wmax, ws, fminf(best.dist, pend.dist))) {
                    GeomXf X;
                    load_inverse(L, g, X);
                    float len;
                    object_space_ray_x(X, wr, osr, len);
                    ix = safe_rcp(osr.dx); iy = safe_rcp(osr.dy); iz = safe_rcp(osr.dz);
                    ox = -osr.ox * ix; oy = -osr.oy * iy; oz = -osr.oz * iz;
                    scale = len * ws.inv_len * 1.00001f;          // object-space t per unit of world distance
                    wscale = __builtin_amdgcn_rcpf(len * ws.inv_len); // world distance per unit of object-space t
                    mg = g;
                    cur = root;
                    sp = 0;
                }
            }
        }
        if (__ballot(cur != kDone) == 0ull) break;

        // (b) inner-node phase: runs until no lane of the wave sits on an inner node
        for (;;) {
            const bool inner = cur >= 0 && cur != kDone;
            if (__ballot(inner) == 0ull) break;
            if (inner) {
                uint4 q0, q1, q2, q3;
                fetch_node(L, nodes, cur, q0, q1, q2, q3);
                if (STATS) cnt.nodes += 1;
                const float tbound = (fminf(best.dist, pend.dist) * 1.001f + 1.0e-3f) * scale;
                // left box: q0.xyz = min, q1.xyz = max; right box: q2.xyz = min, q3.xyz = max (pruning only: FMA + approximate 1/d)
                float a0 = __builtin_fmaf(__uint_as_float(q0.x), ix, ox), a1 = __builtin_fmaf(__uint_as_float(q1.x), ix, ox);
                float b0 = __builtin_fmaf(__uint_as_float(q0.y), iy, oy), b1 = __builtin_fmaf(__uint_as_float(q1.y), iy, oy);
                float c0 = __builtin_fmaf(__uint_as_float(q0.z), iz, oz), c1 = __builtin_fmaf(__uint_as_float(q1.z), iz, oz);
                const float ln = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
                const float lf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tbound));
                a0 = __builtin_fmaf(__uint_as_float(q2.x), ix, ox); a1 = __builtin_fmaf(__uint_as_float(q3.x), ix, ox);
                b0 = __builtin_fmaf(__uint_as_float(q2.y), iy, oy); b1 = __builtin_fmaf(__uint_as_float(q3.y), iy, oy);
                c0 = __builtin_fmaf(__uint_as_float(q2.z), iz, oz); c1 = __builtin_fmaf(__uint_as_float(q3.z), iz, oz);
                const float rn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
                const float rf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tbound));
                const bool hl = ln <= lf * 1.000002f, hr = rn <= rf * 1.000002f;
                const int left = (int)q0.w, right = (int)q1.w;
                if (hl && hr) {
                    const bool swap = rn < ln;
                    stack_push(L, sp, swap ? left : right);
                    ++sp;
                    cur = swap ? right : left;
                } else if (hl) {
                    cur = left;
                } else if (hr) {
                    cur = right;
                } else if (sp > 0) {
                    --sp;
                    cur = stack_pop(L, sp);
                } else {
                    cur = kDone;
                }
            }
        }

        // (c) leaf phase: every lane holding a leaf tests its triangles, then takes the next entry off its stack
        if (cur < 0) {
            const int ref = ~cur;
            const int first = ref >> 3, count = (ref & 7) + 1;
            const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)first * 3;
            for (int k = 0; k < count; ++k) {
                const float4 A = tp[3 * k], B = tp[3 * k + 1], C = tp[3 * k + 2];
                if (STATS) cnt.tris += 1;
                // kernel.cu:44-75: exact up to the division; every accept/reject comparison is the reference's own
                const float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z;
                const float e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;
                const float nx = e1y * e2z - e2y * e1z, ny = e1z * e2x - e2z * e1x, nz = e1x * e2y - e2x * e1y;
                const float px = osr.dy * e2z - e2y * osr.dz, py = osr.dz * e2x - e2z * osr.dx, pz = osr.dx * e2y - e2x * osr.dy;
                const float det = dot3(e1x, e1y, e1z, px, py, pz);
                const float tx = osr.ox - A.x, ty = osr.oy - A.y, tz = osr.oz - A.z;
                const float u = dot3(tx, ty, tz, px, py, pz);
                const float qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;
                const float v = dot3(osr.dx, osr.dy, osr.dz, qx, qy, qz);
                const float tn = dot3(e2x, e2y, e2z, qx, qy, qz);
                bool ok = !(dot3(osr.dx, osr.dy, osr.dz, nx, ny, nz) > 0.0f) && !(det < kTriEpsilon) && !(u < 0.0f || u > det) &&
                          !(v < 0.0f || u + v > det);
                if (ok) {
                    float ta = tn * __builtin_amdgcn_rcpf(det); // approximate t (kernel.cu:77-79 is exact: 1/det, then multiply)
                    if (ta < kTriEpsilon * 1.001f) {
                        if (ta > kTriEpsilon * 0.999f) {
                            ta = tn * (1.0f / det); // within the margin of the t > EPSILON test: decide exactly (kernel.cu:97)
                            ok = ta > kTriEpsilon;
                        } else {
                            ok = false;
                        }
                    }
                    if (ok) offer(L, geoms, tris, wr, ta * wscale, mg, first + k, pend, best);
                }
            }
            if (sp > 0) {
                --sp;
                cur = stack_pop(L, sp);
            } else {
                cur = kDone;
            }
        }
    }

    } // geometry groups

    // 4. settle what is still pending (the common case: one exact evaluation per ray, all hitting lanes together)
    if (pend.geom >= 0) resolve_pending(L, geoms, tris, wr, pend, best);
    cnt.rays += 1;
}

// Brute-force closest hit: the reference's loop (kernel.cu:133-155) with the triangle array streamed through LDS in
// batches that the whole workgroup stages with coalesced 16-byte loads and then reads at a wave-uniform address.
// Must be called by every thread of the workgroup (it contains barriers); `live` masks lanes without a ray.
template <bool STATS>
__device__ __forceinline__ void closest_hit_brute(const GeomRecord* __restrict__ geoms, int num_geoms, const TriRecord* __restrict__ tris,
                                                  float4* batch, bool live, const Ray& wr, Best& best, Counters& cnt)
{
    best.dist = kInf;
    best.geom = -1;
    best.rec = -1;
    best.px = best.py = best.pz = 0.0f;
    for (int g = 0; g < num_geoms; ++g) {
        const GeomRecord& G = geoms[g];
        Ray osr;
        float len;
        object_space_ray(G, wr, osr, len);
        if (G.type == FF_GEOM_TRIANGLEMESH) {
            float tbound = kInf;
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
                        if (t > 0.0f) consider(G, g, G.tri_first + base + k, __float_as_int(a.w), t, osr, wr, len, geoms, tris, best, tbound);
                    }
                    if (STATS) cnt.tris += (unsigned long long)nb;
                }
            }
        } else if (live) {
            if (STATS) cnt.planes += 1;
            const float t = plane_t(G, osr);
            float tb = kInf;
            if (t > 0.0f) consider(G, g, -1, -1, t, osr, wr, len, geoms, tris, best, tb);
        }
    }
    if (live) cnt.rays += 1;
}

// World-space normal of the closest hit: inverse(transpose(M)) * vec4(n_obj, 0)  (kernel.cu:117), with
// n_obj = normalize(cross(e1, e2)) for triangles (kernel.cu:101) or the plane's m_normal (kernel.cu:26).
__device__ __forceinline__ void world_normal(const GeomRecord& G, const TriRecord* __restrict__ tris, int rec, float& nx, float& ny, float& nz)
{
    float ox, oy, oz;
    if (rec >= 0) {
        const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)rec * 3;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
        const float cx = e1y * e2z - e2y * e1z, cy = e1z * e2x - e2z * e1x, cz = e1x * e2y - e2x * e1y;
        const float inv = 1.0f / sqrtf(dot3(cx, cy, cz, cx, cy, cz));
        ox = cx * inv;
        oy = cy * inv;
        oz = cz * inv;
    } else {
        ox = G.plane_n[0];
        oy = G.plane_n[1];
        oz = G.plane_n[2];
    }
    nx = (G.nrm_c0[0] * ox + G.nrm_c1[0] * oy) + (G.nrm_c2[0] * oz + G.nrm_c0[3]);
    ny = (G.nrm_c0[1] * ox + G.nrm_c1[1] * oy) + (G.nrm_c2[1] * oz + G.nrm_c1[3]);
    nz = (G.nrm_c0[2] * ox + G.nrm_c1[2] * oy) + (G.nrm_c2[2] * oz + G.nrm_c2[3]);
}

// world_normal with the geometry record gathered from LDS (BVH kernels)
__device__ __forceinline__ void world_normal_lds(const Lds& L, int g, const TriRecord* __restrict__ tris, int rec, float& nx, float& ny, float& nz)
{
    float ox, oy, oz;
    if (rec >= 0) {
        const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)rec * 3;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
        const float cx = e1y * e2z - e2y * e1z, cy = e1z * e2x - e2z * e1x, cz = e1x * e2y - e2x * e1y;
        const float inv = 1.0f / sqrtf(dot3(cx, cy, cz, cx, cy, cz));
        ox = cx * inv;
        oy = cy * inv;
        oz = cz * inv;
    } else {
        const float4 pn = lds_geom4(L, g, 11);
        ox = pn.x;
        oy = pn.y;
        oz = pn.z;
    }
    const float4 n0 = lds_geom4(L, g, 8), n1 = lds_geom4(L, g, 9), n2 = lds_geom4(L, g, 10);
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
        const unsigned hi = __umulhi(0xD256D193u, c0), lo = 0xD256D193u * c0;
        c0 = hi ^ key ^ c1;
        c1 = lo;
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
    const float r = sqrtf(u1);
    x = r * cs;
    y = r * sn;
    z = sqrtf(fmaxf(0.0f, 1.0f - u1));
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

// ---- the mega-kernel ----------------------------------------------------------------------------------------------

template <int MODE, bool STATS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void trace_kernel(const KParams p)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);

    Lds L;
    L.node_count = p.lds_nodes;
    L.stride = BLOCK;
    L.stack_base = p.lds_nodes * 16 + tid;
    L.geom_base = p.lds_nodes * 4 + (p.stack_depth * BLOCK) / 4;
    if (MODE == FF_TRACE_BVH) {
        // Stage the top of the BVH and the geometry records once per workgroup: coalesced 16-byte loads, 1 KiB per
        // wave-instruction.  The workgroup is persistent, so this is paid once per launch, not per ray.
        const uint4* src = reinterpret_cast<const uint4*>(p.nodes);
        for (int i = tid; i < p.lds_nodes * 4; i += BLOCK) ff_smem[i] = src[i];
        const uint4* gsrc = reinterpret_cast<const uint4*>(p.geoms);
        for (int i = tid; i < p.num_geoms * kGeomVec4; i += BLOCK) ff_smem[L.geom_base + i] = gsrc[i];
        __syncthreads();
    }
    float4* batch = reinterpret_cast<float4*>(ff_smem); // brute-force mode: triangle batch buffer

    Counters cnt = { 0, 0, 0, 0 };

    // per-lane path state
    bool active = false, exhausted = false;
    int lpix = 0;          // local pixel index (row-major in the local image)
    unsigned gpix = 0;     // global pixel index y*W+x (kernel.cu:191), the RNG counter
    int s = 0, b = 0;      // current sample / segment
    float pdx = 0.f, pdy = 0.f, pdz = 0.f; // primary direction of the pixel (no jitter: kernel.cu:200-205 uses the pixel corner)
    Ray ray = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f };
    float bx = 1.f, by = 1.f, bz = 1.f; // path throughput
    float Lx = 0.f, Ly = 0.f, Lz = 0.f; // radiance of the current path
    float ax = 0.f, ay = 0.f, az = 0.f; // running sum over samples
    const bool debug_shade = p.shade_mode == FF_SHADE_NORMAL_DEBUG;

    for (;;) {
        // ---- refill: lanes without a pixel pull the next work items (wave-wide ballot + prefix compaction) ----
        // Repeats until every lane of the wave either owns a traceable pixel or has seen the end of the queue (items that
        // fall on tile padding or outside the traced region are consumed and skipped).
        for (;;) {
            const bool need = !active && !exhausted;
            const unsigned long long need_mask = __ballot(need);
            if (need_mask == 0ull) break;
            unsigned base = 0;
            const int leader = __ffsll((long long)need_mask) - 1;
            if (lane == leader) base = atomicAdd(p.queue, (unsigned)__popcll(need_mask));
            base = __shfl(base, leader);
            if (need) {
                const unsigned item = base + (unsigned)__popcll(need_mask & ((1ull << lane) - 1ull));
                if (item >= p.total_items) {
                    exhausted = true;
                } else {
                    const int tile = (int)(item >> 6), in = (int)(item & 63u);
                    const int lx = (tile % p.tiles_per_row) * 8 + (in & 7);
                    const int ly = (tile / p.tiles_per_row) * 8 + (in >> 3);
                    const int strip = ly / p.strip_rows;
                    const int gy = (strip * p.num_parts + p.part) * p.strip_rows + (ly - strip * p.strip_rows);
                    if (lx < p.xlim && ly < p.local_rows && gy < p.ylim) {
                        active = true;
                        lpix = ly * p.width + lx;
                        gpix = (unsigned)(gy * p.width + lx);
                        // kernel.cu:200-205
                        const float Px = ((float)lx / p.screen_w) * 2.f - 1.f;
                        const float Py = 1.f - ((float)gy / p.screen_h) * 2.f;
                        const float v0 = Px * p.far_clip, v1 = Py * p.far_clip, v2 = 1.f * p.far_clip, v3 = 1.f * p.far_clip;
                        const float wx = (p.cam_c0[0] * v0 + p.cam_c1[0] * v1) + (p.cam_c2[0] * v2 + p.cam_c3[0] * v3);
                        const float wy = (p.cam_c0[1] * v0 + p.cam_c1[1] * v1) + (p.cam_c2[1] * v2 + p.cam_c3[1] * v3);
                        const float wz = (p.cam_c0[2] * v0 + p.cam_c1[2] * v1) + (p.cam_c2[2] * v2 + p.cam_c3[2] * v3);
                        const float ddx = wx - p.cam_pos[0], ddy = wy - p.cam_pos[1], ddz = wz - p.cam_pos[2];
                        const float inv = 1.0f / sqrtf(dot3(ddx, ddy, ddz, ddx, ddy, ddz));
                        pdx = ddx * inv;
                        pdy = ddy * inv;
                        pdz = ddz * inv;
                        s = p.spp_begin;
                        b = 0;
                        ray.ox = p.cam_pos[0]; ray.oy = p.cam_pos[1]; ray.oz = p.cam_pos[2];
                        ray.dx = pdx; ray.dy = pdy; ray.dz = pdz;
                        bx = by = bz = 1.f;
                        Lx = Ly = Lz = 0.f;
                        if (p.first_chunk) {
                            ax = ay = az = 0.f;
                        } else {
                            const float4 prev = reinterpret_cast<const float4*>(p.accum)[lpix];
                            ax = prev.x; ay = prev.y; az = prev.z;
                        }
                    }
                }
            }
        }
        bool any_active;
        if (MODE == FF_TRACE_BRUTE_FORCE) any_active = __syncthreads_or(active ? 1 : 0) != 0;
        else any_active = __ballot(active) != 0ull;
        if (!any_active) {
            // In BVH mode a wave leaves once the queue is drained and all its lanes are done.  (A wave with some lanes
            // waiting for work cannot get here: `need` lanes were refilled or marked exhausted above.)
            break;
        }

        // ---- one path segment for every live lane ----
        Best best;
        if (MODE == FF_TRACE_BRUTE_FORCE) {
            closest_hit_brute<STATS>(p.geoms, p.num_geoms, p.tris, batch, active, ray, best, cnt);
        } else if (active) {
            closest_hit_deferred<STATS>(L, p.geoms, p.num_geoms, p.num_planes, p.tris, p.nodes, ray, best, cnt);
        }
        if (!active) continue;

        bool path_done = true;
        if (best.geom >= 0) {
            float nx, ny, nz;
            int bxdf_type;
            float4 albedo, emission;
            if (MODE == FF_TRACE_BVH) {
                world_normal_lds(L, best.geom, p.tris, best.rec, nx, ny, nz);
                bxdf_type = lds_geom_i4(L, best.geom, 16).y;
                albedo = lds_geom4(L, best.geom, 12);
                emission = lds_geom4(L, best.geom, 13);
            } else {
                const GeomRecord& G = p.geoms[best.geom];
                world_normal(G, p.tris, best.rec, nx, ny, nz);
                bxdf_type = G.bxdf_type;
                albedo = make_float4(G.albedo[0], G.albedo[1], G.albedo[2], 0.f);
                emission = make_float4(G.emission[0], G.emission[1], G.emission[2], 0.f);
            }
            if (debug_shade) {
                // shade(), kernel.cu:178-184
                Lx = fabsf(nx); Ly = fabsf(ny); Lz = fabsf(nz);
            } else if (bxdf_type == FF_BXDF_EMITTER) {
                // utilities.h:96-103: two-sided emitter, m_emissiveColor * m_intensity
                Lx = Lx + bx * emission.x;
                Ly = Ly + by * emission.y;
                Lz = Lz + bz * emission.z;
            } else {
                // everything else is diffuse (utilities.h:109); cosine-weighted sampling, so f*cos/pdf = albedo
                bx = bx * albedo.x;
                by = by * albedo.y;
                bz = bz * albedo.z;
                if (b != p.bounces - 1) {
                    const float ninv = 1.0f / sqrtf(dot3(nx, ny, nz, nx, ny, nz));
                    float ux = nx * ninv, uy = ny * ninv, uz = nz * ninv;
                    if (dot3(ux, uy, uz, ray.dx, ray.dy, ray.dz) > 0.0f) { ux = -ux; uy = -uy; uz = -uz; }
                    unsigned r0, r1;
                    philox2x32_10(gpix, ((unsigned)s << 8) | ((unsigned)b & 0xFFu), p.key, r0, r1);
                    const float u1 = (float)(r0 >> 8) * 5.9604644775390625e-08f;
                    float wlx, wly, wlz;
                    cosine_sample(u1, r1 >> 8, wlx, wly, wlz);
                    // orthonormal basis (Duff et al. 2017)
                    const float sign = copysignf(1.0f, uz);
                    const float aa = -1.0f / (sign + uz);
                    const float bb = (ux * uy) * aa;
                    const float t0 = 1.0f + ((sign * ux) * ux) * aa, t1 = sign * bb, t2 = -sign * ux;
                    const float s0 = bb, s1 = sign + (uy * uy) * aa, s2 = -uy;
                    const float wox = (t0 * wlx + s0 * wly) + ux * wlz;
                    const float woy = (t1 * wlx + s1 * wly) + uy * wlz;
                    const float woz = (t2 * wlx + s2 * wly) + uz * wlz;
                    const float winv = 1.0f / sqrtf(dot3(wox, woy, woz, wox, woy, woz));
                    ray.ox = best.px + ux * kRayEps;
                    ray.oy = best.py + uy * kRayEps;
                    ray.oz = best.pz + uz * kRayEps;
                    ray.dx = wox * winv;
                    ray.dy = woy * winv;
                    ray.dz = woz * winv;
                    ++b;
                    path_done = false;
                }
            }
        }
        if (path_done) {
            ax = ax + Lx;
            ay = ay + Ly;
            az = az + Lz;
            ++s;
            if (s >= p.spp_end || debug_shade) {
                // pixel finished for this launch
                if (p.last_chunk) {
                    float rx, ry, rz;
                    if (debug_shade) {
                        rx = ax; ry = ay; rz = az;
                    } else {
                        const float inv = 1.0f / (float)p.spp_total;
                        rx = ax * inv; ry = ay * inv; rz = az * inv;
                    }
                    if (p.radiance) {
                        p.radiance[3 * (size_t)lpix] = rx;
                        p.radiance[3 * (size_t)lpix + 1] = ry;
                        p.radiance[3 * (size_t)lpix + 2] = rz;
                    }
                    if (p.rgb8) {
                        p.rgb8[3 * (size_t)lpix] = to_u8(rx);
                        p.rgb8[3 * (size_t)lpix + 1] = to_u8(ry);
                        p.rgb8[3 * (size_t)lpix + 2] = to_u8(rz);
                    }
                } else {
                    reinterpret_cast<float4*>(p.accum)[lpix] = make_float4(ax, ay, az, 0.f);
                }
                active = false;
            } else {
                b = 0;
                ray.ox = p.cam_pos[0]; ray.oy = p.cam_pos[1]; ray.oz = p.cam_pos[2];
                ray.dx = pdx; ray.dy = pdy; ray.dz = pdz;
                bx = by = bz = 1.f;
                Lx = Ly = Lz = 0.f;
            }
        }
    }

    // wave-reduced counters, one atomic per wave and counter
    const unsigned long long rays = wave_sum(cnt.rays);
    if (lane == 0 && rays) atomicAdd(&p.counters[0], rays);
    if (STATS) {
        const unsigned long long n = wave_sum(cnt.nodes), t = wave_sum(cnt.tris), pl = wave_sum(cnt.planes);
        if (lane == 0) {
            if (n) atomicAdd(&p.counters[1], n);
            if (t) atomicAdd(&p.counters[2], t);
            if (pl) atomicAdd(&p.counters[3], pl);
        }
    }
}

// Batch closest-hit query: intersectRays (kernel.cu:127-176) for caller-supplied rays, one thread per ray.
template <int MODE>
__global__ __launch_bounds__(kBlockThreads) void ray_batch_kernel(const RayBatchParams p)
{
    const int tid = threadIdx.x;
    Lds L;
    L.node_count = p.lds_nodes;
    L.stride = kBlockThreads;
    L.stack_base = p.lds_nodes * 16 + tid;
    L.geom_base = p.lds_nodes * 4 + (p.stack_depth * kBlockThreads) / 4;
    if (MODE == FF_TRACE_BVH) {
        const uint4* src = reinterpret_cast<const uint4*>(p.nodes);
        for (int i = tid; i < p.lds_nodes * 4; i += kBlockThreads) ff_smem[i] = src[i];
        const uint4* gsrc = reinterpret_cast<const uint4*>(p.geoms);
        for (int i = tid; i < p.num_geoms * kGeomVec4; i += kBlockThreads) ff_smem[L.geom_base + i] = gsrc[i];
        __syncthreads();
    }
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
    best.dist = kInf; best.geom = -1; best.rec = -1; best.px = best.py = best.pz = 0.f;
    Counters cnt = { 0, 0, 0, 0 };
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
        float nx, ny, nz;
        world_normal(G, p.tris, best.rec, nx, ny, nz);
        out.m_intersectionPoint.x = best.px; out.m_intersectionPoint.y = best.py; out.m_intersectionPoint.z = best.pz;
        out.m_normal.x = nx; out.m_normal.y = ny; out.m_normal.z = nz;
        out.m_t = best.dist;   // kernel.cu:119: the world distance
        out.m_hit = 1;
        out.geometryIndex = G.orig_index;
        out.triangleIndex = best.rec >= 0 ? p.tris[best.rec].orig_index : -1;
    }
    p.out[i] = out;
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

} // namespace

size_t bvh_lds_bytes(int lds_nodes, int stack_depth, int block_threads, int num_geoms)
{
    return (size_t)lds_nodes * sizeof(BvhNode) + (size_t)stack_depth * (size_t)block_threads * sizeof(unsigned) +
           (size_t)num_geoms * sizeof(GeomRecord);
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
    // Only the BVH kernels go past the 64 KiB default (node cache + stacks); the brute-force kernels use a 48 KiB batch
    // buffer plus a little static LDS, and asking for the full 160 KiB on top of static LDS is rejected.
    FF_SET_LDS((trace_kernel<FF_TRACE_BVH, false, 512>))
    FF_SET_LDS((trace_kernel<FF_TRACE_BVH, true, 512>))
    FF_SET_LDS((trace_kernel<FF_TRACE_BVH, false, 1024>))
    FF_SET_LDS((trace_kernel<FF_TRACE_BVH, true, 1024>))
    FF_SET_LDS((ray_batch_kernel<FF_TRACE_BVH>))
#undef FF_SET_LDS
    return hipSuccess;
}

hipError_t launch_trace(const KParams& p, int trace_mode, bool collect_stats, int grid_blocks, int block_threads, hipStream_t stream)
{
    const dim3 grid(grid_blocks), block(block_threads);
    if (trace_mode == FF_TRACE_BVH) {
        const size_t lds = bvh_lds_bytes(p.lds_nodes, p.stack_depth, block_threads, p.num_geoms);
        if (block_threads == 1024) {
            if (collect_stats) hipLaunchKernelGGL((trace_kernel<FF_TRACE_BVH, true, 1024>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((trace_kernel<FF_TRACE_BVH, false, 1024>), grid, block, lds, stream, p);
        } else {
            if (collect_stats) hipLaunchKernelGGL((trace_kernel<FF_TRACE_BVH, true, 512>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((trace_kernel<FF_TRACE_BVH, false, 512>), grid, block, lds, stream, p);
        }
    } else {
        const size_t lds = (size_t)kBruteBatchTris * sizeof(TriRecord);
        if (collect_stats) hipLaunchKernelGGL((trace_kernel<FF_TRACE_BRUTE_FORCE, true, 512>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((trace_kernel<FF_TRACE_BRUTE_FORCE, false, 512>), grid, block, lds, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_ray_batch(const RayBatchParams& p, int trace_mode, hipStream_t stream)
{
    if (p.n <= 0) return hipSuccess;
    const size_t lds = trace_mode == FF_TRACE_BVH ? bvh_lds_bytes(p.lds_nodes, p.stack_depth, kBlockThreads, p.num_geoms) : (size_t)kBruteBatchTris * sizeof(TriRecord);
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

} // namespace ff
