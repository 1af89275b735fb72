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
// reference's iteration order (first geometry, then lowest triangle index), independent of the visiting order.
__device__ __forceinline__ void consider(const GeomRecord& G, int g, int rec, int orig_tri, float t, const Ray& osr, const Ray& wr,
                                         float len, const TriRecord* __restrict__ tris, Best& best, float& tbound)
{
    const float Px = osr.ox + osr.dx * t, Py = osr.oy + osr.dy * t, Pz = osr.oz + osr.dz * t; // kernel.cu:99 / :16
    const float wx = (G.mod_c0[0] * Px + G.mod_c1[0] * Py) + (G.mod_c2[0] * Pz + G.mod_c3[0]); // kernel.cu:113
    const float wy = (G.mod_c0[1] * Px + G.mod_c1[1] * Py) + (G.mod_c2[1] * Pz + G.mod_c3[1]);
    const float wz = (G.mod_c0[2] * Px + G.mod_c1[2] * Py) + (G.mod_c2[2] * Pz + G.mod_c3[2]);
    const float vx = wr.ox - wx, vy = wr.oy - wy, vz = wr.oz - wz;
    const float dist = sqrtf((vx * vx + vy * vy) + vz * vz); // glm distance, kernel.cu:114
    bool take = dist < best.dist;                             // kernel.cu:115
    if (!take && dist == best.dist && best.geom == g && rec >= 0 && best.rec >= 0) take = orig_tri < tris[best.rec].orig_index;
    if (take) {
        best.dist = dist;
        best.geom = g;
        best.rec = rec;
        best.px = wx;
        best.py = wy;
        best.pz = wz;
        tbound = (dist * 1.001f + 1.0e-3f) * len; // conservative object-space bound for pruning only
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

// ---- BVH traversal of one mesh (object space) ------------------------------------------------------------------

struct Traversal {
    const uint4* lds_nodes;   // staged nodes [0, lds_count)
    int lds_count;
    unsigned* stack;          // this lane's stack base in LDS
    int stride;               // distance between consecutive stack entries of one lane (= block size)
};

template <bool STATS>
__device__ __forceinline__ void traverse_mesh(const GeomRecord& G, int g, const Ray& osr, const Ray& wr, float len,
                                              const TriRecord* __restrict__ tris, const BvhNode* __restrict__ nodes,
                                              const Traversal& T, Best& best, Counters& cnt)
{
    // Box tests only prune: approximate reciprocal + FMA form, inflated far plane, padded boxes.
    const float sdx = fabsf(osr.dx) < 1e-30f ? copysignf(1e-30f, osr.dx) : osr.dx;
    const float sdy = fabsf(osr.dy) < 1e-30f ? copysignf(1e-30f, osr.dy) : osr.dy;
    const float sdz = fabsf(osr.dz) < 1e-30f ? copysignf(1e-30f, osr.dz) : osr.dz;
    const float ix = __builtin_amdgcn_rcpf(sdx), iy = __builtin_amdgcn_rcpf(sdy), iz = __builtin_amdgcn_rcpf(sdz);
    const float ox = -osr.ox * ix, oy = -osr.oy * iy, oz = -osr.oz * iz;
    float tbound = (best.dist * 1.001f + 1.0e-3f) * len;

    int cur = G.bvh_root;
    int sp = 0;
    for (;;) {
        bool pop = true;
        if (cur >= 0) {
            uint4 q0, q1, q2, q3;
            if (cur < T.lds_count) {
                const uint4* p = T.lds_nodes + (size_t)cur * 4;
                q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
            } else {
                const uint4* p = reinterpret_cast<const uint4*>(nodes) + (size_t)cur * 4;
                q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
            }
            if (STATS) cnt.nodes += 1;
            // left box: q0.xyz = min, q1.xyz = max; right box: q2.xyz = min, q3.xyz = max
            float a0 = __builtin_fmaf(__uint_as_float(q0.x), ix, ox), a1 = __builtin_fmaf(__uint_as_float(q1.x), ix, ox);
            float b0 = __builtin_fmaf(__uint_as_float(q0.y), iy, oy), b1 = __builtin_fmaf(__uint_as_float(q1.y), iy, oy);
            float c0 = __builtin_fmaf(__uint_as_float(q0.z), iz, oz), c1 = __builtin_fmaf(__uint_as_float(q1.z), iz, oz);
            float ln = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
            float lf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tbound));
            a0 = __builtin_fmaf(__uint_as_float(q2.x), ix, ox); a1 = __builtin_fmaf(__uint_as_float(q3.x), ix, ox);
            b0 = __builtin_fmaf(__uint_as_float(q2.y), iy, oy); b1 = __builtin_fmaf(__uint_as_float(q3.y), iy, oy);
            c0 = __builtin_fmaf(__uint_as_float(q2.z), iz, oz); c1 = __builtin_fmaf(__uint_as_float(q3.z), iz, oz);
            float rn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
            float rf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tbound));
            const bool hl = ln <= lf * 1.000002f;
            const bool hr = rn <= rf * 1.000002f;
            const int left = (int)q0.w, right = (int)q1.w;
            if (hl && hr) {
                const bool swap = rn < ln;
                const int far = swap ? left : right;
                cur = swap ? right : left;
                T.stack[sp * T.stride] = (unsigned)far;
                ++sp;
                pop = false;
            } else if (hl) {
                cur = left;
                pop = false;
            } else if (hr) {
                cur = right;
                pop = false;
            }
        } else {
            const int ref = ~cur;
            const int first = ref >> 3, count = (ref & 7) + 1;
            const float4* tp = reinterpret_cast<const float4*>(tris) + (size_t)first * 3;
            for (int k = 0; k < count; ++k) {
                const float4 a = tp[3 * k], b = tp[3 * k + 1], c = tp[3 * k + 2];
                if (STATS) cnt.tris += 1;
                const float t = triangle_t(a, b, c, osr);
                if (t > 0.0f && t <= tbound) consider(G, g, first + k, __float_as_int(a.w), t, osr, wr, len, tris, best, tbound);
            }
        }
        if (pop) {
            if (sp == 0) break;
            --sp;
            cur = (int)T.stack[sp * T.stride];
        }
    }
}

// Closest hit = intersectRays (kernel.cu:127-176) with each mesh's triangle loop replaced by its BVH.
template <bool STATS>
__device__ __forceinline__ void closest_hit_bvh(const GeomRecord* __restrict__ geoms, int num_geoms, const TriRecord* __restrict__ tris,
                                                const BvhNode* __restrict__ nodes, const Traversal& T, const Ray& wr, Best& best,
                                                Counters& cnt)
{
    best.dist = kInf; // kernel.cu:131
    best.geom = -1;
    best.rec = -1;
    best.px = best.py = best.pz = 0.0f;
    for (int g = 0; g < num_geoms; ++g) { // kernel.cu:133 (wave-uniform loop: records come in through scalar loads)
        const GeomRecord& G = geoms[g];
        Ray osr;
        float len;
        object_space_ray(G, wr, osr, len);
        if (G.type == FF_GEOM_TRIANGLEMESH) {
            if (G.bvh_root >= 0) traverse_mesh<STATS>(G, g, osr, wr, len, tris, nodes, T, best, cnt);
        } else {
            if (STATS) cnt.planes += 1;
            const float t = plane_t(G, osr);
            float tb = kInf;
            if (t > 0.0f) consider(G, g, -1, -1, t, osr, wr, len, tris, best, tb);
        }
    }
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
                        if (t > 0.0f) consider(G, g, G.tri_first + base + k, __float_as_int(a.w), t, osr, wr, len, tris, best, tbound);
                    }
                    if (STATS) cnt.tris += (unsigned long long)nb;
                }
            }
        } else if (live) {
            if (STATS) cnt.planes += 1;
            const float t = plane_t(G, osr);
            float tb = kInf;
            if (t > 0.0f) consider(G, g, -1, -1, t, osr, wr, len, tris, best, tb);
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

template <int MODE, bool STATS>
__global__ __launch_bounds__(kBlockThreads) void trace_kernel(const KParams p)
{
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);

    Traversal T;
    T.lds_nodes = smem;
    T.lds_count = p.lds_nodes;
    T.stride = kBlockThreads;
    T.stack = reinterpret_cast<unsigned*>(smem + (size_t)p.lds_nodes * 4) + tid;
    if (MODE == FF_TRACE_BVH) {
        // Stage the top of the BVH once per workgroup: coalesced 16-byte loads, 1 KiB per wave-instruction.
        const uint4* src = reinterpret_cast<const uint4*>(p.nodes);
        for (int i = tid; i < p.lds_nodes * 4; i += kBlockThreads) smem[i] = src[i];
        __syncthreads();
    }
    float4* batch = reinterpret_cast<float4*>(smem); // brute-force mode: triangle batch buffer

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
            closest_hit_bvh<STATS>(p.geoms, p.num_geoms, p.tris, p.nodes, T, ray, best, cnt);
        }
        if (!active) continue;

        bool path_done = true;
        if (best.geom >= 0) {
            const GeomRecord& G = p.geoms[best.geom];
            float nx, ny, nz;
            world_normal(G, p.tris, best.rec, nx, ny, nz);
            if (debug_shade) {
                // shade(), kernel.cu:178-184
                Lx = fabsf(nx); Ly = fabsf(ny); Lz = fabsf(nz);
            } else if (G.bxdf_type == FF_BXDF_EMITTER) {
                // utilities.h:96-103: two-sided emitter, m_emissiveColor * m_intensity
                Lx = Lx + bx * G.emission[0];
                Ly = Ly + by * G.emission[1];
                Lz = Lz + bz * G.emission[2];
            } else {
                // everything else is diffuse (utilities.h:109); cosine-weighted sampling, so f*cos/pdf = albedo
                bx = bx * G.albedo[0];
                by = by * G.albedo[1];
                bz = bz * G.albedo[2];
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
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x;
    Traversal T;
    T.lds_nodes = smem;
    T.lds_count = p.lds_nodes;
    T.stride = kBlockThreads;
    T.stack = reinterpret_cast<unsigned*>(smem + (size_t)p.lds_nodes * 4) + tid;
    if (MODE == FF_TRACE_BVH) {
        const uint4* src = reinterpret_cast<const uint4*>(p.nodes);
        for (int i = tid; i < p.lds_nodes * 4; i += kBlockThreads) smem[i] = src[i];
        __syncthreads();
    }
    float4* batch = reinterpret_cast<float4*>(smem);
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
    else if (live) closest_hit_bvh<false>(p.geoms, p.num_geoms, p.tris, p.nodes, T, wr, best, cnt);
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
        out.geometryIndex = best.geom;
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

size_t bvh_lds_bytes(int lds_nodes, int stack_depth)
{
    return (size_t)lds_nodes * sizeof(BvhNode) + (size_t)stack_depth * kBlockThreads * sizeof(unsigned);
}

int max_lds_nodes(int stack_depth)
{
    const long avail = (long)kLdsBudgetBytes - (long)stack_depth * kBlockThreads * (long)sizeof(unsigned);
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
    FF_SET_LDS((trace_kernel<FF_TRACE_BVH, false>))
    FF_SET_LDS((trace_kernel<FF_TRACE_BVH, true>))
    FF_SET_LDS((ray_batch_kernel<FF_TRACE_BVH>))
#undef FF_SET_LDS
    return hipSuccess;
}

hipError_t launch_trace(const KParams& p, int trace_mode, bool collect_stats, int grid_blocks, hipStream_t stream)
{
    const size_t lds = trace_mode == FF_TRACE_BVH ? bvh_lds_bytes(p.lds_nodes, p.stack_depth) : (size_t)kBruteBatchTris * sizeof(TriRecord);
    const dim3 grid(grid_blocks), block(kBlockThreads);
    if (trace_mode == FF_TRACE_BVH) {
        if (collect_stats) hipLaunchKernelGGL((trace_kernel<FF_TRACE_BVH, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((trace_kernel<FF_TRACE_BVH, false>), grid, block, lds, stream, p);
    } else {
        if (collect_stats) hipLaunchKernelGGL((trace_kernel<FF_TRACE_BRUTE_FORCE, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((trace_kernel<FF_TRACE_BRUTE_FORCE, false>), grid, block, lds, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_ray_batch(const RayBatchParams& p, int trace_mode, hipStream_t stream)
{
    if (p.n <= 0) return hipSuccess;
    const size_t lds = trace_mode == FF_TRACE_BVH ? bvh_lds_bytes(p.lds_nodes, p.stack_depth) : (size_t)kBruteBatchTris * sizeof(TriRecord);
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
