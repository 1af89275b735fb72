// ff_build.hip — LBVH construction and BVH refit on the device (gfx950).  See ff_build.h.
//
// Per mesh (T triangles, object space):
//   bounds    vertex AABB, wave-reduced then 6 ordered-integer atomics per wave
//   morton    63-bit Morton code of each triangle's centroid (21 bits per axis inside the AABB) + triangle id
//   sort      rocprim::radix_sort_pairs on the codes
//   hierarchy Karras 2012: one thread per internal node finds its range and split from common-prefix lengths
//             (ties between equal codes are broken by position, so the tree is well-formed for duplicate centroids)
//   fit       one thread per leaf walks up; the second thread to reach a node merges its children's boxes
//   rank      every internal node whose range holds more than max_leaf triangles becomes a 64-byte traversal node;
//             they are ranked by depth with a second radix sort, so node numbers grow level by level (root first)
//   emit      traversal nodes (both child boxes padded like the host builder's, links to nodes / collapsed leaves)
//   records   TriRecords in sorted (= leaf) order
//
// Refit: triangle records are rewritten from the new vertices, leaf boxes recomputed, and complete nodes propagate their
// union into the parent's child slot; a per-node arrival counter (2 arrivals = both child boxes final) decides which
// thread continues upward, so no thread ever waits for another.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ff_build.h"

namespace ff {

namespace {

constexpr int kBuildBlock = 256;
constexpr int kWaveSize = 64;
constexpr unsigned kNotEmitted = 255u; // depth key of internal nodes that are collapsed into leaves

#define FFB_HIP(call)                                                                                                    \
    do {                                                                                                                 \
        hipError_t _e = (call);                                                                                          \
        if (_e != hipSuccess) return fail(FF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

inline int grid_for(long long n) { return (int)((n + kBuildBlock - 1) / kBuildBlock); }

// float <-> int with the same ordering (for atomicMin / atomicMax on floats)
__device__ __forceinline__ int ordered_int(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

struct Box6 {
    float mn[3], mx[3];
};

__device__ __forceinline__ void grow(Box6& b, const FfVec3& v)
{
    b.mn[0] = fminf(b.mn[0], v.x); b.mn[1] = fminf(b.mn[1], v.y); b.mn[2] = fminf(b.mn[2], v.z);
    b.mx[0] = fmaxf(b.mx[0], v.x); b.mx[1] = fmaxf(b.mx[1], v.y); b.mx[2] = fmaxf(b.mx[2], v.z);
}
__device__ __forceinline__ Box6 empty_box()
{
    const float inf = __builtin_huge_valf();
    return Box6{ { inf, inf, inf }, { -inf, -inf, -inf } };
}

__global__ void init_bounds_kernel(int* bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = ordered_int(__builtin_huge_valf());
    else if (threadIdx.x < 6) bounds[threadIdx.x] = ordered_int(-__builtin_huge_valf());
}

__global__ __launch_bounds__(kBuildBlock) void bounds_kernel(const FfTriangle* __restrict__ src, int T, int* bounds)
{
    Box6 b = empty_box();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) {
        grow(b, src[i].m_v0);
        grow(b, src[i].m_v1);
        grow(b, src[i].m_v2);
    }
    for (int off = kWaveSize / 2; off > 0; off >>= 1) {
        for (int k = 0; k < 3; ++k) {
            b.mn[k] = fminf(b.mn[k], __shfl_xor(b.mn[k], off));
            b.mx[k] = fmaxf(b.mx[k], __shfl_xor(b.mx[k], off));
        }
    }
    if ((threadIdx.x & (kWaveSize - 1)) == 0) {
        for (int k = 0; k < 3; ++k) {
            atomicMin(&bounds[k], ordered_int(b.mn[k]));
            atomicMax(&bounds[3 + k], ordered_int(b.mx[k]));
        }
    }
}

// Spread the low 21 bits of v so that two zero bits follow each of them.
__device__ __forceinline__ uint64_t spread21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(kBuildBlock) void morton_kernel(const FfTriangle* __restrict__ src, int T, const int* __restrict__ bounds,
                                                              uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const FfTriangle& t = src[i];
    const float c[3] = { (t.m_v0.x + t.m_v1.x + t.m_v2.x) * (1.0f / 3.0f), (t.m_v0.y + t.m_v1.y + t.m_v2.y) * (1.0f / 3.0f),
                         (t.m_v0.z + t.m_v1.z + t.m_v2.z) * (1.0f / 3.0f) };
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = ordered_float(bounds[k]), hi = ordered_float(bounds[3 + k]);
        const float ext = hi - lo;
        const float n = ext > 0.0f ? (c[k] - lo) / ext : 0.5f;
        q[k] = (uint32_t)fminf(fmaxf(n * 2097152.0f, 0.0f), 2097151.0f);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = (uint32_t)i;
}

// Length of the common prefix of the (code, position) keys at sorted positions i and j; -1 outside the array.
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int T, int i, int j)
{
    if (j < 0 || j >= T) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz(i ^ j);
    return __clzll((long long)(a ^ b));
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", section 4.
// Children: >= 0 internal node index, < 0 leaf at sorted position ~child.
__global__ __launch_bounds__(kBuildBlock) void hierarchy_kernel(const uint64_t* __restrict__ keys, int T, int* __restrict__ left, int* __restrict__ right,
                                                                 int* __restrict__ first, int* __restrict__ last, int* __restrict__ node_parent,
                                                                 int* __restrict__ leaf_parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T - 1) return;
    const int d = common_prefix(keys, T, i, i + 1) - common_prefix(keys, T, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, T, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, T, i, i + lmax * d) > dmin && lmax < (1 << 30)) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (common_prefix(keys, T, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(keys, T, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (common_prefix(keys, T, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lc = lo == gamma ? ~gamma : gamma;
    const int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    first[i] = lo;
    last[i] = hi;
    if (lc >= 0) node_parent[lc] = i; else leaf_parent[~lc] = i;
    if (rc >= 0) node_parent[rc] = i; else leaf_parent[~rc] = i;
    if (i == 0) node_parent[0] = -1;
}

// Read a value another workgroup wrote earlier in this launch (after the arrival-counter handshake): agent-scope load.
__device__ __forceinline__ float coherent_load(const float* p)
{
    return __int_as_float(__hip_atomic_load(reinterpret_cast<int*>(const_cast<float*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void coherent_store_int(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(float* p, float v)
{
    __hip_atomic_store(reinterpret_cast<int*>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What the bottom-up kernels hand from thread to thread (boxes, costs, links) is written with agent-scope stores and read with
// agent-scope loads; the hand-over itself is an arrival counter.  Then the only ordering needed is "my stores have completed
// before my arrival counts" / "my loads start after it": release_arrival() / acquire_arrival().  A __threadfence() would also
// write back and invalidate the XCD's whole L2 (the L2s of the eight XCDs are not coherent with each other for ordinary
// accesses), thousands of times per launch: the 983 040-triangle treelet pass took 12 ms with it and 3 ms without.
// The workgroup-scope fence only stops the COMPILER from moving accesses across it (on gfx950 it emits no instruction); what makes
// "my stores have completed" true in the binary is the explicit wait for the wave's outstanding vector-memory operations in front
// of it: the sc1 stores above and the arrival atomic hit different lines and L2 channels and may otherwise complete out of order
// (the Makefile's `check-arrivals` target greps the ISA for an sc1 store that reaches an arrival atomic without that wait).
__device__ __forceinline__ void release_arrival()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
__device__ __forceinline__ void acquire_arrival() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ int arrive(int* counter, int n = 1) { return __hip_atomic_fetch_add(counter, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int box_slot(int child, int T) { return child >= 0 ? child : (T - 1) + ~child; }

// boxes: (2T - 1) x 6 floats, internal nodes first, then leaves by sorted position.
__global__ __launch_bounds__(kBuildBlock) void fit_kernel(const FfTriangle* __restrict__ src, const uint32_t* __restrict__ vals, int T,
                                                           const int* __restrict__ left, const int* __restrict__ right,
                                                           const int* __restrict__ node_parent, const int* __restrict__ leaf_parent, float* boxes,
                                                           int* arrivals)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    const FfTriangle& t = src[vals[j]];
    Box6 b = empty_box();
    grow(b, t.m_v0);
    grow(b, t.m_v1);
    grow(b, t.m_v2);
    float* mine = boxes + (size_t)((T - 1) + j) * 6;
    for (int k = 0; k < 3; ++k) { coherent_store(mine + k, b.mn[k]); coherent_store(mine + 3 + k, b.mx[k]); }
    int cur = leaf_parent[j];
    for (int guard = 0; guard < 4096; ++guard) { // a path to the root is at most 64 + 31 links long
        release_arrival(); // the box written above is out before it is announced
        const int earlier = arrive(&arrivals[cur]);
        if (earlier == 0) return; // the sibling subtree is not finished: its last thread completes this node
        acquire_arrival();
        const float* lb = boxes + (size_t)box_slot(left[cur], T) * 6;
        const float* rb = boxes + (size_t)box_slot(right[cur], T) * 6;
        float* nb = boxes + (size_t)cur * 6;
        for (int k = 0; k < 3; ++k) {
            coherent_store(nb + k, fminf(coherent_load(lb + k), coherent_load(rb + k)));
            coherent_store(nb + 3 + k, fmaxf(coherent_load(lb + 3 + k), coherent_load(rb + 3 + k)));
        }
        if (cur == 0) return;
        cur = node_parent[cur];
    }
}

// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) --------------------------------------------------
//
// Bottom-up agglomeration on the Morton-sorted leaves: every cluster finds, within a window of +-kPlocRadius positions,
// the neighbour whose union with it has the smallest surface area; mutual choices merge into a new node; the survivors
// are compacted in order and the loop repeats until one cluster is left.  Trees are close to SAH quality, unlike the
// LBVH's midpoint splits.  Internal node ids are handed out from T-2 downwards so that the last merge - the root - is
// node 0, which is what the ranking / emission stages expect.

constexpr int kPlocRadius = 16; // default; FF_PLOC_RADIUS (1..256) overrides it for experiments

int ploc_radius()
{
    if (const char* e = std::getenv("FF_PLOC_RADIUS")) {
        const int r = std::atoi(e);
        if (r >= 1 && r <= 256) return r;
    }
    return kPlocRadius;
}

__global__ __launch_bounds__(kBuildBlock) void leaf_boxes_kernel(const FfTriangle* __restrict__ src, const uint32_t* __restrict__ vals, int T,
                                                                  float* __restrict__ boxes, int* __restrict__ clusters)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    const FfTriangle& t = src[vals[j]];
    Box6 b = empty_box();
    grow(b, t.m_v0);
    grow(b, t.m_v1);
    grow(b, t.m_v2);
    float* mine = boxes + (size_t)((T - 1) + j) * 6;
    for (int k = 0; k < 3; ++k) { mine[k] = b.mn[k]; mine[3 + k] = b.mx[k]; }
    clusters[j] = ~j;
}

__device__ __forceinline__ float union_area(const float* a, const float* b)
{
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
    const float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
    const float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(kBuildBlock) void ploc_neighbour_kernel(int n, int T, const int* __restrict__ clusters, const float* __restrict__ boxes,
                                                                      int* __restrict__ nearest, int radius)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mine[6];
    const float* mb = boxes + (size_t)box_slot(clusters[i], T) * 6;
    for (int k = 0; k < 6; ++k) mine[k] = mb[k];
    float best = __builtin_huge_valf();
    int best_j = -1;
    // Candidates are visited from the closest position outwards and only a strictly smaller area replaces the choice, so
    // among equal areas (regular tessellations are full of them) the closest position wins; at equal distance an even
    // position looks right first and an odd one left first, which makes (2k, 2k+1) choose each other.
    for (int d = 1; d <= radius; ++d) {
        const int first_j = (i & 1) ? i - d : i + d, second_j = (i & 1) ? i + d : i - d;
        const int cand[2] = { first_j, second_j };
        for (int c = 0; c < 2; ++c) {
            const int j = cand[c];
            if (j < 0 || j >= n) continue;
            const float a = union_area(mine, boxes + (size_t)box_slot(clusters[j], T) * 6);
            if (a < best) {
                best = a;
                best_j = j;
            }
        }
    }
    nearest[i] = best_j;
}

// Mutual nearest neighbours merge (the lower position creates the node); everything else survives unchanged.
__global__ __launch_bounds__(kBuildBlock) void ploc_merge_kernel(int n, int T, const int* __restrict__ clusters, const int* __restrict__ nearest,
                                                                  float* __restrict__ boxes, int* __restrict__ left, int* __restrict__ right,
                                                                  int* __restrict__ node_parent, int* __restrict__ leaf_parent, int* __restrict__ sizes,
                                                                  int* counters /* [2] nodes created */, int* __restrict__ merged, uint32_t* __restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = nearest[i];
    const bool mutual = j >= 0 && nearest[j] == i;
    if (!mutual) {
        merged[i] = clusters[i];
        keep[i] = 1u;
        return;
    }
    if (i > j) {
        keep[i] = 0u;
        return;
    }
    const int a = clusters[i], b = clusters[j];
    const int id = (T - 2) - atomicAdd(&counters[2], 1);
    if (id == 0) node_parent[0] = -1; // the last merge is the root
    left[id] = a;
    right[id] = b;
    if (a >= 0) node_parent[a] = id; else leaf_parent[~a] = id;
    if (b >= 0) node_parent[b] = id; else leaf_parent[~b] = id;
    sizes[id] = (a >= 0 ? sizes[a] : 1) + (b >= 0 ? sizes[b] : 1);
    const float* ba = boxes + (size_t)box_slot(a, T) * 6;
    const float* bb = boxes + (size_t)box_slot(b, T) * 6;
    float* nb = boxes + (size_t)id * 6;
    for (int k = 0; k < 3; ++k) {
        nb[k] = fminf(ba[k], bb[k]);
        nb[3 + k] = fmaxf(ba[3 + k], bb[3 + k]);
    }
    merged[i] = id;
    keep[i] = 1u;
}

__global__ __launch_bounds__(kBuildBlock) void ploc_compact_kernel(int n, const int* __restrict__ merged, const uint32_t* __restrict__ keep,
                                                                    const uint32_t* __restrict__ position, int* __restrict__ clusters_out, int* counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (keep[i]) clusters_out[position[i]] = merged[i];
    if (i == n - 1) counters[3] = (int)(position[i] + keep[i]); // clusters left
}

// ---- treelet restructuring (Karras & Aila 2013, "Fast parallel construction of high-quality bounding volume hierarchies") ----
//
// A post-pass on the binary topology either builder produced (left / right / node_parent / leaf_parent / sizes / boxes), before
// the nodes are ranked and emitted.  Bottom-up, like the box fit: the second thread to arrive at an internal node owns it.  A node
// with at least kTreeletLeaves triangles below becomes the root of a treelet: its two children, then repeatedly the treelet leaf
// of the largest surface area replaced by ITS children, until there are kTreeletLeaves treelet leaves (subtrees kept whole) under
// kTreeletLeaves - 1 internal nodes.  The topology of those internal nodes is then rebuilt as the one of least SAH cost over all
// binary trees on the seven leaves - dynamic programming over the 127 subsets, each subset's best split found by enumerating its
// partitions - and written back into the same node slots.  The LBVH's midpoint splits lose 20-30 % of the trace rate of a SAH
// tree; two passes of this get most of it back for a few hundred microseconds on a 5 000-triangle mesh.
constexpr int kTreeletLeaves = 7;
constexpr float kSahNode = 1.2f, kSahTri = 1.0f; // the host builder's costs (BvhBuildParams::c_trav against one triangle test)

__device__ __forceinline__ int coherent_load_int(const int* p)
{
    return __hip_atomic_load(const_cast<int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float box_half_area6(const float* b)
{
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(kBuildBlock) void sizes_from_ranges_kernel(int T, const int* __restrict__ first, const int* __restrict__ last, int* __restrict__ sizes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T - 1) sizes[i] = last[i] - first[i] + 1;
}

// The dynamic programme's work list, built once on the host (treelet_tables): for every subset size k = 2..7 the (subset,
// partition) pairs - a partition is a non-empty proper part of the subset that leaves out its lowest member, so every split comes
// up once - and the subsets of that size.  966 pairs, 120 subsets.
struct TreeletTables {
    unsigned short pair[1024];   // subset | partition << 7, grouped by subset size
    unsigned short pair_first[9]; // pair_first[k] .. pair_first[k + 1]: the pairs of the subsets of size k (k = 2..7)
    unsigned char subset[128];   // the subsets grouped by size
    unsigned short subset_first[9];
};

// cost: (2T - 1) floats, SAH cost of every subtree (internal nodes first, then leaves by sorted position), written here.
// One thread per triangle climbs like the box fit; the treelets its wave's lanes come to own are optimised one after the other by
// the WHOLE wave: subset areas two per lane, the partitions of each subset size spread over the lanes with the best one found by
// a 64-bit LDS minimum on (cost bits, partition), the rebuilt nodes written by one lane.  (A thread per treelet runs the same
// programme out of scratch memory: 4 ms per pass on a 5 000-triangle mesh, most of it on the few treelets at the top, one after
// the other.)
__global__ __launch_bounds__(kBuildBlock) void treelet_kernel(int T, int* left, int* right, int* node_parent, int* leaf_parent, int* sizes, float* boxes,
                                                               float* cost, int* arrivals, int* counters /* [4] treelets rebuilt */,
                                                               const TreeletTables* __restrict__ tables)
{
    constexpr int kSets = 1 << kTreeletLeaves;
    constexpr int kWavesPerBlock = kBuildBlock / kWaveSize;
    __shared__ float s_area[kWavesPerBlock][kSets];
    __shared__ float s_copt[kWavesPerBlock][kSets];
    __shared__ unsigned long long s_best[kWavesPerBlock][kSets];
    __shared__ unsigned char s_popt[kWavesPerBlock][kSets];
    __shared__ int s_plan[kWavesPerBlock][(kTreeletLeaves - 1) * 4]; // the rebuilt nodes: subset, node slot, left child, right child
    const int lane = threadIdx.x & (kWaveSize - 1), wave = threadIdx.x / kWaveSize;
    float* area = s_area[wave];
    float* copt = s_copt[wave];
    unsigned long long* best = s_best[wave];
    unsigned char* popt = s_popt[wave];
    // (LDS operations of ONE wave are carried out in the order they were issued: what the lanes of this wave exchange through LDS
    // needs the compiler to keep that order - a wavefront-scope fence and a scheduling barrier - and no wait on the memory system)
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = j < T;
    int cur = -1;
    if (alive) {
        coherent_store(&cost[(T - 1) + j], kSahTri * box_half_area6(boxes + (size_t)((T - 1) + j) * 6));
        cur = leaf_parent[j];
    }
    for (int guard = 0; guard < 4096; ++guard) {
        int owned = -1;
        if (alive) {
            // Everything this thread (or its wave) wrote for the subtree below went out with agent-scope stores: waiting for them to
            // complete is all the release the arrival needs, and the reads on the other side are agent-scope loads.  (A full
            // __threadfence writes back and invalidates the XCD's whole L2 on this chip: the climb does thousands of them.)
            release_arrival();
            if (arrive(&arrivals[cur]) == 0) alive = false; // the sibling subtree is not finished
            else owned = cur;
        }
        acquire_arrival();
        // small subtrees: their owner computes the cost alone; treelet roots wait for the wave
        bool wants_wave = false;
        if (owned >= 0) {
            if (coherent_load_int(&sizes[owned]) >= kTreeletLeaves) {
                wants_wave = true;
            } else {
                const int l = coherent_load_int(&left[owned]), r = coherent_load_int(&right[owned]);
                float bb[6];
                for (int q = 0; q < 6; ++q) bb[q] = coherent_load(boxes + (size_t)owned * 6 + q);
                coherent_store(&cost[owned], kSahNode * box_half_area6(bb) + coherent_load(&cost[box_slot(l, T)]) + coherent_load(&cost[box_slot(r, T)]));
            }
        }
        unsigned long long waiting = __ballot(wants_wave);
        while (waiting != 0ull) {
            const int leader = __ffsll((long long)waiting) - 1;
            waiting &= waiting - 1ull;
            const int root = __builtin_amdgcn_readlane(owned, leader);
            // ---- the whole wave on the treelet under `root` ----
            // Forming the treelet is a chain of dependent reads of what other waves wrote (children of the node picked, then their
            // boxes): each step is ONE round trip, its values fetched by as many lanes as there are and broadcast from there.
            int leaf[kTreeletLeaves], inner[kTreeletLeaves - 1];
            float lbox[kTreeletLeaves][6], larea[kTreeletLeaves];
            auto bcast = [](float v, int l) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); };
            auto load_children = [&](int n, int& l, int& r) {
                int v = 0;
                if (lane < 2) v = coherent_load_int(lane == 0 ? &left[n] : &right[n]);
                l = __builtin_amdgcn_readlane(v, 0);
                r = __builtin_amdgcn_readlane(v, 1);
            };
            auto load_two_boxes = [&](int ka, int kb) { // the boxes of treelet leaves ka and kb: lanes 0..5 and 6..11
                float v = 0.0f;
                if (lane < 12) v = coherent_load(boxes + (size_t)box_slot(leaf[lane < 6 ? ka : kb], T) * 6 + (lane < 6 ? lane : lane - 6));
                for (int q = 0; q < 6; ++q) {
                    lbox[ka][q] = bcast(v, q);
                    lbox[kb][q] = bcast(v, 6 + q);
                }
                larea[ka] = box_half_area6(lbox[ka]);
                larea[kb] = box_half_area6(lbox[kb]);
            };
            int nl = 2, ni = 1;
            inner[0] = root;
            load_children(root, leaf[0], leaf[1]);
            load_two_boxes(0, 1);
            while (nl < kTreeletLeaves) {
                int pick = -1;
                float big = -1.0f;
                for (int k = 0; k < nl; ++k)
                    if (leaf[k] >= 0 && larea[k] > big) { big = larea[k]; pick = k; }
                if (pick < 0) break; // (cannot happen below a node of at least kTreeletLeaves triangles)
                const int n = leaf[pick];
                inner[ni++] = n;
                load_children(n, leaf[pick], leaf[nl]);
                load_two_boxes(pick, nl);
                ++nl;
            }
            float lcost[kTreeletLeaves];
            int lsize[kTreeletLeaves];
            float now = 0.0f;
            {
                // costs and sizes of the treelet leaves (lanes 0..6), areas of its internal nodes as they stand (lanes 8..13)
                float cv = 0.0f, av = 0.0f;
                int sv = 1;
                if (lane < nl) {
                    const int ref = leaf[lane];
                    cv = coherent_load(&cost[box_slot(ref, T)]);
                    if (ref >= 0) sv = coherent_load_int(&sizes[ref]);
                } else if (lane >= 8 && lane < 8 + ni) {
                    float bb[6];
                    for (int r = 0; r < 6; ++r) bb[r] = coherent_load(boxes + (size_t)inner[lane - 8] * 6 + r);
                    av = box_half_area6(bb);
                }
                for (int k = 0; k < kTreeletLeaves; ++k) {
                    lcost[k] = bcast(cv, k);
                    lsize[k] = __builtin_amdgcn_readlane(sv, k);
                    if (k < nl) now += lcost[k];
                }
                for (int q = 0; q < kTreeletLeaves - 1; ++q)
                    if (q < ni) now += kSahNode * bcast(av, 8 + q);
            }
            float total = now;
            if (nl == kTreeletLeaves) {
                auto subset_box = [&](int s, float* bb, int& sz) {
                    for (int q = 0; q < 3; ++q) { bb[q] = __builtin_huge_valf(); bb[3 + q] = -__builtin_huge_valf(); }
                    sz = 0;
                    for (int k = 0; k < kTreeletLeaves; ++k)
                        if (s & (1 << k)) {
                            for (int q = 0; q < 3; ++q) { bb[q] = fminf(bb[q], lbox[k][q]); bb[3 + q] = fmaxf(bb[3 + q], lbox[k][3 + q]); }
                            sz += lsize[k];
                        }
                };
                for (int s = lane; s < kSets; s += kWaveSize) {
                    if (s != 0) {
                        float bb[6];
                        int sz;
                        subset_box(s, bb, sz);
                        area[s] = box_half_area6(bb);
                    }
                    best[s] = ~0ull;
                    if (s != 0 && (s & (s - 1)) == 0) copt[s] = lcost[__ffs(s) - 1];
                }
                wave_sync();
                for (int k = 2; k <= kTreeletLeaves; ++k) {
                    for (int q = tables->pair_first[k] + lane; q < tables->pair_first[k + 1]; q += kWaveSize) {
                        const int e = tables->pair[q], s = e & (kSets - 1), p = e >> kTreeletLeaves;
                        const float c = copt[p] + copt[s ^ p];
                        atomicMin(&best[s], ((unsigned long long)__float_as_uint(c) << 32) | (unsigned)p); // (costs are positive: their bits order like they do)
                    }
                    wave_sync();
                    for (int q = tables->subset_first[k] + lane; q < tables->subset_first[k + 1]; q += kWaveSize) {
                        const int s = tables->subset[q];
                        const unsigned long long b = best[s];
                        copt[s] = kSahNode * area[s] + __uint_as_float((unsigned)(b >> 32));
                        popt[s] = (unsigned char)(b & (kSets - 1));
                    }
                    wave_sync();
                }
                const float optimum = copt[kSets - 1];
                if (optimum < now * 0.9999f) {
                    total = optimum;
                    // The optimal topology, walked from the full set (a few register steps, every lane the same), gives each of the six
                    // node slots its subset and its two children; then lane q writes slot q: one round of stores.
                    int* plan = s_plan[wave];
                    if (lane == 0) {
                        int next_slot = 1, done = 0;
                        int todo_set[kTreeletLeaves], todo_node[kTreeletLeaves], top = 0;
                        todo_set[top] = kSets - 1;
                        todo_node[top] = root;
                        ++top;
                        while (top > 0) {
                            --top;
                            const int s = todo_set[top], n = todo_node[top];
                            const int part[2] = { popt[s], s ^ popt[s] };
                            int child[2];
                            for (int side = 0; side < 2; ++side) {
                                const int ps = part[side];
                                if ((ps & (ps - 1)) == 0) {
                                    child[side] = leaf[__ffs(ps) - 1];
                                } else {
                                    const int c = inner[next_slot++];
                                    child[side] = c;
                                    todo_set[top] = ps;
                                    todo_node[top] = c;
                                    ++top;
                                }
                            }
                            plan[done * 4 + 0] = s;
                            plan[done * 4 + 1] = n;
                            plan[done * 4 + 2] = child[0];
                            plan[done * 4 + 3] = child[1];
                            ++done;
                        }
                        atomicAdd(&counters[4], 1);
                    }
                    wave_sync();
                    if (lane < kTreeletLeaves - 1) {
                        const int s = plan[lane * 4], n = plan[lane * 4 + 1], c0 = plan[lane * 4 + 2], c1 = plan[lane * 4 + 3];
                        coherent_store_int(&left[n], c0);
                        coherent_store_int(&right[n], c1);
                        coherent_store_int(c0 >= 0 ? &node_parent[c0] : &leaf_parent[~c0], n);
                        coherent_store_int(c1 >= 0 ? &node_parent[c1] : &leaf_parent[~c1], n);
                        float bb[6];
                        int sz;
                        subset_box(s, bb, sz);
                        float* nb = boxes + (size_t)n * 6;
                        for (int q = 0; q < 6; ++q) coherent_store(nb + q, bb[q]);
                        coherent_store_int(&sizes[n], sz);
                        coherent_store(&cost[n], copt[s]);
                    }
                }
                wave_sync();
            }
            if (lane == 0) coherent_store(&cost[root], total);
        }
        // owners climb
        if (owned >= 0) {
            if (owned == 0) alive = false;
            else cur = coherent_load_int(&node_parent[owned]);
        }
        if (__ballot(alive) == 0ull) break;
    }
}

// ---- parallel reinsertion (after Meister & Bittner 2018, "Parallel reinsertion for bounding volume hierarchy optimization") ------
//
// The device twin of the host builder's insertion-based optimisation (csrc/ff_scene.cpp Builder::optimise), in passes of five
// launches over the build arrays (left / right / parents / boxes):
//   find   every node x - internal node or leaf - below the root's children looks for the place where it would add the least area:
//          a depth-first search from the root that carries the area the boxes on the way down grow by when x joins them and prunes
//          on the best place so far.  Taking x out makes its parent p disappear (its sibling moves up), which saves area(p); the
//          shrinking of p's ancestors is NOT counted, so a positive gain is a real one.
//   lock   a move rewires six nodes - x, p, the sibling, p's parent, the target y and y's parent: every candidate writes
//          (gain, x) into their lock words with an atomic maximum;
//   check  a candidate that holds all six is a winner - unless its target lies inside another winner's moving subtree (both moves
//          together could close a cycle): it walks up from y and gives up if it meets one;
//   apply  the winners rewire (p becomes the parent of y and x where y was);
//   refit  boxes and triangle counts of all internal nodes, bottom-up with arrival counters.
// Slots: [0, T-1) internal nodes, [T-1, 2T-1) leaves by sorted position (the numbering of `boxes`).
__device__ __forceinline__ int slot_ref(int slot, int T) { return slot < T - 1 ? slot : ~(slot - (T - 1)); }
__device__ __forceinline__ int slot_parent(int slot, int T, const int* node_parent, const int* leaf_parent)
{
    return slot < T - 1 ? node_parent[slot] : leaf_parent[slot - (T - 1)];
}
__device__ __forceinline__ float union_half_area6(const float* a, const float* b)
{
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(kBuildBlock) void reinsert_find_kernel(int T, const int* __restrict__ left, const int* __restrict__ right,
                                                                     const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                                                                     const float* __restrict__ boxes, int* __restrict__ target, float* __restrict__ gain,
                                                                     unsigned long long* __restrict__ lock)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * T - 1) return;
    target[x] = -1;
    gain[x] = 0.f;
    lock[x] = 0ull;
    if (x == 0) return;
    const int p = slot_parent(x, T, node_parent, leaf_parent);
    if (p <= 0) return; // a child of the root: no grandparent to hand the sibling to
    float xb[6];
    for (int k = 0; k < 6; ++k) xb[k] = boxes[(size_t)x * 6 + k];
    const float xa = box_half_area6(xb);
    const float saved = box_half_area6(boxes + (size_t)p * 6);
    float best_cost = saved * 0.9999f; // (a move has to pay by more than rounding)
    int best = -1;
    constexpr int kStack = 64;
    int st_node[kStack];
    float st_ind[kStack];
    int sp = 0;
    st_node[sp] = left[0]; st_ind[sp] = 0.f; ++sp;   // (never the root's own place: node 0 stays the root)
    st_node[sp] = right[0]; st_ind[sp] = 0.f; ++sp;
    for (int visited = 0; sp > 0 && visited < 1024; ++visited) {
        --sp;
        const int ref = st_node[sp];
        const float ind = st_ind[sp];
        const int n = box_slot(ref, T);
        if (n == x) continue; // (x and what hangs below it move: not a place)
        if (ind + xa >= best_cost) continue;
        const float* nb = boxes + (size_t)n * 6;
        const float direct = union_half_area6(nb, xb);
        if (n != p && ind + direct < best_cost) { // (p disappears with the move; its other child is reached through it)
            best_cost = ind + direct;
            best = n;
        }
        if (ref >= 0) {
            const float below = n == p ? ind : ind + (direct - box_half_area6(nb));
            if (below + xa < best_cost && sp + 2 <= kStack) {
                st_node[sp] = left[ref]; st_ind[sp] = below; ++sp;
                st_node[sp] = right[ref]; st_ind[sp] = below; ++sp;
            }
        }
    }
    if (best < 0) return;
    // (the sibling's place is where x already is: never a gain by construction, but rounding must not make it one)
    const int xref = slot_ref(x, T);
    const int sib = box_slot(left[p] == xref ? right[p] : left[p], T);
    if (best == sib) return;
    target[x] = best;
    gain[x] = saved - best_cost;
}

__device__ __forceinline__ unsigned long long reinsert_key(float g, int x) { return ((unsigned long long)__float_as_uint(g) << 32) | (unsigned)x; }

__global__ __launch_bounds__(kBuildBlock) void reinsert_lock_kernel(int T, const int* __restrict__ left, const int* __restrict__ right,
                                                                     const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                                                                     const int* __restrict__ target, const float* __restrict__ gain, unsigned long long* lock)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * T - 1 || target[x] < 0) return;
    const int p = slot_parent(x, T, node_parent, leaf_parent);
    const int g = node_parent[p];
    const int xref = slot_ref(x, T);
    const int sib = box_slot(left[p] == xref ? right[p] : left[p], T);
    const int y = target[x];
    const int q = slot_parent(y, T, node_parent, leaf_parent);
    const unsigned long long key = reinsert_key(gain[x], x);
    atomicMax(&lock[x], key);
    atomicMax(&lock[p], key);
    atomicMax(&lock[sib], key);
    atomicMax(&lock[g], key);
    atomicMax(&lock[y], key);
    atomicMax(&lock[q], key);
}

// ok[x]: 1 = x holds its six locks (a winner so far)
__global__ __launch_bounds__(kBuildBlock) void reinsert_hold_kernel(int T, const int* __restrict__ left, const int* __restrict__ right,
                                                                     const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                                                                     const int* __restrict__ target, const float* __restrict__ gain,
                                                                     const unsigned long long* __restrict__ lock, int* __restrict__ ok)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * T - 1) return;
    int holds = 0;
    if (target[x] >= 0) {
        const int p = slot_parent(x, T, node_parent, leaf_parent);
        const int g = node_parent[p];
        const int xref = slot_ref(x, T);
        const int sib = box_slot(left[p] == xref ? right[p] : left[p], T);
        const int y = target[x];
        const int q = slot_parent(y, T, node_parent, leaf_parent);
        const unsigned long long key = reinsert_key(gain[x], x);
        holds = lock[x] == key && lock[p] == key && lock[sib] == key && lock[g] == key && lock[y] == key && lock[q] == key && q >= 0;
    }
    ok[x] = holds;
}

// ok[x] 1 -> 2 for the winners whose target does not lie inside another winner's moving subtree
__global__ __launch_bounds__(kBuildBlock) void reinsert_check_kernel(int T, const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                                                                      const int* __restrict__ target, const int* __restrict__ ok, int* __restrict__ go)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * T - 1) return;
    int fine = ok[x];
    if (fine) {
        int a = target[x];
        if (a != x && ok[a]) fine = 0; // (the target itself moves)
        a = slot_parent(a, T, node_parent, leaf_parent);
        for (int guard = 0; fine && a > 0 && guard < 4096; ++guard) {
            if (ok[a]) fine = 0;
            a = node_parent[a];
        }
    }
    go[x] = fine;
}

__global__ __launch_bounds__(kBuildBlock) void reinsert_apply_kernel(int T, int* left, int* right, int* node_parent, int* leaf_parent,
                                                                      const int* __restrict__ target, const int* __restrict__ go, int* counters)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * T - 1 || !go[x]) return;
    auto set_parent = [&](int slot, int parent) {
        if (slot < T - 1) node_parent[slot] = parent;
        else leaf_parent[slot - (T - 1)] = parent;
    };
    const int p = slot_parent(x, T, node_parent, leaf_parent);
    const int g = node_parent[p];
    const int xref = slot_ref(x, T), pref = p;
    const int sibref = left[p] == xref ? right[p] : left[p];
    const int y = target[x];
    const int yref = slot_ref(y, T);
    const int q = slot_parent(y, T, node_parent, leaf_parent);
    // the sibling takes p's place under p's parent
    if (left[g] == pref) left[g] = sibref;
    else right[g] = sibref;
    set_parent(box_slot(sibref, T), g);
    // p becomes the parent of y and x where y was (read q's links after the step above: q may be g)
    if (left[q] == yref) left[q] = pref;
    else right[q] = pref;
    node_parent[p] = q;
    left[p] = yref;
    right[p] = xref;
    set_parent(y, p);
    set_parent(x, p);
    atomicAdd(&counters[5], 1); // (how many moves the build made in all: a statistic)
}

// boxes and triangle counts of all internal nodes from the leaves' boxes (arrivals zeroed by the host)
__global__ __launch_bounds__(kBuildBlock) void reinsert_refit_kernel(int T, const int* __restrict__ left, const int* __restrict__ right,
                                                                      const int* __restrict__ node_parent, const int* __restrict__ leaf_parent, float* boxes,
                                                                      int* sizes, int* arrivals)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    int cur = leaf_parent[j];
    for (int guard = 0; guard < 4096; ++guard) {
        release_arrival();
        const int earlier = arrive(&arrivals[cur]);
        if (earlier == 0) return;
        acquire_arrival();
        const int l = left[cur], r = right[cur];
        const float* lb = boxes + (size_t)box_slot(l, T) * 6;
        const float* rb = boxes + (size_t)box_slot(r, T) * 6;
        float* nb = boxes + (size_t)cur * 6;
        for (int k = 0; k < 3; ++k) {
            coherent_store(nb + k, fminf(coherent_load(lb + k), coherent_load(rb + k)));
            coherent_store(nb + 3 + k, fmaxf(coherent_load(lb + 3 + k), coherent_load(rb + 3 + k)));
        }
        const int nl = l >= 0 ? __hip_atomic_load(&sizes[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
        const int nr = r >= 0 ? __hip_atomic_load(&sizes[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
        coherent_store_int(&sizes[cur], nl + nr);
        if (cur == 0) return;
        cur = node_parent[cur];
    }
}

// Depth-first position of every node's first leaf: the sum, over the ancestors it reaches as a RIGHT child, of the left
// sibling's size.  Gives every internal node its contiguous triangle range and every leaf its place in the leaf order.
__global__ __launch_bounds__(kBuildBlock) void ploc_ranges_kernel(int T, const int* __restrict__ left, const int* __restrict__ right,
                                                                   const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                                                                   const int* __restrict__ sizes, int* __restrict__ first, int* __restrict__ last,
                                                                   int* __restrict__ leaf_position)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x; // [0, T-1): internal nodes, [T-1, 2T-1): leaves
    if (x >= 2 * T - 1) return;
    const bool is_leaf = x >= T - 1;
    int ref = is_leaf ? ~(x - (T - 1)) : x;
    int parent = is_leaf ? leaf_parent[x - (T - 1)] : node_parent[x];
    if (!is_leaf && x == 0) parent = -1;
    int offset = 0;
    for (int guard = 0; parent >= 0 && guard < 4096; ++guard) {
        if (right[parent] == ref) {
            const int l = left[parent];
            offset += l >= 0 ? sizes[l] : 1;
        }
        ref = parent;
        parent = parent == 0 ? -1 : node_parent[parent];
    }
    if (is_leaf) {
        leaf_position[x - (T - 1)] = offset;
    } else {
        first[x] = offset;
        last[x] = offset + sizes[x] - 1;
    }
}

// Depth key of every internal node: its depth (root = 1) if it becomes a traversal node, kNotEmitted otherwise.
__global__ __launch_bounds__(kBuildBlock) void rank_key_kernel(int T, int max_leaf, const int* __restrict__ first, const int* __restrict__ last,
                                                                const int* __restrict__ node_parent, uint32_t* __restrict__ depth_key,
                                                                uint32_t* __restrict__ ids, int* counters /* [0] emitted, [1] max depth */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T - 1) return;
    ids[i] = (uint32_t)i;
    if (last[i] - first[i] + 1 <= max_leaf) {
        depth_key[i] = kNotEmitted;
        return;
    }
    int depth = 1;
    for (int p = node_parent[i]; p >= 0 && depth < 200; p = node_parent[p]) ++depth;
    depth_key[i] = (uint32_t)depth;
    atomicAdd(&counters[0], 1);
    atomicMax(&counters[1], depth);
}

__global__ __launch_bounds__(kBuildBlock) void rank_scatter_kernel(int n, const uint32_t* __restrict__ sorted_ids, int* __restrict__ new_index)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) new_index[sorted_ids[r]] = r;
}

__device__ __forceinline__ float mesh_pad(const int* bounds)
{
    // the host builder's padding (ff_scene.cpp): 1e-4 of the mesh's largest |coordinate|
    float big = 0.0f;
    for (int k = 0; k < 6; ++k) big = fmaxf(big, fabsf(ordered_float(bounds[k])));
    return 1.0e-4f * fmaxf(big, 1.0e-3f);
}

__global__ __launch_bounds__(kBuildBlock) void emit_kernel(int T, int emitted, int max_leaf, int tri_first, int node_base, const int* __restrict__ bounds,
                                                            const uint32_t* __restrict__ sorted_ids, const int* __restrict__ new_index,
                                                            const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ first,
                                                            const int* __restrict__ last, const float* __restrict__ boxes, const int* __restrict__ leaf_position,
                                                            BvhNode* __restrict__ nodes)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= emitted) return;
    const int i = (int)sorted_ids[r];
    const float pad = mesh_pad(bounds);
    BvhNode nd;
    nd.pad0 = 0;
    nd.pad1 = 0;
    const int child[2] = { left[i], right[i] };
    int link[2];
    for (int side = 0; side < 2; ++side) {
        const int c = child[side];
        const float* b = boxes + (size_t)box_slot(c, T) * 6;
        float* mn = side == 0 ? nd.lmin : nd.rmin;
        float* mx = side == 0 ? nd.lmax : nd.rmax;
        for (int k = 0; k < 3; ++k) {
            mn[k] = b[k] - pad;
            mx[k] = b[3 + k] + pad;
        }
        if (c < 0) {
            link[side] = ~(((tri_first + (leaf_position ? leaf_position[~c] : ~c)) << 3) | 0);
        } else {
            const int len = last[c] - first[c] + 1;
            link[side] = len <= max_leaf ? ~(((tri_first + first[c]) << 3) | (len - 1)) : node_base + new_index[c];
        }
    }
    nd.left = link[0];
    nd.right = link[1];
    nodes[node_base + r] = nd;
}

// The record of ff_scene.cpp's build_mesh_bvh, computed the same way (edges by one fp32 subtraction, margin in double).
__device__ __forceinline__ void write_record(TriRecord& r, const FfTriangle& t, int orig_index)
{
    r.v0[0] = t.m_v0.x; r.v0[1] = t.m_v0.y; r.v0[2] = t.m_v0.z;
    r.orig_index = orig_index;
    const float e1[3] = { t.m_v1.x - t.m_v0.x, t.m_v1.y - t.m_v0.y, t.m_v1.z - t.m_v0.z };
    const float e2[3] = { t.m_v2.x - t.m_v0.x, t.m_v2.y - t.m_v0.y, t.m_v2.z - t.m_v0.z };
    for (int k = 0; k < 3; ++k) { r.e1[k] = e1[k]; r.e2[k] = e2[k]; }
    const double m = ((double)fabsf(e1[0]) + (double)fabsf(e1[1]) + (double)fabsf(e1[2])) * ((double)fabsf(e2[0]) + (double)fabsf(e2[1]) + (double)fabsf(e2[2]));
    const float f = (float)(32.0 * 5.9604644775390625e-8 * m);
    r.cull_margin = __int_as_float(__float_as_int(f) + 1); // nextafter towards +inf for f >= 0
    r.pad1 = 0;
}

__device__ __forceinline__ void write_normals(TriNormals& n, const FfTriangle& t)
{
    n.n0[0] = t.m_n0.x; n.n0[1] = t.m_n0.y; n.n0[2] = t.m_n0.z; n.pad0 = 0.f;
    n.n1[0] = t.m_n1.x; n.n1[1] = t.m_n1.y; n.n1[2] = t.m_n1.z; n.pad1 = 0.f;
    n.n2[0] = t.m_n2.x; n.n2[1] = t.m_n2.y; n.n2[2] = t.m_n2.z; n.pad2 = 0.f;
}

__global__ __launch_bounds__(kBuildBlock) void records_kernel(const FfTriangle* __restrict__ src, const uint32_t* __restrict__ vals, int T, int tri_first,
                                                               const int* __restrict__ leaf_position, TriRecord* __restrict__ tris,
                                                               TriNormals* __restrict__ normals)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    const int orig = (int)vals[j];
    const int dst = tri_first + (leaf_position ? leaf_position[j] : j);
    TriRecord r;
    write_record(r, src[orig], orig);
    tris[dst] = r;
    TriNormals n;
    write_normals(n, src[orig]);
    normals[dst] = n;
}

// ---- refit ------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kBuildBlock) void refresh_records_kernel(const FfTriangle* __restrict__ src, int T, int tri_first, TriRecord* __restrict__ tris,
                                                                       TriNormals* __restrict__ normals)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    const int orig = tris[tri_first + j].orig_index;
    if (orig < 0 || orig >= T) return; // never true for records this library wrote
    TriRecord r;
    write_record(r, src[orig], orig);
    tris[tri_first + j] = r;
    TriNormals n;
    write_normals(n, src[orig]);
    normals[tri_first + j] = n;
}

__global__ __launch_bounds__(kBuildBlock) void link_parents_kernel(const BvhNode* __restrict__ nodes, int node_first, int node_count, int* __restrict__ parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_count) return;
    const BvhNode& nd = nodes[node_first + i];
    const int l = nd.left, r = nd.right;
    if (l >= 0 && l >= node_first && l < node_first + node_count) parent[l - node_first] = ((node_first + i) << 1) | 0;
    if (r >= 0 && r >= node_first && r < node_first + node_count) parent[r - node_first] = ((node_first + i) << 1) | 1;
}

__device__ __forceinline__ void leaf_box(const FfTriangle* __restrict__ src, const TriRecord* __restrict__ tris, int link, float pad, float* mn, float* mx)
{
    const int ref = ~link, f = ref >> 3, count = (ref & 7) + 1;
    Box6 b = empty_box();
    for (int k = 0; k < count; ++k) {
        const FfTriangle& t = src[tris[f + k].orig_index];
        grow(b, t.m_v0);
        grow(b, t.m_v1);
        grow(b, t.m_v2);
    }
    for (int k = 0; k < 3; ++k) {
        mn[k] = b.mn[k] - pad;
        mx[k] = b.mx[k] + pad;
    }
}

// One thread per inner node: write the boxes of its leaf children, then carry completed nodes upward.
__global__ __launch_bounds__(kBuildBlock) void refit_kernel(const FfTriangle* __restrict__ src, const TriRecord* __restrict__ tris, const int* __restrict__ bounds,
                                                             int node_first, int node_count, const int* __restrict__ parent, BvhNode* nodes, int* arrivals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_count) return;
    const float pad = mesh_pad(bounds);
    int n = node_first + i;
    BvhNode* nd = nodes + n;
    const int l = nd->left, r = nd->right;
    int leaves = 0;
    {
        // the boxes of its leaf children (agent-scope stores: a thread on another XCD reads them when it completes this node)
        float mn[3], mx[3];
        if (l < 0) { leaf_box(src, tris, l, pad, mn, mx); for (int k = 0; k < 3; ++k) { coherent_store(nd->lmin + k, mn[k]); coherent_store(nd->lmax + k, mx[k]); } ++leaves; }
        if (r < 0) { leaf_box(src, tris, r, pad, mn, mx); for (int k = 0; k < 3; ++k) { coherent_store(nd->rmin + k, mn[k]); coherent_store(nd->rmax + k, mx[k]); } ++leaves; }
    }
    if (leaves == 0) return; // both boxes come from below
    release_arrival();
    if (arrive(&arrivals[i], leaves) + leaves < 2) return;
    for (int guard = 0; guard < 4096; ++guard) {
        // node n is complete: both of its child boxes are final
        acquire_arrival();
        const int p = parent[n - node_first];
        if (p < 0) return; // the root
        const float* a = reinterpret_cast<const float*>(nodes + n);
        float mn[3], mx[3];
        for (int k = 0; k < 3; ++k) {
            // (the child boxes already carry the padding)
            mn[k] = fminf(coherent_load(a + k), coherent_load(a + 8 + k));
            mx[k] = fmaxf(coherent_load(a + 4 + k), coherent_load(a + 12 + k));
        }
        const int pn = p >> 1, side = p & 1;
        BvhNode* pd = nodes + pn;
        float* dmn = side == 0 ? pd->lmin : pd->rmin;
        float* dmx = side == 0 ? pd->lmax : pd->rmax;
        for (int k = 0; k < 3; ++k) { coherent_store(dmn + k, mn[k]); coherent_store(dmx + k, mx[k]); }
        release_arrival();
        if (arrive(&arrivals[pn - node_first]) + 1 < 2) return;
        n = pn;
    }
}

// ---- 4-wide collapse ---------------------------------------------------------------------------------------------------
//
// Which binary nodes become 4-wide nodes ("roots") and which are absorbed into the 4-wide node above them is chosen to
// minimise the summed surface area of the 4-wide nodes - the expected number of node visits of a random ray - by the dynamic
// programme of Ylitie, Karras & Laine 2017 (there for 8-wide nodes): C(n, i) = least cost of representing the subtree of binary
// node n by at most i slots of its parent 4-wide node,
//     C(n, 1) = area(n) + min_k [ C(left, k) + C(right, 4 - k) ]                     (n becomes a 4-wide node itself)
//     C(n, i) = min( C(n, i - 1), min_k [ C(left, k) + C(right, i - k) ] )          (n is absorbed: its children share i slots)
// with C(leaf, .) = 0.  Against the fixed rule "every node at even depth, slots = grandchildren" (the fallback for trees deeper
// than 62 binary levels) the benchmark meshes get 15-30 % fewer 4-wide nodes with 3.5 instead of 3.0 slots in use and 5-10 %
// fewer expected visits.  The roles are kept per mesh (one byte per binary node): a refit changes boxes, not the topology.

// role bytes
constexpr unsigned char kRoleAbsorbed = 0, kRoleRoot = 1;

// decision word of a binary node: for budget b = 1..4 five bits at (b - 1) * 5: [1:0] slots given to the left child (0: the node
// becomes a 4-wide node and takes ONE slot), [4:2] the budget actually used (<= b); bits [21:20]: as a 4-wide node, the slots its
// left child gets (1..3; the right one gets the rest of 4)
__device__ __forceinline__ unsigned dec_left(unsigned d, int b) { return (d >> ((b - 1) * 5)) & 3u; }
__device__ __forceinline__ unsigned dec_used(unsigned d, int b) { return (d >> ((b - 1) * 5 + 2)) & 7u; }
__device__ __forceinline__ unsigned dec_root_left(unsigned d) { return (d >> 20) & 3u; }

__device__ __forceinline__ float half_area_of(const BvhNode& nd)
{
    const float dx = fmaxf(nd.lmax[0], nd.rmax[0]) - fminf(nd.lmin[0], nd.rmin[0]);
    const float dy = fmaxf(nd.lmax[1], nd.rmax[1]) - fminf(nd.lmin[1], nd.rmin[1]);
    const float dz = fmaxf(nd.lmax[2], nd.rmax[2]) - fminf(nd.lmin[2], nd.rmin[2]);
    return dx * dy + dy * dz + dz * dx;
}

// Bottom-up: a thread starts at every node whose children are both leaves and climbs while it is the last child to arrive.
// (The values are a pure function of the tree: the order of arrival does not show in the result.)
__global__ __launch_bounds__(kBuildBlock) void collapse_cost_kernel(const BvhNode* __restrict__ nodes, int node_first, int node_count,
                                                                     const int* __restrict__ parent, float4* __restrict__ cost, unsigned* __restrict__ decision,
                                                                     int* __restrict__ arrivals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_count) return;
    {
        const BvhNode& nd = nodes[node_first + i];
        if (nd.left >= 0 || nd.right >= 0) return; // an inner child will bring this node its turn
    }
    int n = i;
    for (int guard = 0; guard < 4096; ++guard) {
        const BvhNode nd = nodes[node_first + n];
        float cl[5] = { 0.f, 0.f, 0.f, 0.f, 0.f }, cr[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
        // (written by other threads, possibly on other CUs, before the arrival counter was bumped: read past this CU's L1)
        if (nd.left >= 0) {
            const volatile float* c = reinterpret_cast<const volatile float*>(&cost[nd.left - node_first]);
            cl[1] = c[0]; cl[2] = c[1]; cl[3] = c[2]; cl[4] = c[3];
        }
        if (nd.right >= 0) {
            const volatile float* c = reinterpret_cast<const volatile float*>(&cost[nd.right - node_first]);
            cr[1] = c[0]; cr[2] = c[1]; cr[3] = c[2]; cr[4] = c[3];
        }
        auto distribute = [&](int b, int& best_k) {
            float best = __builtin_huge_valf();
            best_k = 1;
            for (int k = 1; k < b; ++k) {
                const float v = cl[k] + cr[b - k];
                if (v < best) { best = v; best_k = k; } // (ties: the smallest k, deterministic)
            }
            return best;
        };
        float c[5];
        unsigned d = 0u;
        int k4;
        c[1] = half_area_of(nd) + distribute(4, k4);
        d |= (unsigned)k4 << 20; // as a 4-wide node
        d |= (0u | (1u << 2)) << 0; // budget 1: be a 4-wide node
        for (int b = 2; b <= 4; ++b) {
            int k;
            const float v = distribute(b, k);
            if (v < c[b - 1]) {
                c[b] = v;
                d |= ((unsigned)k | ((unsigned)b << 2)) << ((b - 1) * 5);
            } else {
                c[b] = c[b - 1];
                d |= ((d >> ((b - 2) * 5)) & 31u) << ((b - 1) * 5);
            }
        }
        {
            float* out = reinterpret_cast<float*>(&cost[n]);
            for (int b = 0; b < 4; ++b) coherent_store(out + b, c[b + 1]);
        }
        decision[n] = d; // (read by the NEXT kernel only)
        const int p = parent[n];
        if (p < 0) return;
        const int pn = (p >> 1) - node_first;
        const BvhNode& pd = nodes[node_first + pn];
        const int need = (pd.left >= 0 ? 1 : 0) + (pd.right >= 0 ? 1 : 0);
        release_arrival();
        if (arrive(&arrivals[pn]) + 1 < need) return;
        acquire_arrival();
        n = pn;
    }
}

// Top-down, one thread per binary node: its path from the root (side bits collected on the way up), replayed with the budgets
// the decisions hand down.  role[i] = the node becomes a 4-wide node; counters[3] = depth of the 4-wide tree; counters[2] != 0:
// some node sits deeper than 62 levels and the whole mesh falls back to the parity rule (role_par).
__global__ __launch_bounds__(kBuildBlock) void collapse_role_kernel(const BvhNode* __restrict__ nodes, int node_first, int node_count,
                                                                     const int* __restrict__ parent, const unsigned* __restrict__ decision,
                                                                     unsigned char* __restrict__ role_dp, unsigned char* __restrict__ role_par, int* counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_count) return;
    unsigned long long sides = 0ull;
    int depth = 0;
    for (int p = parent[i]; p >= 0 && depth < 4096; p = parent[(p >> 1) - node_first]) {
        if (depth < 64) sides = (sides << 1) | (unsigned long long)(p & 1); // the LAST step up is the FIRST step down: it ends in bit 0
        ++depth;
    }
    const bool even = (depth & 1) == 0;
    role_par[i] = even ? kRoleRoot : kRoleAbsorbed;
    if (even) atomicMax(&counters[1], depth / 2 + 1);
    if (depth > 62) {
        counters[2] = 1;
        role_dp[i] = kRoleAbsorbed;
        return;
    }
    // replay: (n, as a 4-wide node?  budget otherwise)
    int n = 0; // the mesh's root is its first node
    bool is_root = true;
    int budget = 4, depth4 = 1;
    for (int s = 0; s < depth; ++s) {
        const unsigned d = decision[n];
        const int side = (int)((sides >> s) & 1ull);
        int left_slots, total;
        if (is_root) {
            left_slots = (int)dec_root_left(d);
            total = 4;
        } else {
            left_slots = (int)dec_left(d, budget);
            total = (int)dec_used(d, budget);
        }
        const int child_budget = side == 0 ? left_slots : total - left_slots;
        const BvhNode& nd = nodes[node_first + n];
        n = (side == 0 ? nd.left : nd.right) - node_first;
        budget = child_budget;
        is_root = dec_left(decision[n], budget) == 0u; // given `budget` slots the child takes one as a 4-wide node, or is absorbed
        if (is_root) ++depth4;
    }
    role_dp[i] = is_root ? kRoleRoot : kRoleAbsorbed;
    if (is_root) atomicMax(&counters[3], depth4);
}

// flag[i] = 1 for the binary nodes that become 4-wide nodes: the kept roles, or this call's (optimal, or parity for a tree too deep)
__global__ __launch_bounds__(kBuildBlock) void collapse_flag_kernel(int node_count, const unsigned char* __restrict__ role_dp, const unsigned char* __restrict__ role_par,
                                                                     const int* __restrict__ counters, unsigned char* __restrict__ role_keep,
                                                                     uint32_t* __restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_count) return;
    unsigned char r;
    if (role_dp) {
        r = counters[2] != 0 ? role_par[i] : role_dp[i];
        role_keep[i] = r;
    } else {
        r = role_keep[i];
    }
    flag[i] = r == kRoleRoot ? 1u : 0u;
}

__global__ __launch_bounds__(kBuildBlock) void collapse_emit_kernel(const BvhNode* __restrict__ nodes, int node_first, int node_count,
                                                                     const uint32_t* __restrict__ flag, const uint32_t* __restrict__ index4,
                                                                     Bvh4Node* __restrict__ nodes4, int node4_first)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_count || flag[i] == 0u) return;
    Bvh4Node out;
    int slots = 0;
    auto add = [&](const float* mn, const float* mx, int link) {
        for (int q = 0; q < slots; ++q)
            if (link < 0 && out.link[q] == link) return; // a mesh that fits one leaf carries that leaf on both links
        if (slots >= 4) return; // (cannot happen: the roles were derived from slot budgets)
        for (int k = 0; k < 3; ++k) {
            out.mn[k][slots] = mn[k];
            out.mx[k][slots] = mx[k];
        }
        out.link[slots] = link < 0 ? link : (int)index4[link - node_first];
        ++slots;
    };
    // the slots of this 4-wide node: the children of binary node i, an absorbed child replaced by ITS children, and so on (at most
    // three absorbed nodes below one 4-wide node).  The order of the slots inside a node is a fixed function of the tree and of no
    // consequence: the traversal sorts the slots it hits by distance.
    int todo[4], top = 0;
    todo[top++] = node_first + i;
    while (top > 0) {
        const BvhNode nd = nodes[todo[--top]];
        for (int side = 0; side < 2; ++side) {
            const int link = side == 0 ? nd.left : nd.right;
            if (link >= 0 && flag[link - node_first] == 0u) {
                if (top < 4) todo[top++] = link; // an absorbed child: its children take its place
            } else {
                add(side == 0 ? nd.lmin : nd.rmin, side == 0 ? nd.lmax : nd.rmax, link);
            }
        }
    }
    for (int q = slots; q < 4; ++q) {
        for (int k = 0; k < 3; ++k) {
            out.mn[k][q] = __builtin_huge_valf();
            out.mx[k][q] = -__builtin_huge_valf();
        }
        out.link[q] = kEmptyLink;
    }
    nodes4[node4_first + (int)index4[i]] = out;
}

__global__ void collapse_count_kernel(int node_count, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ index4, int* counters, int fresh_roles)
{
    counters[0] = (int)(index4[node_count - 1] + flag[node_count - 1]);
    if (fresh_roles && counters[2] == 0) counters[1] = counters[3]; // depth of the 4-wide tree under the optimal roles
}

// ---- scratch -----------------------------------------------------------------------------------------------------------

struct Carver {
    char* p;
    size_t used = 0;
    explicit Carver(void* base) : p(static_cast<char*>(base)) {}
    template <class T>
    T* take(size_t count)
    {
        used = (used + 255) & ~(size_t)255;
        T* out = p ? reinterpret_cast<T*>(p + used) : nullptr;
        used += count * sizeof(T);
        return out;
    }
};

int ensure_scratch(BuildScratch& s, size_t bytes)
{
    if (s.capacity >= bytes && s.base) return FF_OK;
    if (s.base) (void)hipFree(s.base);
    s.base = nullptr;
    s.capacity = 0;
    if (hipMalloc(&s.base, bytes) != hipSuccess) {
        s.base = nullptr;
        return fail(FF_ERR_OOM, "BVH builder: cannot allocate %zu bytes of device scratch", bytes);
    }
    s.capacity = bytes;
    return FF_OK;
}

const TreeletTables& treelet_tables()
{
    static const TreeletTables tables = [] {
        TreeletTables t;
        std::memset(&t, 0, sizeof t);
        int np = 0, ns = 0;
        for (int k = 2; k <= kTreeletLeaves; ++k) {
            t.pair_first[k] = (unsigned short)np;
            t.subset_first[k] = (unsigned short)ns;
            for (int s = 1; s < (1 << kTreeletLeaves); ++s) {
                if (__builtin_popcount((unsigned)s) != k) continue;
                t.subset[ns++] = (unsigned char)s;
                const int delta = (s - 1) & s; // s without its lowest member
                for (int p = (-delta) & s; p != 0; p = (p - delta) & s) t.pair[np++] = (unsigned short)(s | (p << kTreeletLeaves));
            }
        }
        t.pair_first[kTreeletLeaves + 1] = (unsigned short)np;
        t.subset_first[kTreeletLeaves + 1] = (unsigned short)ns;
        return t;
    }();
    return tables;
}

struct BuildBuffers {
    int* bounds;
    int* counters;
    uint64_t *keys_in, *keys_out;
    uint32_t *vals_in, *vals_out;
    int *left, *right, *first, *last, *node_parent, *leaf_parent, *arrivals, *new_index;
    float* boxes;
    uint32_t *depth_in, *depth_out, *ids_in, *ids_out;
    int *clusters_a, *clusters_b, *nearest, *merged, *sizes, *leaf_position; // PLOC
    uint32_t *keep, *position;
    float* cost; // treelet restructuring: SAH cost of every subtree (reinsertion passes: the gain of every node's best move)
    int *r_target, *r_ok, *r_go; // reinsertion passes, per slot
    unsigned long long* r_lock;
    TreeletTables* tables;
    void* sort_temp;
    size_t sort_temp_bytes;
    size_t total;
};

BuildBuffers carve_build(void* base, int T, size_t sort_temp_bytes)
{
    Carver c(base);
    BuildBuffers b;
    const size_t n = (size_t)T, m = (size_t)(T > 1 ? T - 1 : 1);
    b.bounds = c.take<int>(8);
    b.counters = c.take<int>(8);
    b.keys_in = c.take<uint64_t>(n);
    b.keys_out = c.take<uint64_t>(n);
    b.vals_in = c.take<uint32_t>(n);
    b.vals_out = c.take<uint32_t>(n);
    b.left = c.take<int>(m);
    b.right = c.take<int>(m);
    b.first = c.take<int>(m);
    b.last = c.take<int>(m);
    b.node_parent = c.take<int>(m);
    b.leaf_parent = c.take<int>(n);
    b.arrivals = c.take<int>(m);
    b.new_index = c.take<int>(m);
    b.boxes = c.take<float>((2 * n) * 6);
    b.depth_in = c.take<uint32_t>(m);
    b.depth_out = c.take<uint32_t>(m);
    b.ids_in = c.take<uint32_t>(m);
    b.ids_out = c.take<uint32_t>(m);
    b.clusters_a = c.take<int>(n);
    b.clusters_b = c.take<int>(n);
    b.nearest = c.take<int>(n);
    b.merged = c.take<int>(n);
    b.sizes = c.take<int>(m);
    b.leaf_position = c.take<int>(n);
    b.keep = c.take<uint32_t>(n);
    b.position = c.take<uint32_t>(n);
    b.cost = c.take<float>(2 * n);
    b.r_target = c.take<int>(2 * n);
    b.r_ok = c.take<int>(2 * n);
    b.r_go = c.take<int>(2 * n);
    b.r_lock = c.take<unsigned long long>(2 * n);
    b.tables = c.take<TreeletTables>(1);
    b.sort_temp = c.take<char>(sort_temp_bytes);
    b.sort_temp_bytes = sort_temp_bytes;
    b.total = c.used + 256;
    return b;
}

} // namespace

void free_build_scratch(BuildScratch& s)
{
    if (s.base) (void)hipFree(s.base);
    s.base = nullptr;
    s.capacity = 0;
}

int gpu_build_mesh(hipStream_t stream, BuildScratch& scratch, const FfTriangle* d_src, int T, int tri_first, int node_base, int max_leaf,
                   TriRecord* d_tris, TriNormals* d_normals, BvhNode* d_nodes, MeshBuildInfo* out, bool ploc)
{
    if (T <= max_leaf || T < 2) return fail(FF_ERR_INVALID_ARG, "gpu_build_mesh: %d triangles fit one leaf", T);
    if (max_leaf < 1 || max_leaf > 8) return fail(FF_ERR_INVALID_ARG, "max_leaf_tris must be 1..8");

    // temporary storage of the two sorts (query with null storage)
    size_t temp_codes = 0, temp_depth = 0;
    FFB_HIP(rocprim::radix_sort_pairs(nullptr, temp_codes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)T, 0u, 63u,
                                      stream));
    FFB_HIP(rocprim::radix_sort_pairs(nullptr, temp_depth, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)(T - 1), 0u,
                                      8u, stream));
    size_t temp_scan = 0;
    FFB_HIP(rocprim::exclusive_scan(nullptr, temp_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)T, rocprim::plus<uint32_t>(), stream));
    size_t sort_temp = temp_codes > temp_depth ? temp_codes : temp_depth;
    if (temp_scan > sort_temp) sort_temp = temp_scan;
    const BuildBuffers sizes = carve_build(nullptr, T, sort_temp);
    int st = ensure_scratch(scratch, sizes.total);
    if (st != FF_OK) return st;
    const BuildBuffers b = carve_build(scratch.base, T, sort_temp);

    const int tri_grid = grid_for(T), node_grid = grid_for(T - 1);
    init_bounds_kernel<<<1, 64, 0, stream>>>(b.bounds);
    FFB_HIP(hipMemsetAsync(b.counters, 0, 8 * sizeof(int), stream));
    FFB_HIP(hipMemsetAsync(b.arrivals, 0, (size_t)(T - 1) * sizeof(int), stream));
    bounds_kernel<<<tri_grid < 1024 ? tri_grid : 1024, kBuildBlock, 0, stream>>>(d_src, T, b.bounds);
    morton_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(d_src, T, b.bounds, b.keys_in, b.vals_in);
    size_t tb = b.sort_temp_bytes;
    FFB_HIP(rocprim::radix_sort_pairs(b.sort_temp, tb, b.keys_in, b.keys_out, b.vals_in, b.vals_out, (size_t)T, 0u, 63u, stream));
    const int* leaf_position = nullptr;
    if (!ploc) {
        hierarchy_kernel<<<node_grid, kBuildBlock, 0, stream>>>(b.keys_out, T, b.left, b.right, b.first, b.last, b.node_parent, b.leaf_parent);
        fit_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(d_src, b.vals_out, T, b.left, b.right, b.node_parent, b.leaf_parent, b.boxes, b.arrivals);
    } else {
        leaf_boxes_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(d_src, b.vals_out, T, b.boxes, b.clusters_a);
        int* cur = b.clusters_a;
        int* nxt = b.clusters_b;
        int n = T;
        const int radius = ploc_radius();
        for (int iter = 0; n > 1; ++iter) {
            if (iter > 4 * 64) return fail(FF_ERR_HIP, "gpu_build_mesh: clustering did not converge (%d clusters left)", n);
            const int g = grid_for(n);
            ploc_neighbour_kernel<<<g, kBuildBlock, 0, stream>>>(n, T, cur, b.boxes, b.nearest, radius);
            ploc_merge_kernel<<<g, kBuildBlock, 0, stream>>>(n, T, cur, b.nearest, b.boxes, b.left, b.right, b.node_parent, b.leaf_parent, b.sizes, b.counters,
                                                             b.merged, b.keep);
            size_t ts = b.sort_temp_bytes;
            FFB_HIP(rocprim::exclusive_scan(b.sort_temp, ts, b.keep, b.position, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
            ploc_compact_kernel<<<g, kBuildBlock, 0, stream>>>(n, b.merged, b.keep, b.position, nxt, b.counters);
            int left_over = 0;
            FFB_HIP(hipMemcpyAsync(&left_over, b.counters + 3, sizeof(int), hipMemcpyDeviceToHost, stream));
            FFB_HIP(hipStreamSynchronize(stream));
            if (left_over < 1 || left_over >= n) return fail(FF_ERR_HIP, "gpu_build_mesh: clustering step went from %d to %d clusters", n, left_over);
            n = left_over;
            int* t = cur;
            cur = nxt;
            nxt = t;
        }
    }
    // Treelet restructuring passes (FF_TREELET_PASSES; default 2 for the LBVH, none for PLOC), then every node's
    // triangle range and every leaf's place in the leaf order from the final topology.
    int passes = ploc ? 0 : 2; // (on PLOC trees a pass buys 1 % on the 5 000-triangle meshes and LOSES 4 % on the regular 983 040-triangle sphere)
    if (const char* e = std::getenv("FF_TREELET_PASSES")) passes = std::max(0, std::min(8, std::atoi(e)));
    if (passes > 0 || ploc) {
        if (!ploc) sizes_from_ranges_kernel<<<node_grid, kBuildBlock, 0, stream>>>(T, b.first, b.last, b.sizes);
        if (passes > 0) FFB_HIP(hipMemcpyAsync(b.tables, &treelet_tables(), sizeof(TreeletTables), hipMemcpyHostToDevice, stream));
        for (int pass = 0; pass < passes; ++pass) {
            FFB_HIP(hipMemsetAsync(b.arrivals, 0, (size_t)(T - 1) * sizeof(int), stream));
            treelet_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.sizes, b.boxes, b.cost, b.arrivals, b.counters,
                                                              b.tables);
        }
    }
    // Reinsertion passes (reinsert_find_kernel; FF_GPU_REINSERT: how many).  Eight by default for meshes of up to 65 536 triangles, four beyond, on both builders: same box, share of the
    // host tree's trace rate, LBVH 87 -> 95 % on C2 (8 passes: 97 %, where it saturates), 86 -> 90-94 % on C3, 90 -> 95 % on the
    // 983 040-triangle sphere; PLOC 91 -> 93 %, 90 -> 95 %, 90 -> 93 %; a pass costs 0.18 ms on 5 172 triangles, 1.25 ms on 983 040
    // (profiles/r04_r_*).
    int reinsert = T <= 65536 ? 8 : 4; // (small meshes: where the gain saturates, 0.18 ms a pass)
    if (const char* e = std::getenv("FF_GPU_REINSERT")) reinsert = std::max(0, std::min(64, std::atoi(e)));
    if (reinsert > 0 && T >= 8) {
        if (!ploc && passes == 0) sizes_from_ranges_kernel<<<node_grid, kBuildBlock, 0, stream>>>(T, b.first, b.last, b.sizes);
        const int slot_grid = grid_for(2 * T - 1);
        for (int pass = 0; pass < reinsert; ++pass) {
            reinsert_find_kernel<<<slot_grid, kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.boxes, b.r_target, b.cost, b.r_lock);
            reinsert_lock_kernel<<<slot_grid, kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.r_target, b.cost, b.r_lock);
            reinsert_hold_kernel<<<slot_grid, kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.r_target, b.cost, b.r_lock, b.r_ok);
            reinsert_check_kernel<<<slot_grid, kBuildBlock, 0, stream>>>(T, b.node_parent, b.leaf_parent, b.r_target, b.r_ok, b.r_go);
            reinsert_apply_kernel<<<slot_grid, kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.r_target, b.r_go, b.counters);
            FFB_HIP(hipMemsetAsync(b.arrivals, 0, (size_t)(T - 1) * sizeof(int), stream));
            reinsert_refit_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.boxes, b.sizes, b.arrivals);
        }
    }
    if (passes > 0 || ploc || reinsert > 0) {
        ploc_ranges_kernel<<<grid_for(2 * T - 1), kBuildBlock, 0, stream>>>(T, b.left, b.right, b.node_parent, b.leaf_parent, b.sizes, b.first, b.last,
                                                                          b.leaf_position);
        leaf_position = b.leaf_position;
    }
    rank_key_kernel<<<node_grid, kBuildBlock, 0, stream>>>(T, max_leaf, b.first, b.last, b.node_parent, b.depth_in, b.ids_in, b.counters);
    tb = b.sort_temp_bytes;
    FFB_HIP(rocprim::radix_sort_pairs(b.sort_temp, tb, b.depth_in, b.depth_out, b.ids_in, b.ids_out, (size_t)(T - 1), 0u, 8u, stream));
    int host_counters[2] = { 0, 0 };
    FFB_HIP(hipMemcpyAsync(host_counters, b.counters, sizeof host_counters, hipMemcpyDeviceToHost, stream));
    FFB_HIP(hipStreamSynchronize(stream));
    const int emitted = host_counters[0], depth = host_counters[1];
    if (emitted < 1 || emitted > T - 1) return fail(FF_ERR_HIP, "gpu_build_mesh: inconsistent node count %d for %d triangles", emitted, T);
    if (depth >= (int)kNotEmitted) return fail(FF_ERR_UNSUPPORTED, "gpu_build_mesh: tree depth %d exceeds the builder's limit", depth);
    rank_scatter_kernel<<<grid_for(emitted), kBuildBlock, 0, stream>>>(emitted, b.ids_out, b.new_index);
    emit_kernel<<<grid_for(emitted), kBuildBlock, 0, stream>>>(T, emitted, max_leaf, tri_first, node_base, b.bounds, b.ids_out, b.new_index, b.left, b.right,
                                                               b.first, b.last, b.boxes, leaf_position, d_nodes);
    records_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(d_src, b.vals_out, T, tri_first, leaf_position, d_tris, d_normals);
    FFB_HIP(hipGetLastError());
    out->root = node_base;
    out->node_count = emitted;
    out->depth = depth;
    return FF_OK;
}

int gpu_collapse_mesh(hipStream_t stream, BuildScratch& scratch, const BvhNode* d_nodes, int node_first, int node_count, const int* d_parent,
                      Bvh4Node* d_nodes4, int node4_first, unsigned char* d_role, Collapse4Info* info)
{
    if (node_count <= 0) {
        if (info) *info = Collapse4Info();
        return FF_OK;
    }
    const bool fresh = info != nullptr; // after a build: choose the roles; after a refit: keep them (same topology, new boxes)
    const bool parity_only = std::getenv("FF_COLLAPSE_PARITY") != nullptr; // experiments: the fixed rule of rounds 1-2
    size_t temp_scan = 0;
    FFB_HIP(rocprim::exclusive_scan(nullptr, temp_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)node_count, rocprim::plus<uint32_t>(), stream));
    Carver probe(nullptr);
    probe.take<int>(8);
    probe.take<uint32_t>((size_t)node_count);
    probe.take<uint32_t>((size_t)node_count);
    probe.take<char>(temp_scan);
    probe.take<float4>((size_t)node_count);
    probe.take<unsigned>((size_t)node_count);
    probe.take<int>((size_t)node_count);
    probe.take<unsigned char>((size_t)node_count);
    probe.take<unsigned char>((size_t)node_count);
    int st = ensure_scratch(scratch, probe.used + 256);
    if (st != FF_OK) return st;
    Carver c(scratch.base);
    int* counters = c.take<int>(8);
    uint32_t* flag = c.take<uint32_t>((size_t)node_count);
    uint32_t* index4 = c.take<uint32_t>((size_t)node_count);
    void* temp = c.take<char>(temp_scan);
    float4* cost = c.take<float4>((size_t)node_count);
    unsigned* decision = c.take<unsigned>((size_t)node_count);
    int* arrivals = c.take<int>((size_t)node_count);
    unsigned char* role_dp = c.take<unsigned char>((size_t)node_count);
    unsigned char* role_par = c.take<unsigned char>((size_t)node_count);
    FFB_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int), stream));
    const int grid = grid_for(node_count);
    if (fresh) {
        FFB_HIP(hipMemsetAsync(arrivals, 0, (size_t)node_count * sizeof(int), stream));
        collapse_cost_kernel<<<grid, kBuildBlock, 0, stream>>>(d_nodes, node_first, node_count, d_parent, cost, decision, arrivals);
        collapse_role_kernel<<<grid, kBuildBlock, 0, stream>>>(d_nodes, node_first, node_count, d_parent, decision, role_dp, role_par, counters);
        if (parity_only) FFB_HIP(hipMemsetAsync(counters + 2, 0xff, sizeof(int), stream));
        collapse_flag_kernel<<<grid, kBuildBlock, 0, stream>>>(node_count, role_dp, role_par, counters, d_role, flag);
    } else {
        collapse_flag_kernel<<<grid, kBuildBlock, 0, stream>>>(node_count, nullptr, nullptr, counters, d_role, flag);
    }
    FFB_HIP(rocprim::exclusive_scan(temp, temp_scan, flag, index4, 0u, (size_t)node_count, rocprim::plus<uint32_t>(), stream));
    collapse_emit_kernel<<<grid, kBuildBlock, 0, stream>>>(d_nodes, node_first, node_count, flag, index4, d_nodes4, node4_first);
    FFB_HIP(hipGetLastError());
    if (info) {
        collapse_count_kernel<<<1, 1, 0, stream>>>(node_count, flag, index4, counters, 1);
        int host[2] = { 0, 0 };
        FFB_HIP(hipMemcpyAsync(host, counters, sizeof host, hipMemcpyDeviceToHost, stream));
        FFB_HIP(hipStreamSynchronize(stream));
        info->node_count = host[0];
        info->depth = host[1];
        if (host[0] < 1 || host[0] > node_count) return fail(FF_ERR_HIP, "gpu_collapse_mesh: inconsistent 4-wide node count %d for %d binary nodes", host[0], node_count);
    }
    return FF_OK;
}

int gpu_link_parents(hipStream_t stream, const BvhNode* d_nodes, int node_first, int node_count, int* d_parent)
{
    if (node_count <= 0) return FF_OK;
    FFB_HIP(hipMemsetAsync(d_parent, 0xff, (size_t)node_count * sizeof(int), stream));
    link_parents_kernel<<<grid_for(node_count), kBuildBlock, 0, stream>>>(d_nodes, node_first, node_count, d_parent);
    FFB_HIP(hipGetLastError());
    return FF_OK;
}

int gpu_refit_mesh(hipStream_t stream, BuildScratch& scratch, const FfTriangle* d_src, int T, int tri_first, int node_first, int node_count,
                   const int* d_parent, TriRecord* d_tris, TriNormals* d_normals, BvhNode* d_nodes)
{
    if (T <= 0 || node_count <= 0) return FF_OK;
    Carver probe(nullptr);
    probe.take<int>(8);
    probe.take<int>((size_t)node_count);
    int st = ensure_scratch(scratch, probe.used + 256);
    if (st != FF_OK) return st;
    Carver c(scratch.base);
    int* bounds = c.take<int>(8);
    int* arrivals = c.take<int>((size_t)node_count);
    init_bounds_kernel<<<1, 64, 0, stream>>>(bounds);
    FFB_HIP(hipMemsetAsync(arrivals, 0, (size_t)node_count * sizeof(int), stream));
    const int tri_grid = grid_for(T);
    bounds_kernel<<<tri_grid < 1024 ? tri_grid : 1024, kBuildBlock, 0, stream>>>(d_src, T, bounds);
    refresh_records_kernel<<<tri_grid, kBuildBlock, 0, stream>>>(d_src, T, tri_first, d_tris, d_normals);
    refit_kernel<<<grid_for(node_count), kBuildBlock, 0, stream>>>(d_src, d_tris, bounds, node_first, node_count, d_parent, d_nodes, arrivals);
    FFB_HIP(hipGetLastError());
    return FF_OK;
}

} // namespace ff
