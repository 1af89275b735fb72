// ff_build.h — device-side BVH construction and refit.  Internal to the library.
//
// The host builder (ff_scene.cpp, binned SAH) is the default: best trees, built once per upload like the reference's
// one-off copy (kernel.cu:268-298).  For geometry that changes between frames the same node / triangle-record arrays
// can be produced on the GPU instead: an LBVH per mesh (Morton codes, rocPRIM radix sort, Karras 2012 hierarchy,
// bottom-up box fit, subtrees of <= max_leaf triangles collapsed into leaves, nodes renumbered level by level so that the
// LDS-resident prefix of the traversal kernel holds the top of the tree), or an existing tree can be refitted in place
// when only vertex positions moved.  Either way the renderer's results do not depend on the tree (every hit is decided
// by the exact reference arithmetic), which is what the parity tests check.
#pragma once

#include <hip/hip_runtime.h>

#include "ff_internal.h"

namespace ff {

// Grow-only device scratch arena owned by the FfState.
struct BuildScratch {
    void* base = nullptr;
    size_t capacity = 0;
};
void free_build_scratch(BuildScratch& s);

struct MeshBuildInfo {
    int root = -1;      // global inner-node index of the mesh's root
    int node_count = 0; // inner nodes written at [node_base, node_base + node_count)
    int depth = 0;      // deepest root-to-leaf path in inner nodes
};

// Upper bound on the inner nodes gpu_build_mesh writes for a mesh of `tri_count` triangles.
inline size_t gpu_build_max_nodes(int tri_count) { return tri_count > 1 ? (size_t)tri_count - 1 : 1; }

// Build the BVH of one mesh on the device.  d_src: the caller's Triangle array copied to the device (tri_count x 96 B).
// Writes TriRecords (and the parallel vertex-normal records) [tri_first, tri_first + tri_count) in leaf order and inner
// nodes from node_base on.  Synchronises the
// stream once (the node count of this mesh places the next one).  tri_count must exceed max_leaf (smaller meshes are a
// single leaf and are assembled on the host).  ploc = false: LBVH (Karras hierarchy on the Morton order); true: PLOC
// (agglomerative clustering on the Morton order: slower to build, close to SAH quality).
int gpu_build_mesh(hipStream_t stream, BuildScratch& scratch, const FfTriangle* d_src, int tri_count, int tri_first, int node_base, int max_leaf,
                   TriRecord* d_tris, TriNormals* d_normals, BvhNode* d_nodes, MeshBuildInfo* out, bool ploc = false);

// parent[n - node_first] = (parent node index << 1 | side) for every inner node n of the mesh, -1 for its root.
int gpu_link_parents(hipStream_t stream, const BvhNode* d_nodes, int node_first, int node_count, int* d_parent);

// Refit the tree of one mesh in place after its vertices moved (same triangle count and order as at build time):
// rewrites the mesh's TriRecords from d_src through their orig_index, recomputes every leaf box and propagates the
// boxes to the root.  Works on trees from either builder.  d_parent: from gpu_link_parents.  d_counters: node_count ints
// of scratch.  Asynchronous on `stream`.
int gpu_refit_mesh(hipStream_t stream, BuildScratch& scratch, const FfTriangle* d_src, int tri_count, int tri_first, int node_first, int node_count,
                   const int* d_parent, TriRecord* d_tris, TriNormals* d_normals, BvhNode* d_nodes);

// The 4-wide tree the trace kernels traverse, derived from one mesh's binary nodes [node_first, node_first + node_count)
// (any builder; node numbers grow level by level, root first).  Which binary nodes become 4-wide nodes and which are absorbed
// into the one above them (its slots are then their children) minimises the summed surface area of the 4-wide nodes: the
// dynamic programme of Ylitie et al. 2017 at width 4 (ff_build.hip; trees deeper than 62 levels: every node at even depth).
// 4-wide nodes keep the binary nodes' order (numbers grow with depth), are written from d_nodes4[node4_first] on, and link to
// each other RELATIVE to node4_first.  d_parent: this mesh's section of the parent array (gpu_link_parents).  d_role: one byte
// per binary node of the mesh, written with `info` (after a build: the call synchronises the stream once and returns the node
// count and the depth of the 4-wide tree) and READ without it (after a refit: same topology, new boxes; asynchronous).
struct Collapse4Info {
    int node_count = 0; // 4-wide nodes written
    int depth = 0;      // deepest root-to-leaf path in 4-wide nodes
};
int gpu_collapse_mesh(hipStream_t stream, BuildScratch& scratch, const BvhNode* d_nodes, int node_first, int node_count, const int* d_parent,
                      Bvh4Node* d_nodes4, int node4_first, unsigned char* d_role, Collapse4Info* info);

} // namespace ff
