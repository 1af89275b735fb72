// ff_internal.h — declarations shared by the library's translation units (not part of the ABI).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/firefly/ff_api.h"

namespace ff {

// ---- error reporting (replaces the reference's print-and-exit macro, utilities.h:27-37) ----
int fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// ---- device-side scene records ---------------------------------------------------------------
//
// All records are built on the host by the scene compiler (ff_scene.cpp) and copied once per upload.

// One 48-byte triangle record, stored in BVH-leaf order.  Only what intersectTriangle reads
// (kernel.cu:39-41) plus the triangle's index in the caller's array (kernel.cu:152).
struct TriRecord {
    float v0[3];
    int32_t orig_index; // index j in Geometry::m_triangles
    float e1[3];        // v1 - v0, the fp32 subtraction of kernel.cu:44 done once on the host (same bits)
    float cull_margin;  // if det >= EPSILON + cull_margin the back-face test of kernel.cu:49 cannot fire (see ff_scene.cpp)
    float e2[3];        // v2 - v0 (kernel.cu:45)
    int32_t pad1;
};
static_assert(sizeof(TriRecord) == 48, "48-byte triangle record");

// The vertex normals a Triangle carries (utilities.h:163-170), same order as the TriRecords; read only when a hit is
// shaded with FF_SHADE_DIFFUSE_PATH_SMOOTH.
struct TriNormals {
    float n0[3], pad0;
    float n1[3], pad1;
    float n2[3], pad2;
};
static_assert(sizeof(TriNormals) == 48, "48-byte vertex-normal record");

// One 64-byte inner BVH node: both child boxes (object space, conservatively padded) and both child links.
// link >= 0: index of an inner node (global node array).  link < 0: leaf, ~link = (first_tri << 3) | (count - 1),
// first_tri indexing the global TriRecord array.  A mesh that fits one leaf has both links on that leaf.
struct BvhNode {
    float lmin[3];
    int32_t left;
    float lmax[3];
    int32_t right;
    float rmin[3];
    int32_t pad0;
    float rmax[3];
    int32_t pad1;
};
static_assert(sizeof(BvhNode) == 64, "64-byte BVH node");

// One 112-byte node of the 4-wide tree the trace kernels traverse.  It is DERIVED from the binary tree above (which stays
// the representation the builders write and refit works on): every binary node at even depth becomes a 4-wide node whose
// slots are its grandchildren (a child that is a leaf fills one slot), see gpu_collapse_mesh.  Seven 16-byte quarters:
// the six box planes, each holding that coordinate for the four slots, and the four links.  A ray reads the three "near"
// and the three "far" planes picked by the signs of its direction, so a slab test is 6 FMAs + max3 + min3.
// link >= 0: index of a 4-wide node RELATIVE to the mesh's first one (its root is 0); link < 0: leaf, ~link =
// (first_tri << 3) | (count - 1) as in BvhNode; kEmptyLink: unused slot (its box is inverted: +inf / -inf, never hit).
struct Bvh4Node {
    float mn[3][4]; // [axis][slot]
    float mx[3][4];
    int32_t link[4];
};
static_assert(sizeof(Bvh4Node) == 112, "112-byte 4-wide BVH node");
constexpr int32_t kEmptyLink = 0x7fffffff;

// Per-geometry record (wave-uniform reads in the kernel).  Matrices are stored as xyz columns.
struct GeomRecord {
    // m_inverseModelMatrix columns 0..3 (xyz).  The w slots of columns 0..2 hold (column3 * 0.0f).xyz, the signed zero
    // glm adds when it transforms a direction (vec4 with w = 0, kernel.cu:138).
    float inv_c0[4], inv_c1[4], inv_c2[4], inv_c3[4];
    float mod_c0[4], mod_c1[4], mod_c2[4], mod_c3[4]; // m_modelMatrix columns (xyz)
    // inverse(transpose(model)) columns 0..2 (kernel.cu:117); w slots hold (column3 * 0.0f).xyz as above.
    float nrm_c0[4], nrm_c1[4], nrm_c2[4];
    float plane_n[4];                                 // m_normal (utilities.h:229)
    float albedo[4];                                  // BXDF::m_albedo
    float emission[4];                                // BXDF::m_emissiveColor * m_intensity (utilities.h:102)
    float wmin[4], wmax[4];                           // world-space AABB of the geometry, conservatively padded (pruning only)
    int32_t type;                                     // FfGeometryType
    int32_t bxdf_type;                                // FfBXDFType
    int32_t tri_first;                                // first TriRecord of this mesh
    int32_t tri_count;
    int32_t bvh_root;                                 // index of this mesh's first (root) node in the BINARY node array, -1 if none
    int32_t orig_index;                               // index i in the caller's Geometry[] (kernel.cu:151,162); records are
                                                      // stored in PROCESSING order: planes first, then meshes
    int32_t node4_first;                              // index of this mesh's root in the 4-wide node array (Bvh4Node links are relative to it)
    int32_t lds_nodes;                                // the mesh's 4-wide nodes [0, lds_nodes) (relative) are cached in LDS by the trace kernels ...
                                                      // ... at LDS node index lds_first + relative index; lds_first lives in the w lane of wmin (as int bits)
};
static_assert(sizeof(GeomRecord) == 16 * 16 + 32, "GeomRecord layout");

// Axis-aligned walls.  A plane whose model matrix maps the unit quad (kernel.cu:18) onto a rectangle parallel to two world axes
// (rotations by multiples of 90 degrees, any scale and translation: every wall of a box scene) is screened in WORLD space by a
// wave-uniform loop over this table (scan_records / screen_walls): parameter t = (c - o_k) / d_k of the plane x_k = c, two range
// checks on the other two coordinates.  The table travels in the kernel arguments (scalar loads, no LDS gathers, no registers
// per record); hits inside the screening margins and near ties fall back to the per-lane screen of the plane's record, which
// decides with the exact reference test, so results do not depend on the table.  Built by the host (build_wall_table).
constexpr int kMaxWalls = 16;
struct Wall {
    float c;      // the plane: world coordinate along its normal axis
    float cu, hu; // centre and half extent along the first of the other two axes (x: y, y: z, z: x)
    float cv, hv; // ... and along the second (x: z, y: x, z: y)
    int geom;     // record index of the plane
    int hi_geom1; // != 0: the entry holds two walls with this rectangle - this one at c and record hi_geom1 - 1 at hi_c > c
    float hi_c;
};
static_assert(sizeof(Wall) == 32, "32-byte wall");
struct WallTable {
    int count[3];   // walls normal to the world x axis: w[0 .. count[0]), then those normal to y, then to z
    unsigned mask;  // bit g: plane record g is in the table
    float margin_s; // S: length scale of the screening margins (world units: the walls' sizes and distances from the origin)
    float graze;    // a ray with |d_k| < graze * |d| is too close to parallel to a wall normal to k for the fast form
    int num_boxes;  // small scenes: the padded world boxes of the meshes that have a tree, so that the candidate test of a query
    int pad;        // reads them through scalar loads as well (box[i] belongs to record box[i].geom)
    Wall w[kMaxWalls];
    struct MeshBox {
        float mn[3];
        int geom;
        float mx[3];
        int pad;
    } box[32];
};

// Fills the table from the plane records [0, limit) (processing order; limit <= 32).  A plane qualifies when the three axis
// columns of its model matrix are parallel to three different world axes to within 2e-7 of their lengths (rotations by multiples
// of 90 degrees come out of glm's cos/sin that exact), its normal is the unit quad's (0, 0, 1) and its scales differ by at most a
// factor of 16.  At most kMaxWalls planes, the leading (largest) ones.
void build_wall_table(const struct GeomRecord* geoms, int limit, WallTable& out, bool pairing = true); // pairing: walls that share a rectangle share an entry
// ... and the mesh boxes of a small scene (records [first, n), n <= 32).
void add_mesh_boxes(const struct GeomRecord* geoms, int first, int n, WallTable& out);

struct CompiledScene {
    std::vector<GeomRecord> geoms;
    std::vector<TriRecord> tris;   // leaf order
    std::vector<TriNormals> normals; // parallel to tris
    std::vector<BvhNode> nodes;    // all meshes, each mesh's nodes contiguous in breadth-first order
    int max_depth = 0;             // deepest root-to-leaf path in inner nodes over all meshes
    uint64_t total_tris = 0;
};

struct BvhBuildParams {
    int max_leaf_tris = 2; // <= 8 (2 measured best on the benchmark scene: 8.1 node visits + 1.7 triangle tests per ray vs 7.9 + 1.9 with 4)
    int device_leaf_tris = 2; // device builders: subtrees of up to this many triangles become one leaf (<= max_leaf_tris)
    int max_depth = 30;    // hard bound on inner-node depth (the traversal stack is sized from the built depth)
    int bins = 64;
    float c_trav = 1.2f;   // SAH cost of an inner-node visit relative to one triangle test
    int opt_passes = 1;    // insertion-based optimisation of the finished tree: at most this many passes (0: none; one pass measured best, profiles/r04_m_*)
    int opt_max_tris = 1 << 18; // ... for meshes of up to this many triangles
};

// Defaults, overridable for experiments through FF_BVH_LEAF / FF_BVH_BINS / FF_BVH_CTRAV.
BvhBuildParams default_bvh_params();

// Build the object-space BVH of one mesh.  Appends inner nodes to `nodes` (breadth-first, root first) and the mesh's
// triangles, in leaf order, to `tris`.  Returns the root inner-node index (into `nodes`) and the tree depth.
int build_mesh_bvh(const FfTriangle* triangles, int count, const BvhBuildParams& params, std::vector<BvhNode>& nodes,
                   std::vector<TriRecord>& tris, int* out_depth, std::vector<TriNormals>* normals = nullptr);

// Flatten host geometries into device records.  Returns an FfStatus.  build_bvh = false fills the geometry records only
// (tri_first / tri_count assigned, bvh_root = -1, no triangle records, no nodes): the device builder's input.
int compile_scene(const FfGeometry* geoms, int n, const BvhBuildParams& params, CompiledScene& out, bool build_bvh = true);

// Binary tree over the padded world boxes of ALL geometry records (planes, spheres and meshes; records in processing order):
// what a query of a scene with more than 32 geometries walks instead of scanning every record (kernel.cu:133's loop).
// nodes[0] is the root; link >= 0: node index, link < 0: ~(record index).  Surface-area-heuristic splits
// (sweep over all three axes: large boxes such as a room's walls are peeled off near the root; median splits below binary
// depth 24, so the depth stays bounded), one geometry per leaf.  Needs at least two records.  Returns the depth (nodes on the longest root-to-leaf path).
int build_geometry_tree(const std::vector<GeomRecord>& geoms, std::vector<BvhNode>& nodes, int first); // over the records [first, n)
constexpr int kMaxScanPlanes = 8;
int count_scan_planes(const std::vector<GeomRecord>& geoms, int num_quads);

// The same tree in the 4-wide form the trace kernels traverse (gpu_collapse_mesh's rule: every binary node at even depth
// becomes a 4-wide node holding its grandchildren), level by level, links relative to node 0, a geometry as
// ~(kGeomLeafBit | record index).  Returns the depth of the 4-wide tree.
constexpr int32_t kGeomLeafBit = 0x40000000;
int collapse_geometry_tree(const std::vector<BvhNode>& binary, std::vector<Bvh4Node>& out);

// World-space AABB of a geometry record from object-space bounds (through the record's model matrix, padded).
void set_world_box(GeomRecord& r, const float omn[3], const float omx[3]);

} // namespace ff
