// ff_scene.cpp — the scene compiler: host Geometry[] (array of structs with owning pointers, utilities.h:219-233)
// -> flat device records (GeomRecord[], 48-byte TriRecord[] in leaf order, 64-byte BvhNode[] per mesh).
//
// This replaces the reference's deep-copy upload (kernel.cu:268-298), which ships the 208-byte Geometry and the
// 96-byte AoS Triangle verbatim.  Triangles stay in OBJECT space and keep their values bit-for-bit, because the
// reference intersects in object space (kernel.cu:138) and ranks hits by world distance (kernel.cu:113-121); the
// BVH therefore is one object-space tree per mesh and only prunes work, it never changes a computed hit.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <queue>
#include <system_error>
#include <thread>

#include "ff_internal.h"
#include "ff_math.h"

namespace ff {

namespace {

struct Box {
    float mn[3], mx[3];
    void reset()
    {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::numeric_limits<float>::infinity();
            mx[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* p)
    {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], p[k]);
            mx[k] = std::max(mx[k], p[k]);
        }
    }
    void grow(const Box& b)
    {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], b.mn[k]);
            mx[k] = std::max(mx[k], b.mx[k]);
        }
    }
    float half_area() const
    {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BuildNode {
    Box box;
    int left = -1, right = -1; // children (BuildNode indices), -1 for leaves
    int start = 0, count = 0;  // range in the permutation array
    int depth = 0;
};

struct Builder {
    const FfTriangle* tris;
    const BvhBuildParams& P;
    std::vector<Box> tbox;
    std::vector<float> cent; // 3 per triangle
    std::vector<int> perm;
    std::vector<BuildNode> bn;

    Builder(const FfTriangle* t, int n, const BvhBuildParams& p) : tris(t), P(p), tbox(n), cent(3 * (size_t)n), perm(n)
    {
        for (int i = 0; i < n; ++i) {
            Box b;
            b.reset();
            b.grow(&t[i].m_v0.x);
            b.grow(&t[i].m_v1.x);
            b.grow(&t[i].m_v2.x);
            tbox[i] = b;
            for (int k = 0; k < 3; ++k) cent[3 * (size_t)i + k] = 0.5f * (b.mn[k] + b.mx[k]);
        }
        std::iota(perm.begin(), perm.end(), 0);
    }

    Box range_box(int start, int count) const
    {
        Box b;
        b.reset();
        for (int i = 0; i < count; ++i) b.grow(tbox[perm[start + i]]);
        return b;
    }

    // Returns the split position (number of triangles going left) after partitioning perm[start, start+count), or 0 for "make a leaf".
    int split(int start, int count, const Box& box, int depth)
    {
        const int remaining = P.max_depth - depth;            // inner levels still allowed below this node
        const long cap_child = remaining >= 1 ? (long)P.max_leaf_tris << std::min(remaining - 1, 24) : 0;
        const bool must_split = count > P.max_leaf_tris;
        if (!must_split && count <= 1) return 0;

        Box cb;
        cb.reset();
        for (int i = 0; i < count; ++i) cb.grow(&cent[3 * (size_t)perm[start + i]]);

        const int NB = P.bins;
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        std::vector<Box> bbox(NB), rbox(NB);
        std::vector<int> bcnt(NB), lcnt(NB);
        for (int ax = 0; ax < 3; ++ax) {
            const float lo = cb.mn[ax], ext = cb.mx[ax] - cb.mn[ax];
            if (!(ext > 0.f)) continue;
            const float sc = (float)NB / ext;
            for (int b = 0; b < NB; ++b) { bbox[b].reset(); bcnt[b] = 0; }
            for (int i = 0; i < count; ++i) {
                const int t = perm[start + i];
                int b = (int)((cent[3 * (size_t)t + ax] - lo) * sc);
                b = std::max(0, std::min(NB - 1, b));
                bbox[b].grow(tbox[t]);
                ++bcnt[b];
            }
            Box acc;
            acc.reset();
            for (int b = NB - 1; b >= 0; --b) { acc.grow(bbox[b]); rbox[b] = acc; }
            acc.reset();
            int nl = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.grow(bbox[b]);
                nl += bcnt[b];
                const int nr = count - nl;
                if (nl == 0 || nr == 0) continue;
                const float cost = acc.half_area() * (float)nl + rbox[b + 1].half_area() * (float)nr;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
            }
        }

        int nleft = 0;
        if (best_axis >= 0) {
            const float parent_area = box.half_area();
            const float c_trav = P.c_trav, c_tri = 1.0f;
            const float split_cost = c_trav + (parent_area > 0.f ? best_cost / parent_area : (float)count) * c_tri;
            if (!must_split && split_cost >= (float)count * c_tri) return 0;
            const float lo = cb.mn[best_axis], sc = (float)NB / (cb.mx[best_axis] - cb.mn[best_axis]);
            auto mid = std::partition(perm.begin() + start, perm.begin() + start + count, [&](int t) {
                int b = (int)((cent[3 * (size_t)t + best_axis] - lo) * sc);
                b = std::max(0, std::min(NB - 1, b));
                return b <= best_bin;
            });
            nleft = (int)(mid - (perm.begin() + start));
        } else if (!must_split) {
            return 0;
        }
        // Depth guard / degenerate SAH: fall back to an object-median split on the widest centroid axis.
        const bool unbalanced = nleft == 0 || nleft == count || (long)std::max(nleft, count - nleft) > cap_child;
        if (unbalanced) {
            int ax = 0;
            float e = -1.f;
            for (int k = 0; k < 3; ++k) {
                const float ek = cb.mx[k] - cb.mn[k];
                if (ek > e) { e = ek; ax = k; }
            }
            nleft = count / 2;
            std::nth_element(perm.begin() + start, perm.begin() + start + nleft, perm.begin() + start + count, [&](int a, int b) {
                const float ca = cent[3 * (size_t)a + ax], cb2 = cent[3 * (size_t)b + ax];
                return ca < cb2 || (ca == cb2 && a < b);
            });
        }
        return nleft;
    }

    // Split node `root` of `v` and everything below it (children are appended to `v`).
    void expand(std::vector<BuildNode>& v, int root)
    {
        std::vector<int> todo{ root };
        while (!todo.empty()) {
            const int id = todo.back();
            todo.pop_back();
            if (split_node(v, id)) {
                todo.push_back(v[id].left);
                todo.push_back(v[id].right);
            }
        }
    }

    // One split: false if node `id` of `v` stays a leaf.
    bool split_node(std::vector<BuildNode>& v, int id)
    {
        const int start = v[id].start, count = v[id].count, depth = v[id].depth;
        const Box box = v[id].box;
        const int nleft = split(start, count, box, depth);
        if (nleft <= 0) return false;
        BuildNode l, r;
        l.start = start; l.count = nleft; l.depth = depth + 1; l.box = range_box(l.start, l.count);
        r.start = start + nleft; r.count = count - nleft; r.depth = depth + 1; r.box = range_box(r.start, r.count);
        const int li = (int)v.size();
        v.push_back(l);
        const int ri = (int)v.size();
        v.push_back(r);
        v[id].left = li;
        v[id].right = ri;
        return true;
    }

    // Large meshes are built by several threads: the biggest pending subtrees are split one by one until there are enough of
    // them, then each is finished by one thread in its own node list (their triangle ranges are disjoint), and the lists are
    // appended in a fixed order.  The tree does not depend on the number of threads: the emitter numbers nodes by their links.
    void build()
    {
        BuildNode root;
        root.start = 0;
        root.count = (int)perm.size();
        root.box = range_box(0, root.count);
        root.depth = 0;
        bn.push_back(root);
        int threads = 1;
        if (root.count >= 65536) threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (const char* e = std::getenv("FF_BVH_THREADS")) threads = std::max(1, std::min(64, std::atoi(e)));
        if (threads == 1) {
            expand(bn, 0);
            return;
        }
        std::vector<int> pending{ 0 };
        while (pending.size() < (size_t)threads * 8) {
            size_t big = 0;
            for (size_t q = 1; q < pending.size(); ++q)
                if (bn[pending[q]].count > bn[pending[big]].count) big = q;
            const int id = pending[big];
            if (bn[id].count < 2048) break;
            pending.erase(pending.begin() + (long)big);
            if (split_node(bn, id)) {
                pending.push_back(bn[id].left);
                pending.push_back(bn[id].right);
            }
        }
        std::sort(pending.begin(), pending.end());
        std::vector<std::vector<BuildNode>> subs(pending.size());
        std::atomic<size_t> next{ 0 };
        auto work = [&]() {
            for (size_t q; (q = next.fetch_add(1)) < pending.size();) {
                subs[q].push_back(bn[pending[q]]);
                expand(subs[q], 0);
            }
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < threads; ++w) {
            try {
                pool.emplace_back(work);
            } catch (const std::system_error&) { // (no more threads to be had: the ones that started, and this one, do the work)
                break;
            }
        }
        work();
        for (std::thread& th : pool) th.join();
        for (size_t q = 0; q < pending.size(); ++q) {
            const std::vector<BuildNode>& sub = subs[q];
            const int base = (int)bn.size() - 1; // sub[k], k >= 1, becomes bn[base + k]
            if (sub[0].left >= 0) {
                bn[pending[q]].left = base + sub[0].left;
                bn[pending[q]].right = base + sub[0].right;
            }
            for (size_t k = 1; k < sub.size(); ++k) {
                BuildNode n = sub[k];
                if (n.left >= 0) {
                    n.left += base;
                    n.right += base;
                }
                bn.push_back(n);
            }
        }
    }

    // ---- insertion-based optimisation of the finished tree (Bittner, Hapala & Havran 2013) ----------------------------------
    //
    // The top-down build decides every split with what it knows at that node; afterwards the tree is improved by taking inner
    // nodes out - the node and its parent disappear, the sibling moves up - and putting the node's two subtrees back where they
    // add the least surface area to the tree (the area of the new common parent plus what the boxes on the way down to it grow
    // by), found by a branch-and-bound search from the root.  Passes over all inner nodes, largest first, until a pass gains less
    // than 0.2 %; the best tree seen that respects max_depth is kept.  Leaves (their triangle ranges) are never touched, so the
    // triangle records, the leaf sizes and every result stay what they were: the tree only prunes.
    float inner_area_sum() const
    {
        double a = 0.0;
        for (const BuildNode& n : bn)
            if (n.left >= 0) a += n.box.half_area();
        return (float)a;
    }

    int assign_depths()
    {
        int deepest = 0;
        std::vector<int> todo{ 0 };
        bn[0].depth = 0;
        while (!todo.empty()) {
            const int i = todo.back();
            todo.pop_back();
            if (bn[i].left < 0) continue;
            deepest = std::max(deepest, bn[i].depth + 1);
            for (int c : { bn[i].left, bn[i].right }) {
                bn[c].depth = bn[i].depth + 1;
                todo.push_back(c);
            }
        }
        return deepest;
    }

    void optimise(int max_passes)
    {
        const int n = (int)bn.size();
        if (max_passes <= 0 || n < 15) return;
        std::vector<int> parent(n, -1);
        for (int i = 0; i < n; ++i)
            if (bn[i].left >= 0) parent[bn[i].left] = parent[bn[i].right] = i;
        auto refit_up = [&](int i) {
            for (; i >= 0; i = parent[i]) {
                Box b = bn[bn[i].left].box;
                b.grow(bn[bn[i].right].box);
                bn[i].box = b;
            }
        };
        auto united = [&](const Box& a, const Box& b) {
            Box u = a;
            u.grow(b);
            return u.half_area();
        };
        struct Cand {
            float induced;
            int node;
            bool operator<(const Cand& o) const { return induced > o.induced; } // (std::priority_queue pops the largest)
        };
        // where subtree x adds the least area (never the root's own place: node 0 stays the root)
        auto best_place = [&](int x) {
            const Box& xb = bn[x].box;
            const float xa = xb.half_area();
            float best = std::numeric_limits<float>::infinity();
            int where = -1;
            std::priority_queue<Cand> q;
            const float at_root = united(bn[0].box, xb) - bn[0].box.half_area();
            q.push({ at_root, bn[0].left });
            q.push({ at_root, bn[0].right });
            // (a search that cannot prune - thousands of boxes on top of each other - settles for the best of its first 4 096
            // candidates: any place is a valid place, and the pass stays linear in the size of the tree)
            for (int visited = 0; !q.empty() && visited < 4096; ++visited) {
                const Cand c = q.top();
                q.pop();
                if (c.induced + xa >= best) break;
                const float direct = united(bn[c.node].box, xb);
                if (c.induced + direct < best) {
                    best = c.induced + direct;
                    where = c.node;
                }
                if (bn[c.node].left >= 0) {
                    const float below = c.induced + (direct - bn[c.node].box.half_area());
                    if (below + xa < best) {
                        q.push({ below, bn[c.node].left });
                        q.push({ below, bn[c.node].right });
                    }
                }
            }
            return where >= 0 ? where : bn[0].left; // (boxes that are not numbers compare false with everything: any place is a valid place)
        };
        auto replace_child = [&](int p, int from, int to) {
            if (bn[p].left == from) bn[p].left = to;
            else bn[p].right = to;
            parent[to] = p;
        };
        std::vector<BuildNode> best_tree = bn;
        float best_cost = inner_area_sum();
        const float first_cost = best_cost;
        std::vector<int> order;
        int stale = 0;
        unsigned long long rng = 0x9E3779B97F4A7C15ull;
        for (int pass = 0; pass < max_passes; ++pass) {
            order.clear();
            for (int i = 1; i < n; ++i)
                if (bn[i].left >= 0 && parent[i] > 0) order.push_back(i); // (an inner node with a grandparent)
            if (pass % 2 == 0) {
                std::sort(order.begin(), order.end(), [&](int a, int b) {
                    const float aa = bn[a].box.half_area(), ab = bn[b].box.half_area();
                    return aa > ab || (aa == ab && a < b);
                });
            } else {
                // (every other pass in a scrambled order - a fixed sequence, the tree must not depend on the run: moves that the
                // largest-first order never tries)
                for (size_t i = order.size(); i > 1; --i) {
                    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                    std::swap(order[i - 1], order[(size_t)((rng >> 33) % i)]);
                }
            }
            for (int nd : order) {
                const int p = parent[nd];
                if (bn[nd].left < 0 || p <= 0) continue; // (an earlier move of this pass made it a child of the root)
                const int g = parent[p];
                const int sib = bn[p].left == nd ? bn[p].right : bn[p].left;
                replace_child(g, p, sib);
                refit_up(g);
                int sub[2] = { bn[nd].left, bn[nd].right };
                if (bn[sub[0]].box.half_area() < bn[sub[1]].box.half_area()) std::swap(sub[0], sub[1]);
                const int spare[2] = { nd, p };
                for (int k = 0; k < 2; ++k) {
                    const int x = sub[k], f = spare[k];
                    const int at = best_place(x);
                    const int ap = parent[at];
                    replace_child(ap, at, f);
                    bn[f].left = at;
                    bn[f].right = x;
                    parent[at] = parent[x] = f;
                    refit_up(f);
                }
            }
            const float cost = inner_area_sum();
            const bool fits = assign_depths() <= P.max_depth;
            if (fits && cost < best_cost) {
                stale = cost > best_cost * 0.999f ? stale + 1 : 0;
                best_cost = cost;
                best_tree = bn;
            } else {
                ++stale;
            }
            if (stale >= 3) break;
        }
        if (std::getenv("FF_BVH_OPT_VERBOSE")) std::fprintf(stderr, "[ff] tree of %d nodes: inner area %.6g -> %.6g\n", n, (double)first_cost, (double)best_cost);
        bn = best_tree;
        assign_depths();
        // triangles back in leaf order (depth first): neighbouring leaves keep neighbouring records
        std::vector<int> new_perm;
        new_perm.reserve(perm.size());
        std::vector<int> todo{ 0 };
        while (!todo.empty()) {
            const int i = todo.back();
            todo.pop_back();
            if (bn[i].left < 0) {
                const int start = (int)new_perm.size();
                for (int k = 0; k < bn[i].count; ++k) new_perm.push_back(perm[bn[i].start + k]);
                bn[i].start = start;
            } else {
                todo.push_back(bn[i].right);
                todo.push_back(bn[i].left);
            }
        }
        perm.swap(new_perm);
    }
};

} // namespace

BvhBuildParams default_bvh_params()
{
    BvhBuildParams p;
    if (const char* e = std::getenv("FF_BVH_LEAF")) p.max_leaf_tris = std::max(1, std::min(8, std::atoi(e)));
    if (const char* e = std::getenv("FF_BVH_BINS")) p.bins = std::max(4, std::min(256, std::atoi(e)));
    if (const char* e = std::getenv("FF_BVH_CTRAV")) p.c_trav = (float)std::atof(e);
    if (const char* e = std::getenv("FF_BVH_OPT_PASSES")) p.opt_passes = std::max(0, std::min(1000, std::atoi(e)));
    return p;
}

int build_mesh_bvh(const FfTriangle* triangles, int count, const BvhBuildParams& params, std::vector<BvhNode>& nodes,
                   std::vector<TriRecord>& tris, int* out_depth, std::vector<TriNormals>* normals)
{
    if (count <= 0) {
        if (out_depth) *out_depth = 0;
        return -1;
    }
    Builder B(triangles, count, params);
    B.build();
    if (count <= params.opt_max_tris) B.optimise(params.opt_passes);

    // Conservative padding: the triangle test accepts hits whose geometric miss distance is a few ulps of the
    // coordinates involved; boxes are grown by 1e-4 of the mesh's largest |coordinate| (>> those ulps) so that every
    // triangle hit the brute-force loop would report lies inside all boxes on its root path.
    float big = 0.f;
    for (int k = 0; k < 3; ++k) big = std::max(big, std::max(std::fabs(B.bn[0].box.mn[k]), std::fabs(B.bn[0].box.mx[k])));
    const float pad = 1e-4f * std::max(big, 1e-3f);

    const int tri_base = (int)tris.size();
    for (int i = 0; i < count; ++i) {
        const FfTriangle& t = triangles[B.perm[i]];
        TriRecord r;
        std::memset(&r, 0, sizeof r);
        std::memcpy(r.v0, &t.m_v0, 12);
        // Edges exactly as the reference forms them per test (kernel.cu:44-45: one fp32 subtraction per component).
        const float e1[3] = { t.m_v1.x - t.m_v0.x, t.m_v1.y - t.m_v0.y, t.m_v1.z - t.m_v0.z };
        const float e2[3] = { t.m_v2.x - t.m_v0.x, t.m_v2.y - t.m_v0.y, t.m_v2.z - t.m_v0.z };
        std::memcpy(r.e1, e1, 12);
        std::memcpy(r.e2, e2, 12);
        // kernel.cu:49 rejects when dot(d, cross(e1, e2)) > 0 and kernel.cu:57 when det = dot(e1, cross(d, e2)) < EPSILON.
        // In exact arithmetic det = -dot(d, cross(e1, e2)); each computed value is off by at most 6 roundings of terms
        // whose magnitudes sum to |d|_max * |e1|_1 * |e2|_1 (|d|_max <= 1 + 3 ulp: the direction is normalised).  So when
        // det >= EPSILON + 2 * 8 eps * |e1|_1 |e2|_1 the back-face test is certain to pass and the kernel skips it; the
        // margin stored here is twice that again, rounded up.
        const double m = ((double)std::fabs(e1[0]) + std::fabs(e1[1]) + std::fabs(e1[2])) * ((double)std::fabs(e2[0]) + std::fabs(e2[1]) + std::fabs(e2[2]));
        r.cull_margin = std::nextafter((float)(32.0 * 5.9604644775390625e-8 * m), 3.0e38f);
        r.orig_index = B.perm[i];
        tris.push_back(r);
        if (normals) {
            TriNormals nr;
            std::memset(&nr, 0, sizeof nr);
            std::memcpy(nr.n0, &t.m_n0, 12);
            std::memcpy(nr.n1, &t.m_n1, 12);
            std::memcpy(nr.n2, &t.m_n2, 12);
            normals->push_back(nr);
        }
    }

    // Breadth-first numbering of inner nodes; a mesh that fits one leaf still gets an inner root with an empty right child.
    const int node_base = (int)nodes.size();
    auto leaf_link = [&](const BuildNode& n) { return ~(((tri_base + n.start) << 3) | (n.count - 1)); };
    auto put_box = [&](const BuildNode& n, float* mn, float* mx) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = n.box.mn[k] - pad;
            mx[k] = n.box.mx[k] + pad;
        }
    };
    int depth = 1;
    if (B.bn[0].left < 0) {
        // Single-leaf mesh: both links point at the same leaf (testing a triangle twice cannot change the closest hit).
        BvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        put_box(B.bn[0], nd.lmin, nd.lmax);
        put_box(B.bn[0], nd.rmin, nd.rmax);
        nd.left = leaf_link(B.bn[0]);
        nd.right = nd.left;
        nodes.push_back(nd);
    } else {
        std::vector<int> order; // build-node ids of inner nodes in BFS order
        std::vector<int> bfs_index(B.bn.size(), -1);
        order.push_back(0);
        bfs_index[0] = 0;
        for (size_t q = 0; q < order.size(); ++q) {
            const BuildNode& n = B.bn[order[q]];
            for (int child : { n.left, n.right }) {
                if (B.bn[child].left >= 0) {
                    bfs_index[child] = (int)order.size();
                    order.push_back(child);
                }
            }
        }
        for (size_t q = 0; q < order.size(); ++q) {
            const BuildNode& n = B.bn[order[q]];
            const BuildNode& l = B.bn[n.left];
            const BuildNode& r = B.bn[n.right];
            BvhNode nd;
            std::memset(&nd, 0, sizeof nd);
            put_box(l, nd.lmin, nd.lmax);
            put_box(r, nd.rmin, nd.rmax);
            nd.left = l.left >= 0 ? node_base + bfs_index[n.left] : leaf_link(l);
            nd.right = r.left >= 0 ? node_base + bfs_index[n.right] : leaf_link(r);
            nodes.push_back(nd);
            depth = std::max(depth, n.depth + 1);
        }
    }
    if (out_depth) *out_depth = depth;
    return node_base;
}

namespace {

struct TopItem {
    int record;
    float mn[3], mx[3], centre[3];
};

// Builds the subtree over items [first, last) and returns its link; inner nodes are appended to `nodes`.
int build_top(std::vector<TopItem>& items, int first, int last, std::vector<BvhNode>& nodes, int depth, int* max_depth, float* out_mn, float* out_mx)
{
    if (last - first == 1) {
        for (int k = 0; k < 3; ++k) { out_mn[k] = items[first].mn[k]; out_mx[k] = items[first].mx[k]; }
        return ~items[first].record;
    }
    // Surface-area heuristic over all three axes (sweep of the boxes sorted by centre): the walls of a room are huge flat
    // boxes that overlap everything, and a median split would drag them through every subtree; the SAH peels them off near the
    // root, so the small objects below get tight boxes a ray can skip.
    const int n = last - first;
    auto half_area = [](const float* mn, const float* mx) {
        const double dx = std::max(0.0, (double)std::min(mx[0], 1.0e30f) - (double)std::max(mn[0], -1.0e30f));
        const double dy = std::max(0.0, (double)std::min(mx[1], 1.0e30f) - (double)std::max(mn[1], -1.0e30f));
        const double dz = std::max(0.0, (double)std::min(mx[2], 1.0e30f) - (double)std::max(mn[2], -1.0e30f));
        return dx * dy + dy * dz + dz * dx;
    };
    int best_axis = 0, best_split = first + n / 2;
    double best_cost = std::numeric_limits<double>::infinity();
    std::vector<double> right_area((size_t)n);
    for (int axis = 0; axis < 3; ++axis) {
        std::sort(items.begin() + first, items.begin() + last,
                  [axis](const TopItem& a, const TopItem& b) { return a.centre[axis] < b.centre[axis] || (a.centre[axis] == b.centre[axis] && a.record < b.record); });
        float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        for (int i = last - 1; i > first; --i) {
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], items[i].mn[k]); mx[k] = std::max(mx[k], items[i].mx[k]); }
            right_area[(size_t)(i - first)] = half_area(mn, mx);
        }
        for (int k = 0; k < 3; ++k) { mn[k] = 3.0e38f; mx[k] = -3.0e38f; }
        for (int i = first; i < last - 1; ++i) {
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], items[i].mn[k]); mx[k] = std::max(mx[k], items[i].mx[k]); }
            const double cost = half_area(mn, mx) * (double)(i + 1 - first) + right_area[(size_t)(i + 1 - first)] * (double)(last - i - 1);
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = axis;
                best_split = i + 1;
            }
        }
    }
    if (depth >= 24) { // (a pathological arrangement must not grow a chain: the traversal stack is sized from the depth)
        best_split = first + n / 2;
    }
    const int axis = best_axis, mid = best_split;
    std::sort(items.begin() + first, items.begin() + last,
              [axis](const TopItem& a, const TopItem& b) { return a.centre[axis] < b.centre[axis] || (a.centre[axis] == b.centre[axis] && a.record < b.record); });
    const int index = (int)nodes.size();
    nodes.push_back(BvhNode());
    *max_depth = std::max(*max_depth, depth + 1);
    float lmn[3], lmx[3], rmn[3], rmx[3];
    const int left = build_top(items, first, mid, nodes, depth + 1, max_depth, lmn, lmx);
    const int right = build_top(items, mid, last, nodes, depth + 1, max_depth, rmn, rmx);
    BvhNode& nd = nodes[index];
    for (int k = 0; k < 3; ++k) {
        nd.lmin[k] = lmn[k]; nd.lmax[k] = lmx[k];
        nd.rmin[k] = rmn[k]; nd.rmax[k] = rmx[k];
        out_mn[k] = std::min(lmn[k], rmn[k]);
        out_mx[k] = std::max(lmx[k], rmx[k]);
    }
    nd.left = left;
    nd.right = right;
    nd.pad0 = nd.pad1 = 0;
    return index;
}

} // namespace

static double box_half_area(const float* mn, const float* mx);

int build_geometry_tree(const std::vector<GeomRecord>& geoms, std::vector<BvhNode>& nodes, int first)
{
    nodes.clear();
    std::vector<TopItem> items(geoms.size() - (size_t)first);
    for (size_t i = (size_t)first; i < geoms.size(); ++i) {
        TopItem& it = items[i - (size_t)first];
        it.record = (int)i;
        for (int k = 0; k < 3; ++k) {
            it.mn[k] = geoms[i].wmin[k];
            it.mx[k] = geoms[i].wmax[k];
            // (a geometry nothing can hit carries an inverted box at 3e38: its centre only has to be finite)
            it.centre[k] = 0.5f * std::min(it.mn[k], 1.0e30f) + 0.5f * std::max(std::min(it.mx[k], 1.0e30f), -1.0e30f);
        }
    }
    int depth = 0;
    float mn[3], mx[3];
    nodes.reserve(geoms.size());
    build_top(items, 0, (int)items.size(), nodes, 0, &depth, mn, mx);
    return depth;
}

static double box_half_area(const float* mn, const float* mx)
{
    const double dx = std::max(0.0, (double)std::min(mx[0], 1.0e30f) - (double)std::max(mn[0], -1.0e30f));
    const double dy = std::max(0.0, (double)std::min(mx[1], 1.0e30f) - (double)std::max(mn[1], -1.0e30f));
    const double dz = std::max(0.0, (double)std::min(mx[2], 1.0e30f) - (double)std::max(mn[2], -1.0e30f));
    return dx * dy + dy * dz + dz * dx;
}

// Scenes that walk the geometry tree: the planes whose world boxes span a good part of the scene (the walls of a room) stay
// OUT of the tree and are screened first by every query, like the planes of a small scene: their boxes would overlap every
// subtree, and the wall a ray ends on bounds the walk through everything else from its first node.  The records hold the
// planes largest first (compile_scene), so these are the leading records: returns how many (at most kMaxScanPlanes, and
// the tree keeps at least two geometries).
int count_scan_planes(const std::vector<GeomRecord>& geoms, int num_quads)
{
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (const GeomRecord& r : geoms) {
        if (!(r.wmin[0] <= r.wmax[0])) continue;
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], r.wmin[k]); mx[k] = std::max(mx[k], r.wmax[k]); }
    }
    const double scene = box_half_area(mn, mx);
    int n = 0;
    while (n < num_quads && n < kMaxScanPlanes && n + 2 < (int)geoms.size() && scene > 0.0 && box_half_area(geoms[n].wmin, geoms[n].wmax) >= 0.08 * scene) ++n;
    return n;
}

void build_wall_table(const GeomRecord* geoms, int limit, WallTable& out, bool pairing)
{
    std::memset(&out, 0, sizeof out);
    struct Found {
        int axis;
        Wall w;
    };
    std::vector<Found> found;
    float scale_s = 0.f, ratio = 1.f;
    for (int g = 0; g < limit && g < 32 && (int)found.size() < kMaxWalls; ++g) {
        const GeomRecord& r = geoms[g];
        if (r.type != FF_GEOM_PLANE) continue;
        if (!(r.plane_n[0] == 0.f && r.plane_n[1] == 0.f && r.plane_n[2] == 1.f)) continue; // utilities.h:229's unit quad only
        const float* col[3] = { r.mod_c0, r.mod_c1, r.mod_c2 };
        int dom[3];
        float len[3];
        bool ok = true;
        for (int j = 0; j < 3 && ok; ++j) {
            const float a0 = std::fabs(col[j][0]), a1 = std::fabs(col[j][1]), a2 = std::fabs(col[j][2]);
            dom[j] = a0 >= a1 && a0 >= a2 ? 0 : (a1 >= a2 ? 1 : 2);
            len[j] = std::sqrt(a0 * a0 + a1 * a1 + a2 * a2);
            if (!(len[j] > 1e-20f && len[j] < 1e20f)) ok = false;
            for (int k = 0; k < 3 && ok; ++k)
                if (k != dom[j] && std::fabs(col[j][k]) > 2.0e-7f * len[j]) ok = false;
        }
        if (!ok || dom[0] == dom[1] || dom[0] == dom[2] || dom[1] == dom[2]) continue;
        const float smax = std::max(len[0], std::max(len[1], len[2])), smin = std::min(len[0], std::min(len[1], len[2]));
        if (!(smax <= 16.f * smin)) continue;
        const float* T = r.mod_c3;
        if (!(std::fabs(T[0]) < 1e8f && std::fabs(T[1]) < 1e8f && std::fabs(T[2]) < 1e8f)) continue;
        Found f;
        f.axis = dom[2]; // the world axis the quad's normal (object z) points along
        const int u = (f.axis + 1) % 3, v = (f.axis + 2) % 3;
        // half extents along the other two world axes: half the length of the column (object x or y) that points along each
        const float hu = 0.5f * (dom[0] == u ? len[0] : len[1]), hv = 0.5f * (dom[0] == v ? len[0] : len[1]);
        std::memset(&f.w, 0, sizeof f.w);
        f.w.c = T[f.axis];
        f.w.cu = T[u];
        f.w.hu = hu;
        f.w.cv = T[v];
        f.w.hv = hv;
        f.w.geom = g;
        found.push_back(f);
        ratio = std::max(ratio, smax / smin);
        scale_s = std::max(scale_s, smax + std::fabs(T[0]) + std::fabs(T[1]) + std::fabs(T[2]));
    }
    // Two walls normal to the same axis with the same rectangle (floor and ceiling, left and right wall of a box) share one entry:
    // a ray between them can reach only the one its direction points at, so the kernel computes both parameters, tests the
    // rectangle once and hands the other wall to the per-lane screens unless its parameter is certainly negative (wall_test_pair).
    std::vector<char> used(found.size(), 0);
    int n = 0;
    for (int axis = 0; axis < 3; ++axis)
        for (size_t i = 0; i < found.size(); ++i) {
            if (found[i].axis != axis || used[i]) continue;
            used[i] = 1;
            Wall w = found[i].w;
            out.mask |= 1u << w.geom;
            for (size_t j = i + 1; pairing && j < found.size(); ++j) {
                const Wall& o = found[j].w;
                if (found[j].axis != axis || used[j] || o.c == w.c) continue;
                if (std::memcmp(&o.cu, &w.cu, 4 * sizeof(float)) != 0) continue; // cu, hu, cv, hv bit for bit
                used[j] = 1;
                out.mask |= 1u << o.geom;
                const Wall& lo = o.c < w.c ? o : w;
                const Wall& hi = o.c < w.c ? w : o;
                Wall both = lo;
                both.hi_geom1 = hi.geom + 1;
                both.hi_c = hi.c;
                w = both;
                break;
            }
            out.w[n++] = w;
            out.count[axis] += 1;
        }
    // margins: both scale with the anisotropy of the walls (a rounding error of the object-space test along a short axis is that
    // much larger in world units along a long one)
    out.margin_s = ratio * scale_s;
    out.graze = 1.2e-7f * ratio; // kernel.cu:12: |n.d'| > 1e-7 for the normalised object-space direction d'
    if (ratio > 1.f) {
        // D = 2e-5 (|o| + S) must cover ratio * (|o| + ...): fold the factor on |o| into S for the scene's extent instead of a
        // per-ray multiply: origins of interest lie within the scene, |o|_1 <= 3 * scale_s
        out.margin_s += (ratio - 1.f) * 3.f * scale_s;
    }
}

void add_mesh_boxes(const GeomRecord* geoms, int first, int n, WallTable& out)
{
    out.num_boxes = 0;
    for (int g = first; g < n && g < 32; ++g) {
        const GeomRecord& r = geoms[g];
        if (r.type != FF_GEOM_TRIANGLEMESH || r.bvh_root < 0) continue;
        WallTable::MeshBox& b = out.box[out.num_boxes++];
        for (int k = 0; k < 3; ++k) { b.mn[k] = r.wmin[k]; b.mx[k] = r.wmax[k]; }
        b.geom = g;
        b.pad = 0;
    }
}

int collapse_geometry_tree(const std::vector<BvhNode>& binary, std::vector<Bvh4Node>& out)
{
    out.clear();
    if (binary.empty()) return 0;
    // breadth-first over the binary nodes that become 4-wide nodes (even depth): a node's index in `order` is its 4-wide index
    std::vector<int> order(1, 0), index4(binary.size(), -1), level(1, 1);
    index4[0] = 0;
    auto grandchildren = [&](int n, int links[4], const float* mn[4], const float* mx[4]) {
        int slots = 0;
        const BvhNode& nd = binary[n];
        for (int side = 0; side < 2; ++side) {
            const int link = side == 0 ? nd.left : nd.right;
            if (link < 0) {
                links[slots] = link; mn[slots] = side == 0 ? nd.lmin : nd.rmin; mx[slots] = side == 0 ? nd.lmax : nd.rmax; ++slots;
            } else {
                const BvhNode& ch = binary[link];
                links[slots] = ch.left; mn[slots] = ch.lmin; mx[slots] = ch.lmax; ++slots;
                links[slots] = ch.right; mn[slots] = ch.rmin; mx[slots] = ch.rmax; ++slots;
            }
        }
        return slots;
    };
    int depth = 1;
    for (size_t i = 0; i < order.size(); ++i) {
        int links[4];
        const float *mn[4], *mx[4];
        const int slots = grandchildren(order[i], links, mn, mx);
        for (int q = 0; q < slots; ++q)
            if (links[q] >= 0) {
                index4[links[q]] = (int)order.size();
                order.push_back(links[q]);
                level.push_back(level[i] + 1);
                depth = std::max(depth, level[i] + 1);
            }
    }
    out.resize(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        int links[4];
        const float *mn[4], *mx[4];
        const int slots = grandchildren(order[i], links, mn, mx);
        Bvh4Node& o = out[i];
        for (int q = 0; q < 4; ++q) {
            for (int k = 0; k < 3; ++k) {
                o.mn[k][q] = q < slots ? mn[q][k] : std::numeric_limits<float>::infinity();
                o.mx[k][q] = q < slots ? mx[q][k] : -std::numeric_limits<float>::infinity();
            }
            o.link[q] = q >= slots ? kEmptyLink : (links[q] >= 0 ? index4[links[q]] : ~(kGeomLeafBit | ~links[q]));
        }
    }
    return depth;
}

// World-space AABB of a geometry from its object-space bounds (pruning only): the eight corners through the model matrix,
// padded.  An empty object box (omn > omx) gives a box no ray can enter.
void set_world_box(GeomRecord& r, const float omn[3], const float omx[3])
{
    using namespace ffm;
    M4 mod;
    mod.c[0] = v4(r.mod_c0[0], r.mod_c0[1], r.mod_c0[2], 0.f);
    mod.c[1] = v4(r.mod_c1[0], r.mod_c1[1], r.mod_c1[2], 0.f);
    mod.c[2] = v4(r.mod_c2[0], r.mod_c2[1], r.mod_c2[2], 0.f);
    mod.c[3] = v4(r.mod_c3[0], r.mod_c3[1], r.mod_c3[2], 1.f);
    Box wb;
    wb.reset();
    if (omn[0] <= omx[0]) {
        for (int c = 0; c < 8; ++c) {
            const V4 pw = mul(mod, v4((c & 1) ? omx[0] : omn[0], (c & 2) ? omx[1] : omn[1], (c & 4) ? omx[2] : omn[2], 1.f));
            wb.grow(&pw.x);
        }
        float big = 0.f;
        for (int k = 0; k < 3; ++k) big = std::max(big, std::max(std::fabs(wb.mn[k]), std::fabs(wb.mx[k])) + (wb.mx[k] - wb.mn[k]));
        const float wpad = 1e-4f * big + 1e-4f;
        for (int k = 0; k < 3; ++k) { r.wmin[k] = wb.mn[k] - wpad; r.wmax[k] = wb.mx[k] + wpad; }
    } else {
        for (int k = 0; k < 3; ++k) { r.wmin[k] = 3.0e38f; r.wmax[k] = 3.0e38f; }
    }
}

int compile_scene(const FfGeometry* geoms, int n, const BvhBuildParams& params, CompiledScene& out, bool build_bvh)
{
    using namespace ffm;
    out = CompiledScene();
    if (!geoms || n <= 0) return fail(FF_ERR_INVALID_ARG, "ff_upload_scene: no geometries");
    if (params.max_leaf_tris < 1 || params.max_leaf_tris > 8) return fail(FF_ERR_INVALID_ARG, "max_leaf_tris must be 1..8");
    out.geoms.resize(n);
    for (int i = 0; i < n; ++i) {
        const FfGeometry& g = geoms[i];
        GeomRecord& r = out.geoms[i];
        std::memset(&r, 0, sizeof r);
        if (g.m_geometryType != FF_GEOM_PLANE && g.m_geometryType != FF_GEOM_TRIANGLEMESH && g.m_geometryType != FF_GEOM_SPHERE)
            return fail(FF_ERR_UNSUPPORTED, "geometry %d: type %d is not a geometry type (kernel.cu:170-173)", i, g.m_geometryType);
        if (g.m_geometryType == FF_GEOM_SPHERE && !(g.m_sphereRadius > 0.f))
            return fail(FF_ERR_INVALID_ARG, "geometry %d: sphere radius %g", i, (double)g.m_sphereRadius);
        if (!g.m_bxdf) return fail(FF_ERR_INVALID_ARG, "geometry %d: m_bxdf is null (the reference copies it unconditionally, kernel.cu:282)", i);
        const M4 inv = load(g.m_inverseModelMatrix.m), mod = load(g.m_modelMatrix.m);
        // The kernel drops the w row of the object-space transform (kernel.cu:138 normalises a vec4 whose w is 0 for
        // affine matrices); refuse matrices for which that is not exact.
        if (inv.c[0].w != 0.f || inv.c[1].w != 0.f || inv.c[2].w != 0.f || mod.c[0].w != 0.f || mod.c[1].w != 0.f || mod.c[2].w != 0.f)
            return fail(FF_ERR_UNSUPPORTED, "geometry %d: model matrix is not affine", i);
        const M4 nrm = inverse(transpose(mod)); // kernel.cu:117, hoisted out of the per-hit path
        std::memcpy(r.inv_c0, &inv.c[0], 16); std::memcpy(r.inv_c1, &inv.c[1], 16);
        std::memcpy(r.inv_c2, &inv.c[2], 16); std::memcpy(r.inv_c3, &inv.c[3], 16);
        std::memcpy(r.mod_c0, &mod.c[0], 16); std::memcpy(r.mod_c1, &mod.c[1], 16);
        std::memcpy(r.mod_c2, &mod.c[2], 16); std::memcpy(r.mod_c3, &mod.c[3], 16);
        std::memcpy(r.nrm_c0, &nrm.c[0], 16); std::memcpy(r.nrm_c1, &nrm.c[1], 16); std::memcpy(r.nrm_c2, &nrm.c[2], 16);
        // glm's mat4*vec4 adds column3*w even when w == 0; keep that signed zero so directions/normals match bit-for-bit.
        r.inv_c0[3] = inv.c[3].x * 0.0f; r.inv_c1[3] = inv.c[3].y * 0.0f; r.inv_c2[3] = inv.c[3].z * 0.0f;
        r.nrm_c0[3] = nrm.c[3].x * 0.0f; r.nrm_c1[3] = nrm.c[3].y * 0.0f; r.nrm_c2[3] = nrm.c[3].z * 0.0f;
        r.plane_n[0] = g.m_normal.x; r.plane_n[1] = g.m_normal.y; r.plane_n[2] = g.m_normal.z;
        r.plane_n[3] = g.m_geometryType == FF_GEOM_SPHERE ? g.m_sphereRadius : 0.f; // w slot: sphere radius
        const FfBXDF& b = *g.m_bxdf;
        // throughput factor of a bounce off this surface: m_specularColor for MIRROR, m_albedo otherwise
        const FfVec3& tint = b.m_type == FF_BXDF_MIRROR ? b.m_specularColor : b.m_albedo;
        r.albedo[0] = tint.x; r.albedo[1] = tint.y; r.albedo[2] = tint.z;
        r.emission[0] = b.m_emissiveColor.x * b.m_intensity; // utilities.h:102
        r.emission[1] = b.m_emissiveColor.y * b.m_intensity;
        r.emission[2] = b.m_emissiveColor.z * b.m_intensity;
        if (b.m_type == FF_BXDF_GLASS) {
            // a dielectric neither emits nor has an albedo: the two colour slots carry m_specularColor (reflection) and
            // m_transmittanceColor (refraction), the spare lane of the first one the refractive index
            if (!(b.m_refractiveIndex > 0.f)) return fail(FF_ERR_INVALID_ARG, "geometry %d: glass needs a positive m_refractiveIndex", i);
            r.albedo[0] = b.m_specularColor.x; r.albedo[1] = b.m_specularColor.y; r.albedo[2] = b.m_specularColor.z;
            r.albedo[3] = b.m_refractiveIndex;
            r.emission[0] = b.m_transmittanceColor.x; r.emission[1] = b.m_transmittanceColor.y; r.emission[2] = b.m_transmittanceColor.z;
        }
        r.type = g.m_geometryType;
        r.bxdf_type = b.m_type;
        r.bvh_root = -1;
        r.orig_index = i;
        // object-space bounds -> world AABB (pruning only, padded below)
        float omn[3] = { -0.5f, -0.5f, 0.f }, omx[3] = { 0.5f, 0.5f, 0.f }; // unit plane quad, kernel.cu:18
        if (g.m_geometryType == FF_GEOM_SPHERE)
            for (int k = 0; k < 3; ++k) { omn[k] = -g.m_sphereRadius; omx[k] = g.m_sphereRadius; }
        if (g.m_geometryType == FF_GEOM_TRIANGLEMESH) {
            Box ob;
            ob.reset();
            const int cnt0 = g.m_triangles ? g.m_numberOfTriangles : 0;
            for (int t = 0; t < cnt0; ++t) {
                ob.grow(&g.m_triangles[t].m_v0.x);
                ob.grow(&g.m_triangles[t].m_v1.x);
                ob.grow(&g.m_triangles[t].m_v2.x);
            }
            for (int k = 0; k < 3; ++k) { omn[k] = ob.mn[k]; omx[k] = ob.mx[k]; }
        }
        set_world_box(r, omn, omx);
        if (g.m_geometryType == FF_GEOM_TRIANGLEMESH) {
            const int cnt = g.m_triangles ? g.m_numberOfTriangles : 0;
            if (cnt < 0) return fail(FF_ERR_INVALID_ARG, "geometry %d: negative triangle count", i);
            if ((uint64_t)out.tris.size() + (uint64_t)cnt >= (1ull << 28))
                return fail(FF_ERR_UNSUPPORTED, "scene exceeds 2^28 triangles");
            if ((uint64_t)out.total_tris + (uint64_t)cnt >= (1ull << 28)) return fail(FF_ERR_UNSUPPORTED, "scene exceeds 2^28 triangles");
            r.tri_first = (int)out.total_tris;
            r.tri_count = cnt;
            if (build_bvh) {
                int depth = 0;
                r.bvh_root = build_mesh_bvh(g.m_triangles, cnt, params, out.nodes, out.tris, &depth, &out.normals);
                out.max_depth = std::max(out.max_depth, depth);
            }
            out.total_tris += (uint64_t)cnt;
        }
    }
    // Processing order: analytic shapes (planes, spheres) first (cheap, they tighten the distance bound), then meshes by ascending triangle count (a
    // lane walks its candidate meshes one after the other; short traversals first keeps the wave's lanes in step for longer).
    // The closest hit does not depend on this order: ties are broken on orig_index exactly like the reference's loop.
    std::stable_sort(out.geoms.begin(), out.geoms.end(), [](const GeomRecord& a, const GeomRecord& b) {
        const bool pa = a.type != FF_GEOM_TRIANGLEMESH, pb = b.type != FF_GEOM_TRIANGLEMESH;
        if (pa != pb) return pa;
        if (pa) {
            if (a.type != b.type) return a.type == FF_GEOM_PLANE; // planes before spheres
            // planes: largest world box first (the walls of a room lead the records: count_scan_planes)
            return a.type == FF_GEOM_PLANE && box_half_area(a.wmin, a.wmax) > box_half_area(b.wmin, b.wmax);
        }
        return a.tri_count < b.tri_count;
    });
    return FF_OK;
}

} // namespace ff

// ---- host-only dry run + structural self-check -------------------------------------------------------------------

namespace ff {
size_t bvh_lds_bytes(int lds_nodes, int stack_depth, int block_threads, int num_geoms);
int max_lds_nodes(int stack_depth, int block_threads, int num_geoms, size_t reserve = 0);

namespace {

// Walks every mesh BVH: each triangle must be referenced by exactly one leaf, and every child box must enclose the
// triangles (and boxes) below it.
bool check_bvh(const CompiledScene& cs, int* out_max_leaf)
{
    std::vector<int> seen(cs.tris.size(), 0);
    int max_leaf = 0;
    bool ok = true;
    struct Item {
        int link;
        float mn[3], mx[3];
    };
    for (const GeomRecord& g : cs.geoms) {
        if (g.type != FF_GEOM_TRIANGLEMESH || g.bvh_root < 0) continue;
        std::vector<Item> todo;
        Item root;
        root.link = g.bvh_root;
        for (int k = 0; k < 3; ++k) { root.mn[k] = -std::numeric_limits<float>::infinity(); root.mx[k] = std::numeric_limits<float>::infinity(); }
        todo.push_back(root);
        const bool single_leaf = cs.nodes[g.bvh_root].left < 0 && cs.nodes[g.bvh_root].left == cs.nodes[g.bvh_root].right;
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            if (it.link >= 0) {
                if (it.link >= (int)cs.nodes.size()) return false;
                const BvhNode& n = cs.nodes[it.link];
                Item l, r;
                l.link = n.left; r.link = n.right;
                for (int k = 0; k < 3; ++k) {
                    l.mn[k] = n.lmin[k]; l.mx[k] = n.lmax[k]; r.mn[k] = n.rmin[k]; r.mx[k] = n.rmax[k];
                    if (n.lmin[k] < it.mn[k] - 1e-3f || n.lmax[k] > it.mx[k] + 1e-3f) ok = false;
                    if (n.rmin[k] < it.mn[k] - 1e-3f || n.rmax[k] > it.mx[k] + 1e-3f) ok = false;
                }
                todo.push_back(l);
                if (!(single_leaf && it.link == g.bvh_root)) todo.push_back(r);
            } else {
                const int ref = ~it.link, first = ref >> 3, count = (ref & 7) + 1;
                max_leaf = std::max(max_leaf, count);
                if (first < g.tri_first || first + count > g.tri_first + g.tri_count) return false;
                for (int i = first; i < first + count; ++i) {
                    ++seen[i];
                    const TriRecord& t = cs.tris[i];
                    // (v0 + e within an ulp of the vertex; the boxes carry a padding of 1e-4 of the mesh extent)
                    const float v1[3] = { t.v0[0] + t.e1[0], t.v0[1] + t.e1[1], t.v0[2] + t.e1[2] };
                    const float v2[3] = { t.v0[0] + t.e2[0], t.v0[1] + t.e2[1], t.v0[2] + t.e2[2] };
                    for (const float* v : { t.v0, v1, v2 })
                        for (int k = 0; k < 3; ++k)
                            if (!(v[k] >= it.mn[k] && v[k] <= it.mx[k])) ok = false;
                }
            }
        }
        // every original index must appear exactly once
        std::vector<int> orig(g.tri_count, 0);
        for (int i = g.tri_first; i < g.tri_first + g.tri_count; ++i) {
            const int o = cs.tris[i].orig_index;
            if (o < 0 || o >= g.tri_count) return false;
            ++orig[o];
        }
        for (int c : orig) if (c != 1) ok = false;
    }
    for (int c : seen) if (c != 1) ok = false;
    if (out_max_leaf) *out_max_leaf = max_leaf;
    return ok;
}

} // namespace
} // namespace ff

extern "C" int ff_scene_info(const FfGeometry* host_geometries, int n, FfSceneInfo* out)
{
    using namespace ff;
    clear_error();
    if (!out) return fail(FF_ERR_INVALID_ARG, "ff_scene_info: out_info is null");
    std::memset(out, 0, sizeof *out);
    CompiledScene cs;
    const BvhBuildParams bp = default_bvh_params();
    const int st = compile_scene(host_geometries, n, bp, cs);
    if (st != FF_OK) return st;
    out->num_geometries = (int)cs.geoms.size();
    for (const GeomRecord& g : cs.geoms) {
        if (g.type == FF_GEOM_TRIANGLEMESH) ++out->num_meshes;
        else ++out->num_planes;
    }
    out->bvh_nodes = (int)cs.nodes.size();
    out->bvh_max_depth = cs.max_depth;
    out->num_triangles = cs.tris.size();
    const int stack_depth = cs.max_depth + 1;
    out->lds_nodes = std::min((int)cs.nodes.size(), std::max(0, max_lds_nodes(stack_depth, 512, (int)cs.geoms.size())));
    out->lds_bytes = (int)bvh_lds_bytes(out->lds_nodes, stack_depth, 512, (int)cs.geoms.size());
    out->device_bytes = cs.geoms.size() * sizeof(GeomRecord) + cs.tris.size() * sizeof(TriRecord) + cs.nodes.size() * sizeof(BvhNode);
    int max_leaf = 0;
    out->valid = check_bvh(cs, &max_leaf) ? 1 : 0;
    out->bvh_max_leaf = max_leaf;
    double area = 0.0;
    for (const BvhNode& nd : cs.nodes) {
        const float* lo[2] = { nd.lmin, nd.rmin };
        const float* hi[2] = { nd.lmax, nd.rmax };
        for (int c = 0; c < 2; ++c) {
            const double dx = (double)hi[c][0] - lo[c][0], dy = (double)hi[c][1] - lo[c][1], dz = (double)hi[c][2] - lo[c][2];
            if (dx >= 0.0 && dy >= 0.0 && dz >= 0.0) area += dx * dy + dy * dz + dz * dx;
        }
    }
    out->bvh_child_area = (float)area;
    return FF_OK;
}

// Host-only: which of the scene's planes the wall table takes (the table the trace kernels screen in world space, see
// ff_internal.h WallTable): out_walls receives up to max_walls entries of 6 values {caller's geometry index, normal axis,
// plane coordinate, centre u, half extent u, ... } - {index, axis, c, cu, hu, cv, hv} as 7 floats each.  Returns the count or a
// negative FfStatus.  Needs no GPU.
extern "C" int ff_debug_wall_table(const FfGeometry* host_geometries, int n, float* out_walls7, int max_walls)
{
    using namespace ff;
    clear_error();
    CompiledScene cs;
    const int st = compile_scene(host_geometries, n, default_bvh_params(), cs, /*build_bvh=*/false);
    if (st != FF_OK) return -st;
    int num_quads = 0;
    for (const GeomRecord& g : cs.geoms) num_quads += g.type == FF_GEOM_PLANE ? 1 : 0;
    WallTable t;
    build_wall_table(cs.geoms.data(), num_quads, t, std::getenv("FF_NO_WALL_PAIRS") == nullptr); // (stateless test helper: the environment decides)
    int count = 0, i = 0;
    for (int axis = 0; axis < 3; ++axis)
        for (int k = 0; k < t.count[axis]; ++k, ++i) {
            const Wall& w = t.w[i];
            for (int side = 0; side < (w.hi_geom1 ? 2 : 1); ++side) { // (an entry that holds two walls: the lower one first)
                if (out_walls7 && count < max_walls) {
                    float* o = out_walls7 + (size_t)count * 7;
                    o[0] = (float)cs.geoms[side ? w.hi_geom1 - 1 : w.geom].orig_index;
                    o[1] = (float)axis;
                    o[2] = side ? w.hi_c : w.c; o[3] = w.cu; o[4] = w.hu; o[5] = w.cv; o[6] = w.hv;
                }
                ++count;
            }
        }
    return count;
}

extern "C" int ff_debug_wall_entries(const FfGeometry* host_geometries, int n)
{
    using namespace ff;
    clear_error();
    CompiledScene cs;
    const int st = compile_scene(host_geometries, n, default_bvh_params(), cs, /*build_bvh=*/false);
    if (st != FF_OK) return -st;
    int num_quads = 0;
    for (const GeomRecord& g : cs.geoms) num_quads += g.type == FF_GEOM_PLANE ? 1 : 0;
    WallTable t;
    build_wall_table(cs.geoms.data(), num_quads, t, std::getenv("FF_NO_WALL_PAIRS") == nullptr); // (stateless test helper: the environment decides)
    return t.count[0] + t.count[1] + t.count[2];
}
