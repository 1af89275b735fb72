"""GPU tests of the dynamic-scene row: BVHs built on the device (LBVH), refitted in place and
re-based by transform updates.  The bar is the parity bar of test_gpu_parity.py: whichever tree the device holds, the
frames must equal the oracle's golden frames bit for bit — plus structural checks of the trees themselves.
"""
import os

import numpy as np
import pytest

from cases import CASES, POSES, build_case
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T
from oracle_lib import oracle_render

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.fixture(params=[T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC], ids=["lbvh", "ploc"])
def lbvh_tracer(request):
    """A tracer whose uploads build their trees on the device (either device builder)."""
    with lib.Tracer(0) as t:
        t.set_builder(request.param)
        yield t


def same_bits(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def check_tree(nodes, tris, table, max_leaf=4):
    """Every mesh: its inner nodes form one tree under the root (each reached exactly once, level by level), all of its
    triangles sit in exactly one leaf, every child box encloses what is below it, node count and depth as reported.
    Vectorised so that the million-triangle tree is checked in seconds.  Returns the number of inner nodes walked."""
    walked = 0
    v0 = tris["v0"].astype(np.float64)
    verts = np.stack([v0, v0 + tris["e1"], v0 + tris["e2"]], axis=1)  # [n, 3, 3]
    tmin, tmax = verts.min(axis=1), verts.max(axis=1)
    for root, node_count, tri_first, tri_count, depth in table:
        if root < 0:
            continue
        slot = nodes[root:root + node_count]
        links = np.stack([slot["left"], slot["right"]], axis=1).astype(np.int64)       # [n, 2]
        cmin = np.stack([slot["lmin"], slot["rmin"]], axis=1).astype(np.float64)       # [n, 2, 3]
        cmax = np.stack([slot["lmax"], slot["rmax"]], axis=1).astype(np.float64)
        inner = links >= 0
        assert ((links[inner] >= root) & (links[inner] < root + node_count)).all(), "inner link outside the mesh slot"
        # reachability, level by level
        seen_node = np.zeros(node_count, dtype=np.int32)
        frontier = np.array([0], dtype=np.int64)
        levels = 0
        while frontier.size:
            levels += 1
            assert levels <= 256
            np.add.at(seen_node, frontier, 1)
            nxt = links[frontier]
            frontier = nxt[nxt >= 0] - root
        assert (seen_node == 1).all(), f"{(seen_node != 1).sum()} nodes are not reached exactly once from the root"
        assert levels <= depth, (levels, depth)
        # leaves: coverage and box enclosure
        single = node_count == 1 and links[0, 0] == links[0, 1]
        ref = ~links
        lf, lc = ref >> 3, (ref & 7) + 1
        leaf = ~inner
        if single:
            leaf = leaf.copy()
            leaf[0, 1] = False  # a one-leaf mesh lists its leaf on both sides (tested twice, harmless)
        assert (lc[leaf] <= max_leaf).all()
        assert ((lf[leaf] >= tri_first) & (lf[leaf] + lc[leaf] <= tri_first + tri_count)).all(), "leaf outside the mesh's triangle range"
        cover = np.zeros(tri_count + 1, dtype=np.int64)
        np.add.at(cover, lf[leaf] - tri_first, 1)
        np.add.at(cover, lf[leaf] + lc[leaf] - tri_first, -1)
        assert (np.cumsum(cover)[:-1] == 1).all(), "some triangles are not in exactly one leaf"
        sub_min = np.full((node_count, 2, 3), np.inf)
        sub_max = np.full((node_count, 2, 3), -np.inf)
        allleaf = ~inner
        for k in range(max_leaf):
            m = allleaf & (lc > k)
            idx = lf[m] + k
            sub_min[m] = np.minimum(sub_min[m], tmin[idx])
            sub_max[m] = np.maximum(sub_max[m], tmax[idx])
        # inner children: the union of the child node's two boxes (already padded) must lie inside the parent's slot,
        # up to the padding's own rounding
        own_min, own_max = cmin.min(axis=1), cmax.max(axis=1)
        child = links[inner] - root
        sub_min[inner] = own_min[child]
        sub_max[inner] = own_max[child]
        tol = 1e-6 * max(1.0, float(np.abs(tmin[tri_first:tri_first + tri_count]).max()), float(np.abs(tmax[tri_first:tri_first + tri_count]).max()))
        assert (cmin <= sub_min + tol).all() and (cmax >= sub_max - tol).all(), "a child box does not enclose its subtree"
        orig = np.sort(tris["orig_index"][tri_first:tri_first + tri_count])
        assert np.array_equal(orig, np.arange(tri_count)), "triangle records are not a permutation of the mesh"
        walked += node_count
    return walked


def check_tree4(nodes4, tris, table, table4, max_leaf=4):
    """The 4-wide trees the kernels traverse (derived on the device from the binary ones): per mesh, every node is reached
    exactly once from node 0 through links relative to the mesh's first node, empty slots are inverted boxes, all triangles
    sit in exactly one leaf, every slot box encloses what is below it, node count and depth as reported, and the LDS shares
    of the meshes do not overlap.  Returns the number of nodes walked."""
    walked = 0
    v0 = tris["v0"].astype(np.float64)
    verts = np.stack([v0, v0 + tris["e1"], v0 + tris["e2"]], axis=1)
    tmin, tmax = verts.min(axis=1), verts.max(axis=1)
    lds_ranges = []
    for (root, _, tri_first, tri_count, _), (first4, count4, depth4, lds_first, lds_nodes, lds_cap) in zip(table, table4):
        if root < 0:
            assert first4 < 0 and count4 == 0
            continue
        assert first4 == root and 1 <= count4 and 0 <= lds_nodes <= count4 and lds_first + lds_nodes <= lds_cap
        lds_ranges.append((lds_first, lds_first + lds_nodes))
        slot = nodes4[first4:first4 + count4]
        links = slot["link"].astype(np.int64)                     # [n, 4]
        bmin = np.transpose(slot["mn"], (0, 2, 1)).astype(np.float64)  # [n, 4 slots, 3 axes]
        bmax = np.transpose(slot["mx"], (0, 2, 1)).astype(np.float64)
        empty = links == T.BVH4_EMPTY_LINK
        inner = (links >= 0) & ~empty
        leaf = links < 0
        assert (~empty[:, 0]).all(), "a node without a first slot"
        assert (np.diff(empty.astype(np.int8), axis=1) >= 0).all(), "empty slots must trail"
        assert np.isposinf(bmin[empty]).all() and np.isneginf(bmax[empty]).all()
        assert (links[inner] < count4).all(), "inner link outside the mesh's nodes"
        seen = np.zeros(count4, dtype=np.int32)
        frontier = np.array([0], dtype=np.int64)
        levels = 0
        while frontier.size:
            levels += 1
            assert levels <= 256
            np.add.at(seen, frontier, 1)
            nxt = links[frontier]
            frontier = nxt[(nxt >= 0) & (nxt != T.BVH4_EMPTY_LINK)]
        assert (seen == 1).all(), f"{(seen != 1).sum()} 4-wide nodes are not reached exactly once"
        assert levels == depth4, (levels, depth4)
        ref = ~links
        lf, lc = ref >> 3, (ref & 7) + 1
        assert (lc[leaf] <= max_leaf).all()
        assert ((lf[leaf] >= tri_first) & (lf[leaf] + lc[leaf] <= tri_first + tri_count)).all()
        cover = np.zeros(tri_count + 1, dtype=np.int64)
        np.add.at(cover, lf[leaf] - tri_first, 1)
        np.add.at(cover, lf[leaf] + lc[leaf] - tri_first, -1)
        assert (np.cumsum(cover)[:-1] == 1).all(), "some triangles are not in exactly one leaf of the 4-wide tree"
        sub_min = np.full((count4, 4, 3), np.inf)
        sub_max = np.full((count4, 4, 3), -np.inf)
        for k in range(max_leaf):
            m = leaf & (lc > k)
            idx = lf[m] + k
            sub_min[m] = np.minimum(sub_min[m], tmin[idx])
            sub_max[m] = np.maximum(sub_max[m], tmax[idx])
        own_min, own_max = bmin.min(axis=1), bmax.max(axis=1)  # (empty slots are inverted: they do not contribute)
        child = links[inner]
        sub_min[inner] = own_min[child]
        sub_max[inner] = own_max[child]
        tol = 1e-6 * max(1.0, float(np.abs(tmin[tri_first:tri_first + tri_count]).max()), float(np.abs(tmax[tri_first:tri_first + tri_count]).max()))
        used = ~empty
        assert (bmin[used] <= sub_min[used] + tol).all() and (bmax[used] >= sub_max[used] - tol).all(), "a slot box does not enclose its subtree"
        walked += count4
    lds_ranges.sort()
    for (a0, a1), (b0, b1) in zip(lds_ranges, lds_ranges[1:]):
        assert a1 <= b0, "LDS shares overlap"
    return walked


def check_trees(t, num_geometries):
    """Binary trees (what the builders wrote) and the 4-wide trees derived from them, as the device holds them now."""
    nodes, tris, table = t.download_bvh(num_geometries)
    walked = check_tree(nodes, tris, table)
    nodes4, table4 = t.download_bvh4(num_geometries)
    check_tree4(nodes4, tris, table, table4)
    return walked


@pytest.mark.parametrize("name", list(CASES))
def test_golden_frames_with_device_built_trees(lbvh_tracer, golden, name):
    scene, cam, params = build_case(name)
    lbvh_tracer.upload_scene(scene)
    bs = lbvh_tracer.build_stats()
    assert bs.builder in (T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC) and bs.num_triangles == scene.triangle_count
    rgb8, rad = lbvh_tracer.render(cam, params)
    assert np.array_equal(rgb8, golden[name + "/rgb8"]), f"{name}: {(rgb8 != golden[name + '/rgb8']).any(axis=2).sum()} pixels differ"
    assert same_bits(rad, golden[name + "/radiance"]), f"{name}: radiance not bit-identical"


@pytest.mark.parametrize("mesh", ["cube", "sphere", "sphereBlender", "wahoo", "rocketman"])
def test_device_built_tree_structure(lbvh_tracer, mesh):
    scene = scenes.reference_scene(scenes.load_mesh(mesh))
    lbvh_tracer.upload_scene(scene)
    nodes, tris, table = lbvh_tracer.download_bvh(len(scene))
    walked = check_tree(nodes, tris, table)
    nodes4, table4 = lbvh_tracer.download_bvh4(len(scene))
    assert 1 <= check_tree4(nodes4, tris, table, table4) <= walked
    bs = lbvh_tracer.build_stats()
    assert walked >= 1 and bs.bvh_max_depth >= 1
    # the records are the host compiler's records (same arithmetic), only the order differs
    with lib.Tracer(0) as host:
        host.upload_scene(scene)
        _, htris, _ = host.download_bvh(len(scene))
    key = np.argsort(tris["orig_index"], kind="stable")
    hkey = np.argsort(htris["orig_index"], kind="stable")
    assert np.array_equal(tris[key].view(np.uint8), htris[hkey].view(np.uint8))


@pytest.mark.parametrize("builder", [T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC], ids=["host_tree", "lbvh_tree", "ploc_tree"])
def test_collapse_rule_and_treelet_passes_change_the_tree_not_the_frame(monkeypatch, builder):
    """The 4-wide collapse that minimises the summed node area (csrc/ff_build.hip) against the fixed rule of rounds 1-2
    (FF_COLLAPSE_PARITY=1), and the device builders with and without their treelet restructuring passes (FF_TREELET_PASSES):
    every variant is a well-formed tree, renders the same bits, and the better tree is the smaller / cheaper one."""
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(96, 72, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    params = lib.render_params(96, 72, 6, 4, 9)
    out = {}
    variants = [("optimal", {}), ("parity", {"FF_COLLAPSE_PARITY": "1"})]
    if builder != T.BUILD_HOST_SAH:
        variants += [("no_treelets", {"FF_TREELET_PASSES": "0"}), ("three_passes", {"FF_TREELET_PASSES": "3"})]
    for name, env in variants:
        for k in ("FF_COLLAPSE_PARITY", "FF_TREELET_PASSES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            check_trees(t, len(scene))
            t.set_collect_stats(True)
            rgb8, rad = t.render(cam, params)
            st = t.stats()
            out[name] = (rgb8, rad, st.scene_bytes_nodes // 112, st.nodes_visited / st.rays_traced, st.rays_traced)
    ref = out["optimal"]
    for name, o in out.items():
        assert np.array_equal(o[0], ref[0]) and same_bits(o[1], ref[1]) and o[4] == ref[4], name
    assert out["optimal"][2] < out["parity"][2]          # fewer 4-wide nodes ...
    assert out["optimal"][3] < out["parity"][3] * 1.02   # ... and no more visits per ray
    if builder != T.BUILD_HOST_SAH:
        assert out["three_passes"][3] < out["no_treelets"][3], "treelet restructuring did not lower the node visits per ray"


@pytest.mark.parametrize("builder", [T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC], ids=["lbvh_tree", "ploc_tree"])
def test_parallel_reinsertion_changes_the_tree_not_the_frame(monkeypatch, builder):
    """csrc/ff_build.hip reinsert_*_kernel (FF_GPU_REINSERT passes, eight by default on meshes of up to 65 536 triangles, four beyond): nodes are taken out and put back where they add
    the least surface area, many at a time (locks on the six nodes a move rewires, no move into another move's subtree).  Every
    variant is a well-formed tree over the same leaves and renders the same bits; the passes lower the node visits per ray; a
    mesh where every move conflicts with its neighbours' (a strip of identical triangles) and tiny meshes come through."""
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(96, 72, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    params = lib.render_params(96, 72, 6, 4, 9)
    out = {}
    for passes in ("0", None, "9"):
        monkeypatch.delenv("FF_GPU_REINSERT", raising=False)
        if passes is not None:
            monkeypatch.setenv("FF_GPU_REINSERT", passes)
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            check_trees(t, len(scene))
            t.set_collect_stats(True)
            rgb8, rad = t.render(cam, params)
            st = t.stats()
            out[passes] = (rgb8, rad, st.nodes_visited / st.rays_traced, st.rays_traced)
    for o in out.values():
        assert np.array_equal(o[0], out["0"][0]) and same_bits(o[1], out["0"][1]) and o[3] == out["0"][3]
    assert out["9"][2] <= out[None][2] * 1.01 and out[None][2] < out["0"][2], [o[2] for o in out.values()]
    # the passes do not depend on the run - which of two conflicting moves wins is decided by (gain, node), not by who came first:
    # two builds walk the same number of nodes and test the same number of triangles for the same frame
    monkeypatch.delenv("FF_GPU_REINSERT", raising=False)
    twice = []
    for _ in range(2):
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            t.set_collect_stats(True)
            t.render(cam, params)
            st = t.stats()
            twice.append((st.nodes_visited, st.tris_tested, st.scene_bytes_nodes))
    if builder == T.BUILD_GPU_LBVH:  # (PLOC numbers its nodes in the order its merges happen to finish: same clusters, another numbering, other ties)
        assert twice[0] == twice[1]
    # degenerate inputs: 300 identical triangles, and meshes of 2 .. 9 triangles, with many passes
    monkeypatch.setenv("FF_GPU_REINSERT", "16")
    red = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))
    one = np.zeros((1, 24), dtype=np.float32)
    one[0, :9] = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    small = scenes.posed_camera(48, 36, position=(0.3, 0.3, 2.4), yaw=-90.0, pitch=0.0)
    sp = lib.render_params(48, 36, 3, 2, 4)
    for count in (300, 2, 3, 5, 9):
        sc = scenes.Scene().add_mesh(np.repeat(one, count, axis=0), bxdf=red).add_plane((0, 0, -1), (0, 0, 0), (4, 4, 4), red).finalize()
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(sc)
            check_trees(t, len(sc))
            got = t.render(small, sp)
            sp.trace_mode = T.TRACE_BRUTE_FORCE
            ref = t.render(small, sp)
            sp.trace_mode = T.TRACE_BVH
            assert np.array_equal(got[0], ref[0]) and same_bits(got[1], ref[1]), count


def test_host_built_tree_structure(tracer):
    scene = scenes.cornell_wahoo_scene()
    tracer.upload_scene(scene)
    nodes, tris, table = tracer.download_bvh(len(scene))
    walked = check_tree(nodes, tris, table)
    nodes4, table4 = tracer.download_bvh4(len(scene))
    walked4 = check_tree4(nodes4, tris, table, table4)
    assert walked4 * 2 <= walked + 2 * len(scene)  # every second level went away
    assert table4[:, 4].sum() == min(table4[:, 1].sum(), table4[0, 5]) or table4[:, 4].sum() <= table4[0, 5]


def test_host_builder_threads_do_not_change_the_tree(tracer, monkeypatch):
    """Meshes of 65 536 triangles and more are built by several host threads (csrc/ff_scene.cpp Builder::build: the big subtrees
    are finished in parallel and spliced in a fixed order): nodes, triangle records and the derived 4-wide tree are the same
    bytes for 1, 3 and 16 threads."""
    scene = scenes.sphere_stress_scene(3)  # 61 440 triangles: below the threshold, so FF_BVH_THREADS forces the threaded path
    assert scene.triangle_count == 61440
    out = []
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("FF_BVH_THREADS", threads)
        tracer.upload_scene(scene)
        nodes, tris, table = tracer.download_bvh(len(scene))
        nodes4, table4 = tracer.download_bvh4(len(scene))
        out.append((nodes.tobytes(), tris.tobytes(), table.tobytes(), nodes4.tobytes(), table4.tobytes()))
    assert out[0] == out[1] == out[2]


def test_duplicate_centroids_and_degenerate_extent(lbvh_tracer):
    """Identical Morton codes (coincident triangles) and a mesh that is flat in one axis must still give a well-formed tree."""
    tri = np.zeros((1, 24), dtype=np.float32)
    tri[0, 0:9] = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    flat = np.repeat(tri, 37, axis=0)          # 37 coincident triangles in the z = 0 plane
    flat[20:, 0:9:3] += 2.0                    # ... and a second coincident pile shifted in x
    scene = scenes.reference_scene(flat)
    lbvh_tracer.upload_scene(scene)
    check_trees(lbvh_tracer, len(scene))


def _with_mesh(scene, geometry_index, triangles=None, position=None, rotation=None, scale=None):
    out = scenes.Scene()
    for i, (kind, pos, rot, scl, tris, bxdf) in enumerate(scene._specs):
        if i == geometry_index:
            tris = tris if triangles is None else np.asarray(triangles, dtype=np.float32)
            pos = pos if position is None else position
            rot = rot if rotation is None else rotation
            scl = scl if scale is None else scale
        out._specs.append((kind, pos, rot, scl, tris, bxdf))
    return out.finalize()


def _mesh_index(scene, which=0):
    idx = [i for i, s in enumerate(scene._specs) if s[0] == T.GEOM_TRIANGLEMESH]
    return idx[which]


def _deform(triangles, phase):
    """A smooth, clearly visible deformation of a mesh (vertex positions only)."""
    out = np.array(triangles, dtype=np.float32, copy=True)
    for v in range(3):
        x, y, z = out[:, 3 * v], out[:, 3 * v + 1], out[:, 3 * v + 2]
        out[:, 3 * v] = (x + 0.35 * np.sin(1.3 * y + phase)).astype(np.float32)
        out[:, 3 * v + 2] = (z * (1.0 + 0.25 * np.cos(0.7 * y + phase))).astype(np.float32)
    return out


@pytest.mark.parametrize("builder", [T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC], ids=["host_tree", "lbvh_tree", "ploc_tree"])
@pytest.mark.parametrize("mode", [T.UPDATE_REFIT, T.UPDATE_REBUILD], ids=["refit", "rebuild"])
def test_update_mesh_matches_fresh_upload_and_oracle(builder, mode):
    scene = scenes.cornell_wahoo_scene()
    gi = _mesh_index(scene, 0)
    moved = _deform(scene._specs[gi][4], phase=0.6)
    scene2 = _with_mesh(scene, gi, triangles=moved)
    cam = scenes.posed_camera(72, 54, **POSES["default"])
    params = lib.render_params(72, 54, 4, 2, 99, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    with lib.Tracer(0) as t:
        t.set_builder(builder)
        t.upload_scene(scene)
        if mode == T.UPDATE_REBUILD and builder == T.BUILD_HOST_SAH:
            with pytest.raises(lib.FireflyError) as e:
                t.update_mesh(gi, moved, mode)
            assert e.value.status == T.FF_ERR_UNSUPPORTED
            return
        before = t.render(cam, params)[1]
        t.update_mesh(gi, moved, mode)
        bs = t.build_stats()
        assert bs.last_operation == (1 if mode == T.UPDATE_REFIT else 2)
        check_trees(t, len(scene))
        rgb8, rad = t.render(cam, params)
    with lib.Tracer(0) as fresh:
        fresh.upload_scene(scene2)
        rgb8_f, rad_f = fresh.render(cam, params)
    assert not same_bits(before, rad), "the deformation should change the image"
    assert np.array_equal(rgb8, rgb8_f) and same_bits(rad, rad_f)
    o_rgb8, o_rad = oracle_render(scene2, cam, params)
    assert np.array_equal(rgb8, o_rgb8) and same_bits(rad, o_rad)


def test_refit_twice_then_rebuild(lbvh_tracer):
    """A tree refitted over several frames and finally rebuilt keeps giving the frames of a fresh upload."""
    scene = scenes.blooper_scene()
    gi = _mesh_index(scene, 0)
    base = scene._specs[gi][4]
    cam = scenes.posed_camera(64, 64, **POSES["oblique"])
    params = lib.render_params(64, 64, 3, 2, 5, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    lbvh_tracer.upload_scene(scene)
    for step, mode in enumerate([T.UPDATE_REFIT, T.UPDATE_REFIT, T.UPDATE_REBUILD, T.UPDATE_REFIT]):
        moved = _deform(base, phase=0.4 * (step + 1))
        lbvh_tracer.update_mesh(gi, moved, mode)
        check_trees(lbvh_tracer, len(scene))
        rad = lbvh_tracer.render(cam, params)[1]
        with lib.Tracer(0) as fresh:
            fresh.upload_scene(_with_mesh(scene, gi, triangles=moved))
            assert same_bits(rad, fresh.render(cam, params)[1]), f"step {step}"


@pytest.mark.parametrize("builder", [T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC], ids=["host_tree", "lbvh_tree", "ploc_tree"])
def test_update_transforms_matches_fresh_upload(builder):
    scene = scenes.cornell_wahoo_scene()
    gi = _mesh_index(scene, 0)
    scene2 = _with_mesh(scene, gi, position=(0.6, -2.2, -0.4), rotation=(0.0, 35.0, 10.0), scale=(0.22, 0.3, 0.25))
    cam = scenes.posed_camera(72, 54, **POSES["default"])
    params = lib.render_params(72, 54, 4, 2, 3, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    with lib.Tracer(0) as t:
        t.set_builder(builder)
        t.upload_scene(scene)
        before = t.render(cam, params)[1]
        t.update_transforms(scene2)
        assert t.build_stats().last_operation == 3
        rgb8, rad = t.render(cam, params)
    o_rgb8, o_rad = oracle_render(scene2, cam, params)
    assert not same_bits(before, rad)
    assert np.array_equal(rgb8, o_rgb8) and same_bits(rad, o_rad)


def test_update_transforms_of_walls(tracer):
    """Translating, rotating and scaling PLANES through ff_update_transforms: the scene compiler orders the planes by the size
    of their world boxes, which all three change, so the update has to match records by the caller's index, not by position
    (ADVICE r2).  Every step gives the frame of a fresh upload and of the oracle; the last one turns an axis-aligned wall
    (screened through the wall table) into an oblique one and back."""
    scene = scenes.cornell_wahoo_scene()
    planes = [i for i, s in enumerate(scene._specs) if s[0] == T.GEOM_PLANE]
    back, floor, ceiling, left, right, light = planes
    cam = scenes.posed_camera(72, 54, **POSES["default"])
    inside = scenes.posed_camera(72, 54, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    params = lib.render_params(72, 54, 4, 2, 3, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    steps = [
        # the back wall shrinks below every other wall, the light grows above them: their positions in the sorted records swap
        [(back, dict(scale=(1.5, 1.5, 1.5))), (light, dict(scale=(6.0, 6.0, 6.0), position=(0.0, 2.45, 0.0)))],
        # the right wall moves out and tilts, the floor is stretched along one axis only
        [(right, dict(position=(2.9, 0.2, 0.1), rotation=(10.0, 70.0, 5.0))), (floor, dict(scale=(7.0, 5.0, 3.0)))],
        # the left wall turns by another quarter (still axis-aligned), the ceiling by 45 degrees about its normal
        [(left, dict(rotation=(0.0, 270.0, 90.0))), (ceiling, dict(rotation=(90.0, 0.0, 45.0)))],
        [],  # and back to the scene as uploaded
    ]
    tracer.upload_scene(scene)
    for k, step in enumerate(steps):
        moved = scene
        for gi, change in step:
            moved = _with_mesh(moved, gi, **change)
        tracer.update_transforms(moved)
        for c in (cam, inside):
            rgb8, rad = tracer.render(c, params)
            with lib.Tracer(0) as fresh:
                fresh.upload_scene(moved)
                f_rgb8, f_rad = fresh.render(c, params)
            assert np.array_equal(rgb8, f_rgb8) and same_bits(rad, f_rad), f"step {k}: differs from a fresh upload"
            o_rgb8, o_rad = oracle_render(moved, c, params)
            assert np.array_equal(rgb8, o_rgb8) and same_bits(rad, o_rad), f"step {k}: differs from the oracle"


def test_update_errors(tracer):
    scene = scenes.cornell_wahoo_scene()
    gi = _mesh_index(scene, 0)
    tris = scene._specs[gi][4]
    with lib.Tracer(0) as t:
        with pytest.raises(lib.FireflyError) as e:
            t.update_mesh(gi, tris)
        assert e.value.status == T.FF_ERR_NO_SCENE
        with pytest.raises(lib.FireflyError) as e:
            t.update_transforms(scene)
        assert e.value.status == T.FF_ERR_NO_SCENE
        with pytest.raises(lib.FireflyError) as e:
            t.set_builder(7)
        assert e.value.status == T.FF_ERR_INVALID_ARG
        t.upload_scene(scene)
        with pytest.raises(lib.FireflyError) as e:
            t.update_mesh(gi, tris[:-1])
        assert e.value.status == T.FF_ERR_INVALID_ARG
        plane = [i for i, s in enumerate(scene._specs) if s[0] == T.GEOM_PLANE][0]
        with pytest.raises(lib.FireflyError) as e:
            t.update_mesh(plane, tris)
        assert e.value.status == T.FF_ERR_INVALID_ARG
        with pytest.raises(lib.FireflyError) as e:
            t.update_transforms(scenes.blooper_scene())
        assert e.value.status == T.FF_ERR_INVALID_ARG


def test_million_triangle_device_build(lbvh_tracer):
    """BASELINE config 4 (983 040 triangles): device build, structure, and the same frame as the host-built tree."""
    scene = scenes.sphere_stress_scene(5)
    lbvh_tracer.upload_scene(scene)
    bs = lbvh_tracer.build_stats()
    assert bs.num_triangles == scene.triangle_count >= 983040
    check_trees(lbvh_tracer, len(scene))
    cam = scenes.posed_camera(160, 90, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    params = lib.render_params(160, 90, 4, 2, 11, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    rgb8, rad = lbvh_tracer.render(cam, params)
    with lib.Tracer(0) as host:
        host.upload_scene(scene)
        rgb8_h, rad_h = host.render(cam, params)
    assert np.array_equal(rgb8, rgb8_h) and same_bits(rad, rad_h)


def _to_u8(mean):
    s = mean * np.float32(255.0)
    out = np.where(s > 0, np.minimum(s, np.float32(255.0)), np.float32(0.0))
    return np.where(s >= 255.0, 255, out.astype(np.int32)).astype(np.uint8)


def test_progressive_accumulation_matches_oracle_frames(tracer):
    """Frame f of a progressive sequence = the oracle's frame with seed + f; running fp32 sum in frame order; output =
    sum * (1 / frames).  Checked bit for bit after every frame, including a restart."""
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(48, 36, **POSES["default"])
    params = lib.render_params(48, 36, 3, 2, 21, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    tracer.upload_scene(scene)
    acc = None
    for f in range(3):
        p = lib.render_params(48, 36, 3, 2, 21 + f, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
        _, frame = oracle_render(scene, cam, p)
        acc = frame.copy() if acc is None else (acc + frame).astype(np.float32)
        mean = (acc * (np.float32(1.0) / np.float32(f + 1))).astype(np.float32)
        rgb8, rad = tracer.render_progressive(cam, params, f)
        assert same_bits(rad, mean), f"frame {f}"
        assert np.array_equal(rgb8, _to_u8(mean)), f"frame {f}"
    with pytest.raises(lib.FireflyError) as e:
        tracer.render_progressive(cam, params, 5)  # does not continue the sequence
    assert e.value.status == T.FF_ERR_INVALID_ARG
    _, first = tracer.render(cam, params)
    rgb8, rad = tracer.render_progressive(cam, params, 0)  # restart: a single frame again
    assert same_bits(rad, first)


@pytest.mark.parametrize("count", [1, 2, 3, 4, 5, 7])
def test_tiny_meshes_on_every_builder(count):
    """Meshes at and around the leaf size (one-leaf trees are assembled on the host, the rest built on the device)."""
    wahoo = scenes.load_mesh("wahoo")
    tris = wahoo[100:100 + count] * np.float32(1.0)
    tris[:, 0:9] *= np.float32(4.0)  # make them visible
    scene = scenes.reference_scene(tris)
    cam = scenes.posed_camera(64, 48, **POSES["oblique"])
    params = lib.render_params(64, 48, 1, 1, 1, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)
    o_rgb8, o_rad = oracle_render(scene, cam, params)
    for builder in (T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC):
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            check_trees(t, len(scene))
            rgb8, rad = t.render(cam, params)
            assert np.array_equal(rgb8, o_rgb8) and same_bits(rad, o_rad), (builder, count)
            if builder != T.BUILD_HOST_SAH:
                t.update_mesh(_mesh_index(scene), tris, T.UPDATE_REBUILD)
            t.update_mesh(_mesh_index(scene), tris, T.UPDATE_REFIT)
            check_trees(t, len(scene))
            assert same_bits(t.render(cam, params)[1], o_rad)


@pytest.mark.parametrize("builder", [T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC, T.BUILD_HOST_SAH])
def test_clustered_mesh_gives_a_deep_tree_that_still_renders(builder):
    """Triangles whose spacing shrinks geometrically (every Morton split peels one off): the device builders produce their
    deepest trees here.  The traversal stack lives in LDS (4 bytes x workgroup size per level of the 4-wide tree), so the
    library must pick a workgroup size whose stacks fit — or say that it cannot — instead of failing the launch."""
    rng = np.random.default_rng(3)
    n = 1500
    tris = np.zeros((n, 24), dtype=np.float32)
    centre = np.stack([2.0 ** -(np.arange(n) / 25.0), 0.7 * 2.0 ** -(np.arange(n) / 31.0), 0.4 * 2.0 ** -(np.arange(n) / 19.0)], axis=1)
    size = 0.02 * 2.0 ** -(np.arange(n) / 25.0)
    for v in range(3):
        tris[:, 3 * v:3 * v + 3] = (centre + rng.normal(size=(n, 3)) * size[:, None]).astype(np.float32)
    red = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.8, 0.3, 0.2))
    light = scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0)
    scene = scenes.Scene().add_mesh(tris, (-1.0, -0.5, 0.0), (0, 30, 0), (3, 3, 3), red) \
        .add_plane((0, 0, -2.5), (0, 0, 0), (8, 8, 8), red).add_plane((0, 3, 0), (90, 0, 0), (8, 8, 8), light).finalize()
    cam = scenes.posed_camera(96, 64, position=(0.0, 0.3, 4.0), yaw=-90.0, pitch=0.0)
    with lib.Tracer(0) as t:
        t.set_builder(builder)
        t.upload_scene(scene)
        check_trees(t, len(scene))
        _, table4 = t.download_bvh4(len(scene))
        depth4 = int(table4[:, 2].max())
        brute = t.render(cam, lib.render_params(96, 64, 4, 2, seed=3, trace_mode=T.TRACE_BRUTE_FORCE))
        assert brute[1].max() > 0
        try:
            bvh = t.render(cam, lib.render_params(96, 64, 4, 2, seed=3))
        except lib.FireflyError as e:
            assert e.status == T.FF_ERR_UNSUPPORTED and "traversal stack" in e.message and depth4 > 70
        else:
            assert np.array_equal(bvh[0], brute[0]) and same_bits(bvh[1], brute[1])
            block = int(t.kernel_name().split(",")[1])
            assert (depth4 + 1) * block * 4 + len(scene) * 288 <= 160 * 1024
            rays = np.stack([np.tile([0.0, 0.3, 4.0], (200, 1)), rng.normal(size=(200, 3))], axis=0).astype(np.float32)
            a = t.intersect_rays(rays[0], rays[1], T.TRACE_BVH)
            b = t.intersect_rays(rays[0], rays[1], T.TRACE_BRUTE_FORCE)
            assert np.array_equal(a["hit"], b["hit"]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
    if builder == T.BUILD_GPU_LBVH:
        assert depth4 >= 12  # the point of the test: this input is deep
