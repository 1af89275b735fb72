"""CPU-only checks of the product library: it loads, exports every symbol of include/firefly/ff_api.h, keeps the
reference's struct layouts, reports errors as status codes (never exit()), and its host-side pieces (scene compiler,
OBJ reader) behave.  No compute call is made here; without a GPU ff_create must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gpupathtracer_amd import scenes
from gpupathtracer_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_RES = "/root/reference/sceneResources"


def test_library_exports_every_declared_symbol(ff):
    lib = ff.load()
    header = open(os.path.join(ROOT, "include", "firefly", "ff_api.h")).read()
    declared = set(re.findall(r"FF_API\s+[\w\s\*]+?\b(ff_\w+)\s*\(", header))
    assert declared == set(ff.EXPORTS), declared ^ set(ff.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ff_version() >= 100


def test_struct_layouts_match_reference_sizes():
    # g++ x86-64 sizes of the reference's structs
    assert (C.sizeof(T.FfTriangle), C.sizeof(T.FfGeometry), C.sizeof(T.FfCamera), C.sizeof(T.FfRay), C.sizeof(T.FfIntersect),
            C.sizeof(T.FfBXDF)) == (96, 208, 108, 24, 40, 60)
    assert T.FfGeometry.m_triangles.offset == 184 and T.FfGeometry.m_numberOfTriangles.offset == 192 and T.FfGeometry.m_bxdf.offset == 200
    assert T.FfGeometry.m_modelMatrix.offset == 40 and T.FfGeometry.m_inverseModelMatrix.offset == 104
    assert T.FfCamera.m_yaw.offset == 60 and T.FfCamera.m_cameraFirstMouseInput.offset == 96 and T.FfCamera.m_xDelta.offset == 100
    assert T.FfIntersect.m_hit.offset == 28 and T.FfIntersect.geometryIndex.offset == 32


def test_bxdf_defaults(ff):
    b = T.FfBXDF()
    ff.load().ff_bxdf_init(C.byref(b))
    assert b.m_type == T.BXDF_COUNT and b.m_albedo.tuple() == (-1, -1, -1) and b.m_intensity == -1  # utilities.h:81-88


def test_no_gpu_means_loud_failure_or_working_device(ff):
    lib = ff.load()
    st = C.c_void_p()
    rc = lib.ff_create(C.byref(st), 0)
    if rc == T.FF_OK:  # running on a GPU box
        lib.ff_destroy(st)
        return
    assert rc in (T.FF_ERR_NO_DEVICE, T.FF_ERR_HIP)
    assert lib.ff_last_error()  # a message, not an exit()
    with pytest.raises(ff.FireflyError):
        ff.Tracer(0)


def test_null_and_invalid_arguments_are_status_codes(ff):
    lib = ff.load()
    assert lib.ff_create(None, 0) == T.FF_ERR_INVALID_ARG
    assert lib.ff_upload_scene(None, None, 0) == T.FF_ERR_INVALID_ARG
    assert lib.ff_render(None, None, None, None, 0, None, 0) == T.FF_ERR_INVALID_ARG
    assert lib.ff_stats(None, None) == T.FF_ERR_INVALID_ARG
    assert lib.ff_destroy(None) == T.FF_OK
    assert lib.ff_scene_info(None, 0, None) == T.FF_ERR_INVALID_ARG
    info = T.FfSceneInfo()
    assert lib.ff_scene_info(None, 0, C.byref(info)) == T.FF_ERR_INVALID_ARG
    assert b"no geometries" in lib.ff_last_error()


def test_unsupported_and_malformed_scenes(ff):
    lib = ff.load()
    info = T.FfSceneInfo()
    # SPHERE (the reference only printf's, kernel.cu:166-169) is a supported geometry; its radius must be positive
    g = (T.FfGeometry * 1)()
    lib.ff_geometry_init(C.byref(g[0]), T.GEOM_SPHERE, T.FfVec3(), T.FfVec3(), T.FfVec3(1, 1, 1), None, 0, 2.0)
    assert g[0].m_sphereRadius == 2.0
    bx = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 1, 1))
    g[0].m_bxdf = C.pointer(bx)
    assert lib.ff_scene_info(g, 1, C.byref(info)) == T.FF_OK and info.num_geometries == 1 and info.num_meshes == 0
    g[0].m_sphereRadius = 0.0
    assert lib.ff_scene_info(g, 1, C.byref(info)) == T.FF_ERR_INVALID_ARG
    g[0].m_geometryType = 7  # not a geometry type (kernel.cu:170-173)
    assert lib.ff_scene_info(g, 1, C.byref(info)) == T.FF_ERR_UNSUPPORTED
    # null BXDF: the reference memcpy's from it unconditionally (kernel.cu:282)
    lib.ff_geometry_init(C.byref(g[0]), T.GEOM_PLANE, T.FfVec3(), T.FfVec3(), T.FfVec3(1, 1, 1), None, 0, 0.0)
    assert lib.ff_scene_info(g, 1, C.byref(info)) == T.FF_ERR_INVALID_ARG
    # non-affine model matrix
    g[0].m_bxdf = C.pointer(bx)
    g[0].m_inverseModelMatrix.m[3] = 0.25
    assert lib.ff_scene_info(g, 1, C.byref(info)) == T.FF_ERR_UNSUPPORTED


@pytest.mark.parametrize("name,builder,tris,planes", [
    ("c1", lambda: scenes.reference_scene(scenes.load_mesh("cube")), 12, 4),
    ("c2", scenes.cornell_wahoo_scene, 5172 + 12, 6),
    ("c3", scenes.blooper_scene, 6036 + 12, 2),
    ("sphere_l2", lambda: scenes.sphere_stress_scene(2), 960 * 16, 6),
])
def test_scene_compiler_self_check(ff, name, builder, tris, planes):
    info = ff.scene_info(builder())
    assert info.valid == 1
    assert info.num_triangles == tris and info.num_planes == planes
    assert 1 <= info.bvh_max_leaf <= 4
    assert info.bvh_max_depth <= 30
    assert info.lds_bytes <= 160 * 1024 and 0 < info.lds_nodes <= info.bvh_nodes


def test_tree_optimiser_lowers_the_summed_box_area_and_keeps_the_tree_valid(ff, monkeypatch):
    """csrc/ff_scene.cpp Builder::optimise (insertion-based optimisation, Bittner et al. 2013): one pass - the default - after the SAH
    build.  The tree stays a valid tree over the same leaves (every triangle in exactly one leaf, boxes enclose, depth bound), the
    summed area of its boxes goes down on the irregular meshes and the result does not depend on the run; a regular tessellation,
    where no move pays, keeps its tree."""
    def info(scene, passes):
        if passes is None:
            monkeypatch.delenv("FF_BVH_OPT_PASSES", raising=False)
        else:
            monkeypatch.setenv("FF_BVH_OPT_PASSES", str(passes))
        return ff.scene_info(scene)
    for builder in (scenes.cornell_wahoo_scene, scenes.blooper_scene):
        scene = builder()
        off, one, dflt, many = info(scene, 0), info(scene, 1), info(scene, None), info(scene, 12)
        for i in (off, one, dflt, many):
            assert i.valid == 1 and i.bvh_max_leaf <= 2 and i.bvh_max_depth <= 30 and i.bvh_nodes == off.bvh_nodes
        assert one.bvh_child_area == dflt.bvh_child_area == info(scene, 1).bvh_child_area
        assert many.bvh_child_area < one.bvh_child_area < 0.98 * off.bvh_child_area
    sphere = scenes.sphere_stress_scene(2)
    assert info(sphere, 0).bvh_child_area == info(sphere, 3).bvh_child_area
    # vertices that are not numbers: the compiler's self-check says so (valid == 0); the optimiser's searches, whose comparisons
    # are all false on such boxes, still end
    rng = np.random.default_rng(3)
    tri = np.zeros((200, 24), dtype=np.float32)
    tri[:, :9] = rng.uniform(-1, 1, (200, 9))
    tri[::7, 3] = np.nan
    tri[::11, 0:9] = np.inf
    red = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))
    broken = scenes.Scene().add_mesh(tri, bxdf=red).add_plane(bxdf=red).finalize()
    assert info(broken, 4).valid == 0 and info(broken, 0).valid == 0


def test_scene_compiler_edge_cases(ff):
    red = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))
    one = np.zeros((1, 24), dtype=np.float32)
    one[0, :9] = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    s = scenes.Scene().add_mesh(one, bxdf=red).add_mesh(np.zeros((0, 24), np.float32), bxdf=red).add_plane(bxdf=red).finalize()
    info = ff.scene_info(s)
    assert info.valid == 1 and info.num_triangles == 1 and info.num_meshes == 2 and info.bvh_nodes == 1
    # degenerate: many identical triangles (SAH cannot separate them -> median splits, bounded depth)
    same = np.repeat(one, 300, axis=0)
    info = ff.scene_info(scenes.Scene().add_mesh(same, bxdf=red).finalize())
    assert info.valid == 1 and info.num_triangles == 300 and info.bvh_max_leaf <= 4 and info.bvh_max_depth <= 30


def test_subdivided_sphere_is_deterministic_and_on_the_sphere():
    base = scenes.load_mesh("sphereBlender")
    a, b = scenes.subdivide_sphere(base, 2), scenes.subdivide_sphere(base, 2)
    assert a.shape == (960 * 16, 24) and np.array_equal(a, b)
    r = np.linalg.norm(a[:, :9].reshape(-1, 3), axis=1)
    assert np.abs(r - 1).max() < 1e-6


def test_obj_loader_handwritten(ff):
    tris = ff.load_obj(os.path.join(ROOT, "tests", "data", "quad_mixed.obj"))
    assert tris.shape == (5, 24)
    # f 1/1/1 2/2/1 3/3/1
    assert tris[0, :9].tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0]
    assert tris[0, 9:15].tolist() == [0, 0, 1, 0, 1, 1] and tris[0, 15:].tolist() == [0, 0, 1] * 3
    # quad 1 2 3 4 -> (1,2,3), (1,3,4); no vt/vn -> zeros (the reference reads out of bounds here, utilities.h:823-824)
    assert tris[1, :9].tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0] and tris[2, :9].tolist() == [0, 0, 0, 1, 1, 0, 0, 1, 0]
    assert not tris[1, 9:].any()
    # negative indices: -1 -> v5, -5 -> v1, -4 -> v2; v//vn corners
    assert tris[3, :9].tolist() == [0.5, 0.5, 1.25, 0, 0, 0, 1, 0, 0] and tris[3, 15:18].tolist() == [0, 1, 0]
    # mixed corner forms on one face
    assert tris[4, :9].tolist() == [0.5, 0.5, 1.25, 0, 0, 0, 1, 0, 0] and tris[4, 11:13].tolist() == [1, 0] and tris[4, 21:].tolist() == [0, 1, 0]


def test_obj_loader_errors(ff, tmp_path):
    with pytest.raises(ff.FireflyError) as e:
        ff.load_obj(str(tmp_path / "missing.obj"))
    assert e.value.status == T.FF_ERR_IO
    p = tmp_path / "empty.obj"
    p.write_text("# nothing\nv 0 0 0\n")
    with pytest.raises(ff.FireflyError) as e:
        ff.load_obj(str(p))
    assert e.value.status == T.FF_ERR_IO


@pytest.mark.skipif(not os.path.isdir(REF_RES), reason="reference assets are only mounted in the build container")
@pytest.mark.parametrize("name", ["cube", "wahoo", "rocketman", "sphereBlender", "sphere"])
def test_obj_loader_matches_reference_tinyobj(ff, name):
    """ff_load_obj vs the fixture the reference's own tiny_obj_loader.h produced (oracle/ref_tinyobj_dump.cpp), bit for bit."""
    mine = ff.load_obj(os.path.join(REF_RES, name + ".obj"))
    ref = scenes.load_mesh(name)
    assert mine.shape == ref.shape
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def test_strip_rows_helper(ff):
    lib = ff.load()
    for h, sr, n in [(1080, 16, 8), (1080, 16, 1), (17, 16, 4), (2160, 16, 8), (5, 16, 3)]:
        rows = [lib.ff_strips_local_rows(h, sr, p, n) for p in range(n)]
        assert sum(rows) == h
    assert lib.ff_strips_local_rows(1080, 16, 0, 8) == 9 * 16 and lib.ff_strips_local_rows(1080, 16, 3, 8) == 8 * 16 + 8
    assert lib.ff_strips_local_rows(0, 16, 0, 1) == 0 and lib.ff_strips_local_rows(10, 0, 0, 1) == 0


def test_save_ppm_matches_reference_format(ff, tmp_path):
    """saveToPPM (utilities.h:842-856): "P3", "W H", "255", then one "r g b" line per pixel in buffer order."""
    import numpy as np
    from gpupathtracer_amd import lib as L
    img = np.arange(4 * 3 * 3, dtype=np.uint8).reshape(3, 4, 3) * 7
    path = tmp_path / "render.ppm"
    L.save_ppm(str(path), img)
    lines = path.read_text().split("\n")
    assert lines[:3] == ["P3", "4 3", "255"]
    body = [tuple(int(v) for v in ln.split()) for ln in lines[3:] if ln]
    assert body == [tuple(int(v) for v in px) for px in img.reshape(-1, 3)]
    with pytest.raises(L.FireflyError) as e:
        L.save_ppm(str(tmp_path / "no_such_dir" / "x.ppm"), img)
    assert e.value.status == 7  # FF_ERR_IO


def test_wall_table_takes_axis_aligned_planes_only(ff):
    """The wall table (csrc/ff_scene.cpp build_wall_table): the planes the kernels screen in world space.  The C2 box: all six
    planes, sorted by normal axis, with the rectangles the model matrices give; a plane turned by 45 degrees, a plane stretched
    1 : 20 and a sphere stay out; quarter turns about any axis (whose cosines glm leaves at 4e-8, not 0) stay in."""
    import ctypes as C
    from gpupathtracer_amd import scenes
    lib = ff.load()

    def table(scene):
        buf = (C.c_float * (7 * 32))()
        n = lib.ff_debug_wall_table(scene.geometries, len(scene), buf, 32)
        assert n >= 0
        return np.frombuffer(buf, dtype=np.float32)[: 7 * n].reshape(n, 7).copy()

    box = scenes.cornell_wahoo_scene()
    w = table(box)
    w_box = w.copy()
    planes = [i for i, s in enumerate(box._specs) if s[0] == T.GEOM_PLANE]
    assert sorted(w[:, 0].astype(int).tolist()) == planes
    assert w[:, 1].astype(int).tolist() == [0, 0, 1, 1, 1, 2]  # left / right, floor / ceiling / light, back
    by_index = {int(r[0]): r for r in w}
    back, floor, ceiling, left, right, light = planes
    assert by_index[back][2] == -2.5 and by_index[floor][2] == -2.5 and by_index[ceiling][2] == 2.5 and by_index[left][2] == -2.5
    assert np.isclose(by_index[light][2], 2.49) and np.allclose(by_index[light][[4, 6]], 1.0)   # scale 2: half extents 1
    assert np.allclose(by_index[back][[3, 5]], 0.0) and np.allclose(by_index[back][[4, 6]], 2.5)
    s = scenes.Scene()
    grey = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.7, 0.7, 0.7))
    s.add_plane((1, 2, 3), (90, 180, 270), (2, 6, 1), grey)        # quarter turns about all three axes, 1 : 3 : 0.5: in
    s.add_plane((0, 0, 0), (0, 45, 0), (3, 3, 3), grey)            # oblique: out
    s.add_plane((0, 1, 0), (90, 0, 0), (1, 20, 1), grey)           # 1 : 20: out
    s.add_plane((0, 0, -4), (0, 0, 30), (3, 3, 3), grey)           # turned about its own normal by 30 degrees: not a rectangle along the axes: out
    s.add_sphere(1.0, (0, 0, 0), (0, 0, 0), (1, 1, 1), grey)
    w = table(s.finalize())
    assert w.shape[0] == 1 and int(w[0, 0]) == 0
    # object z (the quad's normal) ends up along one world axis; the quad's 2 x 6 rectangle along the two others
    assert int(w[0, 1]) in (0, 1, 2) and sorted(np.round(w[0, [4, 6]], 4).tolist()) == [1.0, 3.0]
    assert lib.ff_debug_wall_table(None, 0, None, 0) < 0
    # floor / ceiling and left / right wall have the same rectangles: one entry per pair (the kernel tests the rectangle once, at the
    # wall the ray points at); the light under the ceiling and the back wall stay single
    assert lib.ff_debug_wall_entries(box.geometries, len(box)) == 4
    os.environ["FF_NO_WALL_PAIRS"] = "1"
    try:
        assert lib.ff_debug_wall_entries(box.geometries, len(box)) == 6
        assert np.array_equal(table(box), w_box)
    finally:
        del os.environ["FF_NO_WALL_PAIRS"]


def test_builder_arrivals_wait_for_their_stores():
    """the bottom-up builders of csrc/ff_build.hip hand boxes from thread to thread through agent-scope (sc1)
    stores and an arrival counter; the release is `s_waitcnt vmcnt(0)` in front of the arrival atomic (release_arrival).  The
    check compiles the file for gfx950 (no GPU needed) and scans the ISA: no sc1 store may reach a global_atomic_add without that
    wait in between."""
    import shutil
    import subprocess
    import sys as _sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_arrivals.py")
    out = subprocess.run([_sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 violations" in out.stdout and " 0 arrival atomics" not in out.stdout


def test_bench_reuses_the_one_gpu_counter_file_for_more_ranks():
    """with N > 1 ranks there is no N-GPU counter file (the pool's boxes have one GPU); a rank runs the same
    kernel instantiation on a strip subset of the same frame, so bench.py applies the N = 1 file's per-segment figures when kernel
    name and source hash match, and says so (`pmc_scope: "n1"`).  A file from other sources is refused for every N."""
    import argparse
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ff_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from gpupathtracer_amd.provenance import kernel_source_hash
    files = [f for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if f.endswith("_c2_pmc.json")]
    assert files
    newest = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    w = newest["workload"]
    args = argparse.Namespace(width=w["width"], height=w["height"], bounces=w["bounces"], spp=w["spp"], camera=w["camera"], trace_mode=w["trace"], scene=w.get("scene", "c2"))
    one, why1, scope1 = bench.measured_pmc(args, 1, newest["kernel"])
    many, why8, scope8 = bench.measured_pmc(args, 8, newest["kernel"])
    if newest.get("kernel_source_hash") == kernel_source_hash():
        assert one is not None and scope1 == "exact" and many is not None and scope8 == "n1"
        assert many["valu_insts_per_ray"] == one["valu_insts_per_ray"] and many["file"] == one["file"]
    else:  # the tree has moved on since the counters were taken: withheld for every N, with the reason
        assert one is None and many is None and "other kernel sources" in why1 and scope8 is None
    other, why, _ = bench.measured_pmc(args, 8, "trace_bvh_kernel<false, 512, false, 0, false>")
    assert other is None and why
