"""Randomised parity on the GPU (tools/fuzz_parity.py): random scenes of meshes, planes and spheres under random
rotations and non-uniform scales, random diffuse / mirror / glass / emitter surfaces, random cameras and seeds."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity as fz  # noqa: E402
from gpupathtracer_amd import lib, scenes  # noqa: E402
from gpupathtracer_amd import types as T  # noqa: E402
from oracle_lib import oracle_render  # noqa: E402

pytestmark = pytest.mark.gpu


def test_bvh_equals_brute_force_on_random_scenes():
    """Both tree kinds against the reference loop (kernel.cu:133-155): radiance bits, rgb8 bytes and ray counts."""
    bad, rays = fz.run(40, seed=2026, verbose=False)
    assert bad == 0 and rays > 500000


def test_crowded_scenes_equal_brute_force():
    """Scenes of 35-120 geometries (more than the 32 whose records live in LDS): a query walks the tree over the geometries'
    world boxes instead of testing every record, and must find what the reference's loop over all geometries finds."""
    bad, rays = fz.run(12, seed=77, verbose=False, crowd_fraction=1.0)
    assert bad == 0 and rays > 100000


def test_five_hundred_geometries_equal_brute_force_and_oracle(tracer):
    """kernel.cu:133 loops over any number of geometries: 500 small cubes / spheres / quads around two meshes, bitwise equal to
    the brute-force kernel in path mode, to the oracle on a small frame, and through ff_intersect_rays."""
    from oracle_lib import oracle_intersect
    rng = np.random.default_rng(2027)
    scene = fz.rand_scene(rng, small=True, crowd=500)
    assert len(scene) > 500
    w, h, cam = fz.rand_view(rng, max_w=160, max_h=120)
    for builder in (T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH):
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            out = {}
            for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
                out[mode] = t.render(cam, lib.render_params(w, h, 6, 3, 99, mode, T.SHADE_DIFFUSE_PATH_SMOOTH, T.GRID_FULL, 0)), t.stats().rays_traced
            assert out[T.TRACE_BVH][1] == out[T.TRACE_BRUTE_FORCE][1]
            assert np.array_equal(out[T.TRACE_BVH][0][0], out[T.TRACE_BRUTE_FORCE][0][0])
            assert np.array_equal(out[T.TRACE_BVH][0][1].view(np.uint32), out[T.TRACE_BRUTE_FORCE][0][1].view(np.uint32))
    # (after the brute-force frame the name is the brute-force kernel's: ask right after a BVH frame)
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        t.render(cam, lib.render_params(32, 24, 1, 1))
        assert t.kernel_name().endswith(", true, 2, false>")  # records in global memory, two-level traversal
        t.upload_scene(fz.rand_scene(np.random.default_rng(3), small=True, crowd=80))
        t.render(cam, lib.render_params(32, 24, 1, 1))
        assert t.kernel_name().endswith(", true, 1, false>")  # records still in LDS
    small = scenes.posed_camera(28, 20, position=(0.5, 0.2, 4.5), yaw=-95.0, pitch=-4.0)
    p = lib.render_params(28, 20, 4, 2, 3, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    tracer.upload_scene(scene)
    rgb8, rad = tracer.render(small, p)
    o_rgb8, o_rad = oracle_render(scene, small, p, threads=16)
    assert np.array_equal(rgb8, o_rgb8) and np.array_equal(rad.view(np.uint32), o_rad.view(np.uint32)) and o_rad.max() > 0
    o = rng.uniform(-3, 3, size=(300, 3)).astype(np.float32)
    d = rng.normal(size=(300, 3)).astype(np.float32)
    got, exp = tracer.intersect_rays(o, d, T.TRACE_BVH), oracle_intersect(scene, o, d)
    hit = exp["hit"] == 1
    assert np.array_equal(got["hit"], exp["hit"]) and np.array_equal(got["geom"][hit], exp["geom"][hit])
    assert np.array_equal(got["t"][hit].view(np.uint32), exp["t"][hit].view(np.uint32))
    # transforms of a big scene: the geometry tree is rebuilt over the moved boxes
    moved = fz.rand_scene(np.random.default_rng(2027), small=True, crowd=500)
    tracer.update_transforms(moved)
    again = tracer.render(small, p)
    assert np.array_equal(again[1].view(np.uint32), rad.view(np.uint32))


@pytest.mark.parametrize("walls,crowd", [(3, 40), (8, 30), (12, 33), (34, 2)])
def test_big_scenes_with_walls_screened_first(tracer, walls, crowd):
    """Scenes of more than 32 geometries keep the planes that span a good part of the scene out of the geometry tree (at most eight,
    largest first; csrc/ff_scene.cpp count_scan_planes) and screen them before the walk through the tree: with fewer walls than the
    cap, exactly the cap, more than the cap (the rest stay in the tree) and almost nothing but walls, the frame equals the
    brute-force loop's bit for bit, with and without the split (FF_NO_SCAN_PLANES), and ff_intersect_rays agrees with the oracle."""
    from oracle_lib import oracle_intersect
    rng = np.random.default_rng(100 + walls)
    cube = scenes.load_mesh("cube")
    s = scenes.Scene()
    for i in range(walls):  # big quads through and around the room, some of them overlapping, one an emitter
        pos = tuple(float(v) for v in rng.uniform(-2.5, 2.5, 3))
        rot = tuple(float(v) for v in rng.uniform(-180, 180, 3))
        size = tuple(float(v) for v in rng.uniform(4.0, 9.0, 3))
        bx = scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0) if i == 0 else scenes.make_bxdf(
            T.BXDF_DIFFUSE, albedo=tuple(float(v) for v in rng.uniform(0.3, 0.9, 3)))
        s.add_plane(pos, rot, size, bx)
    for _ in range(crowd):
        k = int(rng.integers(0, 3))
        pos, rot = tuple(float(v) for v in rng.uniform(-2.2, 2.2, 3)), tuple(float(v) for v in rng.uniform(-180, 180, 3))
        bx = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=tuple(float(v) for v in rng.uniform(0.3, 0.9, 3)))
        if k == 0:
            s.add_mesh(cube, pos, rot, tuple(float(v) for v in rng.uniform(0.1, 0.4, 3)), bx)
        elif k == 1:
            s.add_sphere(float(rng.uniform(0.08, 0.3)), pos, rot, (1, 1, 1), bx)
        else:
            s.add_plane(pos, rot, tuple(float(v) for v in rng.uniform(0.15, 0.5, 3)), bx)
    scene = s.finalize()
    assert len(scene) > 32
    w, h = 96, 64
    cam = scenes.posed_camera(w, h, position=(0.3, 0.2, 6.0), yaw=-92.0, pitch=-3.0)
    tracer.upload_scene(scene)
    bvh = tracer.render(cam, lib.render_params(w, h, 6, 3, 5))
    rays = tracer.stats().rays_traced
    assert ", true, 1, false>" in tracer.kernel_name()
    brute = tracer.render(cam, lib.render_params(w, h, 6, 3, 5, trace_mode=T.TRACE_BRUTE_FORCE))
    assert rays == tracer.stats().rays_traced and bvh[1].any()
    assert np.array_equal(bvh[0], brute[0]) and np.array_equal(bvh[1].view(np.uint32), brute[1].view(np.uint32))
    n = 600
    o = rng.uniform(-2.0, 2.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32) * rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32)
    got = tracer.intersect_rays(o, d)
    exp = oracle_intersect(scene, o, d)
    hit = exp["hit"].astype(bool)
    assert hit.sum() > n // 2 and np.array_equal(got["hit"], exp["hit"]) and np.array_equal(got["geom"][hit], exp["geom"][hit])
    assert np.array_equal(got["t"][hit].view(np.uint32), exp["t"][hit].view(np.uint32))
    assert np.array_equal(got["tri"][hit], exp["tri"][hit]) and np.array_equal(got["point"][hit].view(np.uint32), exp["point"][hit].view(np.uint32))
    os.environ["FF_NO_SCAN_PLANES"] = "1"
    try:
        tracer.reload_switches()  # (the switches are read at ff_create)
        tracer.upload_scene(scene)
        same = tracer.render(cam, lib.render_params(w, h, 6, 3, 5))
    finally:
        del os.environ["FF_NO_SCAN_PLANES"]
        tracer.reload_switches()
    assert np.array_equal(same[1].view(np.uint32), bvh[1].view(np.uint32))


def test_crowded_scene_equals_oracle(tracer):
    rng = np.random.default_rng(5)
    scene = fz.rand_scene(rng, small=True, crowd=45)
    assert len(scene) > 40
    w, h, cam = fz.rand_view(rng, max_w=40, max_h=30)
    p = lib.render_params(w, h, 4, 2, 3, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
    tracer.upload_scene(scene)
    rgb8, rad = tracer.render(cam, p)
    o_rgb8, o_rad = oracle_render(scene, cam, p, threads=16)
    assert np.array_equal(rgb8, o_rgb8) and np.array_equal(rad.view(np.uint32), o_rad.view(np.uint32))


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_scene_equals_oracle(tracer, seed):
    """Small random scenes through the CPU oracle (brute force, seconds) and the BVH kernel: bit-identical."""
    rng = np.random.default_rng(seed)
    scene = fz.rand_scene(rng, small=True)
    w, h, cam = fz.rand_view(rng, max_w=48, max_h=36)
    p = fz.rand_params(rng, w, h, T.TRACE_BVH)
    tracer.upload_scene(scene)
    rgb8, rad = tracer.render(cam, p)
    o_rgb8, o_rad = oracle_render(scene, cam, p, threads=16)
    assert np.array_equal(rgb8, o_rgb8)
    assert np.array_equal(rad.view(np.uint32), o_rad.view(np.uint32))


def _edge_scenes():
    cube = scenes.load_mesh("cube")
    light = lambda: scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=3.0)
    grey = lambda: scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.7, 0.7, 0.7))
    glass = lambda: scenes.make_bxdf(T.BXDF_GLASS, specular=(1, 1, 1), transmittance=(0.9, 0.9, 1.0), ior=1.5)
    out = {}
    out["planes_only"] = scenes.Scene().add_plane((0, 0, -3), (0, 0, 0), (8, 8, 8), grey()).add_plane((0, 3, 0), (90, 0, 0), (8, 8, 8), light()).finalize()
    out["mesh_only"] = scenes.Scene().add_mesh(cube, (0, 0, -2), (20, 30, 0), (2, 2, 2), light()).finalize()
    out["one_triangle"] = scenes.Scene().add_mesh(cube[:1] * np.float32(6.0), (0, 0, -3), (0, 0, 0), (1, 1, 1), light()).finalize()
    out["sphere_only"] = scenes.Scene().add_sphere(1.0, (0, 0, -3), (0, 0, 0), (1, 1, 1), light()).finalize()
    # the camera sits inside a big emitting sphere and inside a glass sphere inside it
    out["inside_spheres"] = (scenes.Scene().add_sphere(20.0, (0, 0, 0), (0, 0, 0), (1, 1, 1), light())
                             .add_sphere(3.5, (0, 0, 4), (0, 0, 0), (1, 1, 1), glass()).add_mesh(cube, (0, 0, -4), (0, 0, 0), (1, 1, 1), grey()).finalize())
    # far from the origin and strongly non-uniform scales
    out["far_and_stretched"] = (scenes.Scene().add_mesh(cube, (1000, 1000, -1005), (10, 20, 30), (8, 0.05, 3), grey())
                                .add_sphere(1.0, (1000, 1002, -1004), (0, 0, 45), (0.2, 3, 1), grey())
                                .add_plane((1000, 1004, -1004), (90, 0, 0), (30, 30, 30), light()).finalize())
    return out


@pytest.mark.parametrize("name", ["planes_only", "mesh_only", "one_triangle", "sphere_only", "inside_spheres", "far_and_stretched"])
def test_edge_scenes_equal_oracle(name):
    scene = _edge_scenes()[name]
    pos = (1000.0, 1000.5, -998.0) if name == "far_and_stretched" else (0.2, 0.3, 4.0)
    cam = scenes.posed_camera(48, 36, position=pos, yaw=-92.0, pitch=-3.0)
    for shade, bounces, spp in ((T.SHADE_NORMAL_DEBUG, 1, 1), (T.SHADE_DIFFUSE_PATH, 5, 3), (T.SHADE_DIFFUSE_PATH_SMOOTH, 3, 2)):
        p = lib.render_params(48, 36, bounces, spp, 42, T.TRACE_BVH, shade, T.GRID_FULL, 0)
        o_rgb8, o_rad = oracle_render(scene, cam, p, threads=16)
        for builder in (T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC):
            with lib.Tracer(0) as t:
                t.set_builder(builder)
                t.upload_scene(scene)
                for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
                    p.trace_mode = mode
                    rgb8, rad = t.render(cam, p)
                    assert np.array_equal(rgb8, o_rgb8), (name, shade, builder, mode)
                    assert np.array_equal(rad.view(np.uint32), o_rad.view(np.uint32)), (name, shade, builder, mode)


def _wall_rooms(rng):
    """A room of axis-aligned quads (rotations by multiples of 90 degrees about all three axes, non-uniform scales up to 1:8,
    off-centre), a small quad just below its ceiling, one oblique quad and a cube: what the wall table (ff_internal.h WallTable)
    screens in world space, next to what it leaves to the per-lane screens."""
    quarter = lambda: float(rng.choice([0.0, 90.0, 180.0, 270.0]))
    c = rng.uniform(-3.0, 3.0, 3)
    half = rng.uniform(1.0, 4.0, 3)
    grey = lambda: scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=tuple(float(v) for v in rng.uniform(0.3, 0.9, 3)))
    s = scenes.Scene()
    s.add_mesh(scenes.load_mesh("cube"), tuple(float(v) for v in c + rng.uniform(-0.5, 0.5, 3)), (10.0, 20.0, 30.0), (0.7, 0.7, 0.7), grey())
    walls = []
    # a plane's quad lies in its object xy plane: rotate it onto each pair of faces, with extra quarter turns about its normal
    faces = [((0, 0, -1), (0.0, 0.0)), ((0, 0, 1), (0.0, 0.0)), ((0, -1, 0), (90.0, 0.0)), ((0, 1, 0), (90.0, 0.0)), ((-1, 0, 0), (0.0, 90.0)), ((1, 0, 0), (0.0, 90.0))]
    for (n, (rx, ry)) in faces:
        if rng.random() < 0.15:
            continue  # an open side now and then
        axis = int(np.argmax(np.abs(n)))
        pos = c.copy()
        pos[axis] += n[axis] * half[axis]
        u, v = [a for a in range(3) if a != axis]
        # world extents 2 * half[u] x 2 * half[v]; which object axis carries which depends on the rotation
        rz = quarter()
        ext = {0: (half[0], half[1]), 1: (half[0], half[2]), 2: (half[2], half[1])}[{2: 0, 1: 1, 0: 2}[axis]]
        sx, sy = (2 * ext[0], 2 * ext[1]) if rz in (0.0, 180.0) else (2 * ext[1], 2 * ext[0])
        rot = (rx + (180.0 if rng.random() < 0.3 and rx else 0.0), ry + (180.0 if rng.random() < 0.3 and ry else 0.0), rz)
        s.add_plane(tuple(float(p) for p in pos), rot, (float(sx), float(sy), float(rng.uniform(0.5, 4.0))), grey())
        walls.append((axis, pos.copy()))
    top = c.copy()
    top[1] += half[1] - 0.01
    s.add_plane(tuple(float(p) for p in top), (90.0, 0.0, quarter()), (float(half[0]), float(half[2]), 1.0),
                scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0))
    s.add_plane(tuple(float(v) for v in c + rng.uniform(-0.5, 0.5, 3)), (30.0, 45.0, 10.0), (1.5, 2.5, 1.0), grey())
    return s.finalize(), c, half


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_wall_table_edges_corners_and_grazing_rays(tracer, seed):
    """Axis-aligned walls are screened in world space by a wave-uniform loop (csrc/ff_kernels.hip screen_walls); everything inside
    its margins goes to the exact reference test.  Rays aimed at the walls' edges and corners (offsets from 0 to 1e-3 of the
    room), rays almost parallel to a wall, origins on and next to wall planes and far outside: ff_intersect_rays through the BVH
    kernel equals the brute-force kernel and the oracle in every field."""
    from oracle_lib import oracle_intersect
    rng = np.random.default_rng(4000 + seed)
    scene, c, half = _wall_rooms(rng)
    n = 1500
    o = (c + rng.uniform(-0.95, 0.95, (n, 3)) * half).astype(np.float64)
    kind = rng.integers(0, 6, n)
    # targets on edges / corners of the room's box: two or three coordinates on a face
    tgt = c + rng.uniform(-1.0, 1.0, (n, 3)) * half
    for i in range(n):
        on = rng.choice(3, size=2 if kind[i] % 2 == 0 else 3, replace=False)
        for a in on:
            tgt[i, a] = c[a] + (1.0 if rng.random() < 0.5 else -1.0) * half[a]
        tgt[i] += rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-5, 1e-4, -1e-3]) * half * rng.uniform(-1, 1, 3)
    d = tgt - o
    graze = kind == 4
    # almost parallel to a wall: one component tiny
    ax = rng.integers(0, 3, n)
    d[graze, ax[graze]] = rng.choice([0.0, 1e-9, -1e-8, 1e-7, -1e-6, 1e-5, 1e-4], graze.sum()) * np.abs(d[graze]).max(axis=1)
    # origins on a wall plane (and a hair off it), and far outside the room
    onw = kind == 5
    o[onw, ax[onw]] = (c[ax[onw]] + np.where(rng.random(onw.sum()) < 0.5, 1.0, -1.0) * half[ax[onw]]
                       + rng.choice([0.0, 1e-6, -1e-6, 1e-4, -1e-4], onw.sum()))
    far = rng.random(n) < 0.1
    o[far] = c + rng.uniform(-6.0, 6.0, (far.sum(), 3)) * half
    d[far] = tgt[far] - o[far]
    d *= rng.uniform(0.1, 3.0, (n, 1))  # kernel.cu:138 normalises in object space; world directions of any length
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    tracer.upload_scene(scene)
    bvh = tracer.intersect_rays(o32, d32, T.TRACE_BVH)
    brute = tracer.intersect_rays(o32, d32, T.TRACE_BRUTE_FORCE)
    exp = oracle_intersect(scene, o32, d32)
    hit = exp["hit"].astype(bool)
    assert 0.3 * n < hit.sum()
    for got in (bvh, brute):
        assert np.array_equal(got["hit"], exp["hit"]) and np.array_equal(got["geom"][hit], exp["geom"][hit])
        assert np.array_equal(got["t"][hit].view(np.uint32), exp["t"][hit].view(np.uint32))
        assert np.array_equal(got["tri"][hit], exp["tri"][hit])
        assert np.array_equal(got["point"][hit].view(np.uint32), exp["point"][hit].view(np.uint32))
        assert np.array_equal(got["normal"][hit].view(np.uint32), exp["normal"][hit].view(np.uint32))
    # and as frames: a camera inside the room looking into a corner, path mode, against brute force
    cam = scenes.posed_camera(96, 72, position=tuple(float(v) for v in c + 0.3 * half), yaw=-135.0, pitch=-35.0)
    p = lib.render_params(96, 72, 8, 4, 11)
    a = tracer.render(cam, p)
    rays = tracer.stats().rays_traced
    p.trace_mode = T.TRACE_BRUTE_FORCE
    b = tracer.render(cam, p)
    assert rays == tracer.stats().rays_traced
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
