"""Pins of the CPU oracle: the known answers the survey recorded from the reference's own device functions
(SURVEY.md §8c), the committed golden frames, the Random123 known answers for Philox, and properties of the
build-defined sampling pieces."""
import ctypes as C
import os

import numpy as np
import pytest

from cases import CASES, build_case
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T
from oracle_lib import farr, oracle_render

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz")


def test_cube_256_known_answer():
    """BASELINE config #1: cube.obj 256x256, 1 bounce, 1 spp -> 5329 (=73^2) lit pixels, all (0,0,51)."""
    scene, cam, params = build_case("c1_cube_256")
    rgb8, rad = oracle_render(scene, cam, params)
    px = rgb8.reshape(-1, 3)
    lit = px[px.any(axis=1)]
    assert len(lit) == 5329
    assert (lit == np.array([0, 0, 51], dtype=np.uint8)).all()


def test_rocketman_800_known_answer():
    """The shipped scene's first frame (kernel.cu:229-258, 800x800): 52441 x (0,0,51) + 1 x (34,129,216)."""
    scene, cam, params = build_case("ref_rocketman_800_default")
    rgb8, _ = oracle_render(scene, cam, params)
    px = rgb8.reshape(-1, 3)
    u, c = np.unique(px[px.any(axis=1)], axis=0, return_counts=True)
    assert {tuple(int(v) for v in k): int(n) for k, n in zip(u, c)} == {(0, 0, 51): 52441, (34, 129, 216): 1}


def test_wahoo_800_known_answer():
    """wahoo.obj in the reference scene, 800x800: 92595 lit pixels, 473 distinct colours counting black."""
    scene = scenes.reference_scene(scenes.load_mesh("wahoo"))
    cam = scenes.default_camera(800, 800)
    params = lib.render_params(800, 800, 1, 1, shade_mode=T.SHADE_NORMAL_DEBUG)
    rgb8, _ = oracle_render(scene, cam, params)
    px = rgb8.reshape(-1, 3)
    assert int(px.any(axis=1).sum()) == 92595
    assert len(np.unique(px, axis=0)) == 473


@pytest.mark.parametrize("name", [n for n in CASES if not n.endswith("800_default")])
def test_golden_frames_reproduce(name):
    """The committed fixtures are what the oracle produces today (guards the fixtures against oracle drift)."""
    golden = np.load(GOLDEN)
    scene, cam, params = build_case(name)
    rgb8, rad = oracle_render(scene, cam, params)
    assert np.array_equal(rgb8, golden[name + "/rgb8"])
    assert np.array_equal(rad.view(np.uint32), golden[name + "/radiance"].view(np.uint32))


def test_floor_grid_leaves_last_rows_untraced():
    """kernel.cu:308-309 floor division: 150 rows -> only 144 traced; 200 columns -> 192 traced."""
    golden = np.load(GOLDEN)
    img = golden["ref_sphere_200x150_floorgrid/rgb8"]
    assert not img[144:].any() and not img[:, 192:].any() and img.any()


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox2x32-10."""
    kat = [((0x00000000, 0x00000000), 0x00000000, (0xff1dae59, 0x6cd10df2)),
           ((0xffffffff, 0xffffffff), 0xffffffff, (0x2c3f628b, 0xab4fd7ad)),
           ((0x243f6a88, 0x85a308d3), 0x13198a2e, (0xdd7ce038, 0xf62a4c12))]
    for (c0, c1), key, (e0, e1) in kat:
        o0, o1 = C.c_uint32(), C.c_uint32()
        oracle.orc_philox2x32_10(c0, c1, key, C.byref(o0), C.byref(o1))
        assert (o0.value, o1.value) == (e0, e1)


def test_cosine_sample_is_unit_and_accurate(oracle):
    rng = np.random.default_rng(0)
    out = np.zeros(3, dtype=np.float32)
    for _ in range(2000):
        k24 = int(rng.integers(0, 1 << 24))
        u1 = np.float32(rng.integers(0, 1 << 24) * 2.0 ** -24)
        oracle.orc_cosine_sample_hemisphere(C.c_float(u1), k24, out.ctypes.data_as(C.POINTER(C.c_float)))
        theta = 2.0 * np.pi * k24 * 2.0 ** -24
        r = np.sqrt(float(u1))
        ref = np.array([r * np.cos(theta), r * np.sin(theta), np.sqrt(max(0.0, 1.0 - float(u1)))])
        assert np.abs(out - ref).max() < 2e-7 * 4
        assert abs(float(np.dot(out.astype(np.float64), out.astype(np.float64))) - 1.0) < 1e-6
    # octant boundaries are exact
    for oct_ in range(8):
        oracle.orc_cosine_sample_hemisphere(C.c_float(1.0 - 2.0 ** -24), oct_ << 21, out.ctypes.data_as(C.POINTER(C.c_float)))
        assert abs(np.hypot(out[0], out[1]) - np.sqrt(1.0 - 2.0 ** -24)) < 1e-6


def test_onb_is_orthonormal(oracle):
    rng = np.random.default_rng(1)
    t = np.zeros(3, dtype=np.float32)
    b = np.zeros(3, dtype=np.float32)
    for _ in range(500):
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        oracle.orc_onb(farr(n)[1], t.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)))
        m = np.stack([t, b, n]).astype(np.float64)
        assert np.abs(m @ m.T - np.eye(3)).max() < 1e-5


def test_path_integrator_reduces_to_reference_semantics():
    """With bounces=1 the integrator returns emitted radiance only (primary hit on the emitter, else 0), and every
    pixel of the un-lit reference scene is black in path mode."""
    scene = scenes.reference_scene(scenes.load_mesh("cube"))
    cam = scenes.default_camera(64, 64)
    params = lib.render_params(64, 64, 4, 2)
    _, rad = oracle_render(scene, cam, params)
    assert not rad.any()  # no emitter in kernel.cu:229-258
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(64, 64, position=(0, 1.5, 2.0), yaw=-90.0, pitch=40.0)  # looking up at the light
    params = lib.render_params(64, 64, 1, 1)
    _, rad = oracle_render(scene, cam, params)
    vals = np.unique(rad)
    assert set(vals.tolist()) <= {0.0, 2.0} and 2.0 in vals  # Le * intensity = (1,1,1)*2 (kernel.cu:243-244)


def test_ray_count_bound_and_counters():
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(32, 24, position=(0, 0, 2.4), yaw=-90.0, pitch=0.0)
    params = lib.render_params(32, 24, 5, 3)
    _, _, ctr = oracle_render(scene, cam, params, want_counters=True)
    assert 32 * 24 * 3 <= ctr.rays <= 32 * 24 * 3 * 5
    assert ctr.tri_tests == ctr.rays * scene.triangle_count
    assert ctr.plane_tests == ctr.rays * 6
    # a window render equals the corresponding crop of the full render
    _, full = oracle_render(scene, cam, params)
    _, win = oracle_render(scene, cam, params, window=(5, 7, 11, 9))
    assert np.array_equal(win.view(np.uint32), full[7:16, 5:16].view(np.uint32))
