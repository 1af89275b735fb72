"""GPU tests (`-m gpu`) that pin what BASELINE.json's configs[2..4] add over the headline config and round 1 never exercised:

  * the sample-block rule beyond 1 024 spp (`block_spp = 64 * ceil(spp / 1024)`, oracle/ff_oracle.c orc_render vs
    csrc/ff_api.cpp render_local) against the CPU oracle, radiance bits equal;
  * C5's shape: 3840x2160 (16-bit packed pixel coordinates, 8.3 M work items per block), 16 bounces;
  * C4 in path mode: 8 diffuse bounces on the 983 040-triangle sphere, where secondary rays leave the tessellated surface
    with the 1e-4 offset — BVH (all three builders) against the brute-force loop and against the oracle.
"""
import numpy as np
import pytest

from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T
from oracle_lib import oracle_render

pytestmark = pytest.mark.gpu


def _inside(w, h):
    return scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.fixture(scope="module")
def c4_scene():
    scene = scenes.sphere_stress_scene(5)
    assert scene.triangle_count == 983040
    return scene


@pytest.mark.parametrize("spp,bounces,size", [(1100, 3, (6, 4)), (2048, 2, (5, 3)), (4096, 3, (6, 4)), (4096 + 37, 2, (4, 3))])
def test_sample_blocks_beyond_1024_spp_match_the_oracle(tracer, spp, bounces, size):
    """block_spp = 128 (1 100, 2 048 spp), 256 (4 096 spp: C3's and C5's sample count) and 320 with a partial last block:
    both trace modes produce the oracle's radiance bits, so the order in which samples and blocks are added is the oracle's."""
    w, h = size
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(w, h)
    tracer.upload_scene(scene)
    params = lib.render_params(w, h, bounces, spp, seed=4242)
    exp = oracle_render(scene, cam, params, threads=16)
    assert exp[1].max() > 0
    for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
        got = tracer.render(cam, lib.render_params(w, h, bounces, spp, seed=4242, trace_mode=mode))
        assert _same(got, exp), (spp, mode)
    # the same frame split into launches of a few blocks each (the accumulate-across-launches path at these block sizes)
    block = 64 * ((spp + 1023) // 1024)
    chunked = tracer.render(cam, lib.render_params(w, h, bounces, spp, seed=4242, spp_per_launch=3 * block))
    assert tracer.stats().kernel_launches == -(-(-(-spp // block)) // 3)
    assert _same(chunked, exp)


def test_c5_shape_4k_16_bounces_bvh_equals_brute_force(tracer):
    """BASELINE configs[4] geometry: 3840x2160, 16 bounces (1 spp so that the brute-force loop finishes in a second): same
    bits, same ray count, and ray counts inside the bounds of the integrator."""
    w, h = 3840, 2160
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(w, h)
    tracer.upload_scene(scene)
    out = {}
    for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
        frame = tracer.render(cam, lib.render_params(w, h, 16, 1, seed=99, trace_mode=mode))
        out[mode] = (frame, tracer.stats().rays_traced)
    assert out[T.TRACE_BVH][1] == out[T.TRACE_BRUTE_FORCE][1]
    assert w * h <= out[T.TRACE_BVH][1] <= w * h * 16
    assert _same(out[T.TRACE_BVH][0], out[T.TRACE_BRUTE_FORCE][0])
    rad = out[T.TRACE_BVH][0][1]
    assert rad.max() > 0 and rad[:, 3000:].max() > 0 and rad[2000:].max() > 0  # columns / rows beyond 2 048 were traced
    # the 8-bounce frame of the same seed differs: paths really continue past bounce 8
    rays8 = None
    shorter = tracer.render(cam, lib.render_params(w, h, 8, 1, seed=99))
    rays8 = tracer.stats().rays_traced
    assert rays8 < out[T.TRACE_BVH][1] and not np.array_equal(shorter[1], rad)


def test_c5_window_of_the_4k_frame_matches_the_oracle(tracer):
    """A window in the far corner of the 4K frame (x >= 3 800, y >= 2 100), 16 bounces, 4 096 + 4 spp through ff_render_tile
    against the oracle: the C5 parameters end to end on pixels whose coordinates need more than 11 bits."""
    w, h = 3840, 2160
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(w, h)
    tracer.upload_scene(scene)
    x0, y0, tw, th = 3805, 2103, 5, 3
    params = lib.render_params(w, h, 16, 4100, seed=5)
    got = tracer.render_tile(cam, params, x0, y0, tw, th)
    exp = oracle_render(scene, cam, params, window=(x0, y0, tw, th), threads=16)
    assert exp[1].max() > 0 and _same(got, exp)


def test_16_bounce_frame_matches_the_oracle(tracer):
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(32, 18)
    tracer.upload_scene(scene)
    params = lib.render_params(32, 18, 16, 4, seed=16)
    exp = oracle_render(scene, cam, params, threads=16)
    for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
        got = tracer.render(cam, lib.render_params(32, 18, 16, 4, seed=16, trace_mode=mode))
        assert _same(got, exp), mode
    tracer.set_collect_stats(True)
    tracer.render(cam, params)
    tracer.set_collect_stats(False)
    assert tracer.stats().rays_traced > 32 * 18 * 4 * 4  # open-front box: paths are long on average


@pytest.mark.parametrize("builder", [T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH, T.BUILD_GPU_PLOC])
def test_c4_path_mode_bvh_equals_brute_force(c4_scene, builder):
    """BASELINE configs[3] in path mode: 8 bounces x 2 spp on a 64x36 tile of the 1080p frame that straddles the sphere's
    silhouette; every tree kind gives the brute-force loop's bits and ray count."""
    cam = _inside(1920, 1080)
    x0, y0, w, h = 1180, 300, 64, 36
    with lib.Tracer(0) as t:
        t.set_builder(builder)
        t.upload_scene(c4_scene)
        bvh = t.render_tile(cam, lib.render_params(1920, 1080, 8, 2, seed=7), x0, y0, w, h)
        rays_bvh = t.stats().rays_traced
        brute = t.render_tile(cam, lib.render_params(1920, 1080, 8, 2, seed=7, trace_mode=T.TRACE_BRUTE_FORCE), x0, y0, w, h)
        rays_brute = t.stats().rays_traced
    assert rays_bvh == rays_brute and rays_bvh > w * h * 2 * 2
    assert _same(bvh, brute)
    assert bvh[1].max() > 0


def test_c4_path_mode_matches_the_oracle_crop(tracer, c4_scene):
    """The same config against the CPU oracle (brute force over 983 040 triangles) on a 14x8 crop across the sphere's
    silhouette (on the sphere itself most paths leave through the open front after two segments and stay black)."""
    cam = _inside(1920, 1080)
    x0, y0, w, h = 1226, 324, 14, 8
    tracer.upload_scene(c4_scene)
    params = lib.render_params(1920, 1080, 8, 6, seed=11)
    got = tracer.render_tile(cam, params, x0, y0, w, h)
    exp = oracle_render(c4_scene, cam, params, window=(x0, y0, w, h), threads=16)
    assert _same(got, exp) and exp[1].max() > 0
    # smooth shading reads the vertex normals of the subdivided mesh through the same tree
    smooth = lib.render_params(1920, 1080, 8, 6, seed=11, shade_mode=T.SHADE_DIFFUSE_PATH_SMOOTH)
    got = tracer.render_tile(cam, smooth, x0, y0, w, h)
    exp = oracle_render(c4_scene, cam, smooth, window=(x0, y0, w, h), threads=16)
    assert _same(got, exp) and exp[1].max() > 0
