"""Randomised parity on the GPU (tools/fuzz_parity.py): random scenes of meshes, planes and spheres under random
rotations and non-uniform scales, random diffuse / mirror / glass / emitter surfaces, random cameras and seeds."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity as fz  # noqa: E402
from gpupathtracer_amd import lib  # noqa: E402
from gpupathtracer_amd import types as T  # noqa: E402
from oracle_lib import oracle_render  # noqa: E402

pytestmark = pytest.mark.gpu


def test_bvh_equals_brute_force_on_random_scenes():
    """Both tree kinds against the reference loop (kernel.cu:133-155): radiance bits, rgb8 bytes and ray counts."""
    bad, rays = fz.run(40, seed=2026, verbose=False)
    assert bad == 0 and rays > 500000


def test_crowded_scenes_equal_brute_force():
    """Scenes of 35-120 geometries: the BVH kernel works through the records in chunks of 32."""
    bad, rays = fz.run(12, seed=77, verbose=False, crowd_fraction=1.0)
    assert bad == 0 and rays > 100000


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
    with pytest.raises(lib.FireflyError) as e:
        tracer.upload_scene(fz.rand_scene(rng, small=True, crowd=140))
        tracer.render(cam, p)
    assert e.value.status == T.FF_ERR_UNSUPPORTED


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
