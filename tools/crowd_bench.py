#!/usr/bin/env python3
"""Render rate against the number of geometries (kernel.cu:133 loops over all of them): the C2 box (wahoo + cube + six
planes) with `crowd` extra small cubes / spheres / quads scattered through the room, 1920x1080, 8 bounces, 16 spp, camera
inside, BVH kernel.  Up to 32 geometries the records sit in LDS and every query screens all of them; beyond, a query walks
the tree over the geometries' world boxes.  Usage: crowd_bench.py [crowd ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuzz_parity as fz
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

crowds = [int(a) for a in sys.argv[1:]] or [0, 8, 20, 40, 100, 250, 500]
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
p = lib.render_params(1920, 1080, 8, 16, 7)
cube = scenes.load_mesh("cube")
with lib.Tracer(0) as t:
    for crowd in crowds:
        rng = np.random.default_rng(1)
        s = scenes.Scene()
        s.add_mesh(scenes.load_mesh("wahoo"), (0, -2.4, 0), (0, 0, 0), (0.28, 0.28, 0.28), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0)))
        s.add_mesh(cube, (1.5, -2.0, 1.0), (0, 0, 0), (1, 1, 1), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75)))
        for _ in range(crowd):
            k = int(rng.integers(0, 3))
            pos, rot = tuple(rng.uniform(-2.2, 2.2, 3)), tuple(rng.uniform(-180, 180, 3))
            bx = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=tuple(float(v) for v in rng.uniform(0.3, 0.9, 3)))
            if k == 0:
                s.add_mesh(cube, pos, rot, tuple(float(v) for v in rng.uniform(0.1, 0.3, 3)), bx)
            elif k == 1:
                s.add_sphere(float(rng.uniform(0.08, 0.2)), pos, rot, (1, 1, 1), bx)
            else:
                s.add_plane(pos, rot, tuple(float(v) for v in rng.uniform(0.15, 0.5, 3)), bx)
        scene = scenes._box(s).finalize()
        t.upload_scene(scene)
        ks = []
        for _ in range(3):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            ks.append(t.stats().kernel_ms)
        st = t.stats()
        name = t.kernel_name() if hasattr(t, "kernel_name") and hasattr(t._lib, "ff_debug_kernel_name") else "?"
        print(f"{len(scene):4d} geometries ({scene.triangle_count} triangles): {st.rays_traced / min(ks) / 1e3:7.0f} Mrays/s, {min(ks):7.2f} ms, "
              f"{st.rays_traced / (1920 * 1080 * 16):.2f} rays per path, kernel {name}", flush=True)
