#!/usr/bin/env python3
"""Frame time of what the reference's viewer renders per frame: primary hit + abs(normal) (kernel.cu:178-184), 1 spp, and of
1-spp path-traced frames (progressive refinement), through the host-buffer API the viewer-less tests use."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

def run(name, scene, cam, p, n=50):
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        for _ in range(3):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
        ks = []
        t0 = time.perf_counter()
        for _ in range(n):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            ks.append(t.stats().kernel_ms)
        wall = (time.perf_counter() - t0) * 1e3 / n
        print(f"{name}: kernel {min(ks):.3f} ms (median {sorted(ks)[n // 2]:.3f}), call {wall:.3f} ms per frame, outputs left on the device")

wahoo = scenes.load_mesh("wahoo")
rocket = scenes.load_mesh("rocketman")
for w, h in ((800, 800), (1920, 1080)):
    dbg = lib.render_params(w, h, 1, 1, 1234, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)
    run(f"reference scene (rocketman), default camera, {w}x{h}, normal shade", scenes.reference_scene(rocket), scenes.default_camera(w, h), dbg)
    run(f"reference scene (wahoo), oblique camera, {w}x{h}, normal shade", scenes.reference_scene(wahoo),
        scenes.posed_camera(w, h, position=(7.0, 3.0, 9.0), yaw=-128.0, pitch=-14.0), dbg)
    cam = scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    run(f"C2 box, camera inside, {w}x{h}, normal shade", scenes.cornell_wahoo_scene(), cam, dbg)
    run(f"C2 box, camera inside, {w}x{h}, path traced 8 bounces 1 spp", scenes.cornell_wahoo_scene(), cam, lib.render_params(w, h, 8, 1))
