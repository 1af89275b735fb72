#!/usr/bin/env python3
"""PLOC search radius (FF_PLOC_RADIUS, csrc/ff_build.hip) against build time and trace rate, with the host SAH tree and the LBVH
as yardsticks.  Usage: ploc_radius_sweep.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes, types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inside = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
for name, scene in (("c2", scenes.cornell_wahoo_scene()), ("c4", scenes.sphere_stress_scene(5))):
    p = lib.render_params(1920, 1080, 8, spp if name == "c2" else max(spp // 2, 1))
    for label, builder, radius in [("sah", T.BUILD_HOST_SAH, None), ("lbvh", T.BUILD_GPU_LBVH, None)] + [(f"ploc r={r}", T.BUILD_GPU_PLOC, r) for r in (8, 16, 32, 64, 128)]:
        if radius is None:
            os.environ.pop("FF_PLOC_RADIUS", None)
        else:
            os.environ["FF_PLOC_RADIUS"] = str(radius)
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            t.upload_scene(scene)  # (warm: allocations, code objects)
            bs = t.build_stats()
            ks = []
            for _ in range(3):
                t.render(inside, p, want_rgb8=False, want_radiance=False)
                ks.append(t.stats().kernel_ms)
            rays = t.stats().rays_traced
        print(f"{name} {label:12s} | build {bs.build_ms:8.2f} ms | {min(ks):8.2f} ms {rays / min(ks) / 1e3:7.0f} Mrays/s", flush=True)
