#!/usr/bin/env python3
"""Render rate of a scene preset as a function of spp (work-item granularity: one item = one pixel x one sample block).
Usage: spp_scaling.py c2|c3|c4|c5 spp [spp ...]   (c5 = the C2 scene at 3840x2160 with 16 bounces)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes
which = sys.argv[1]
W, H, B = (3840, 2160, 16) if which == "c5" else (1920, 1080, 8)
cam = scenes.posed_camera(W, H, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
if which == "c3":
    scene = scenes.blooper_scene()
    cam = scenes.posed_camera(1920, 1080, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)
elif which == "c4":
    scene = scenes.sphere_stress_scene(5)
else:
    scene = scenes.cornell_wahoo_scene()
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    for spp in [int(a) for a in sys.argv[2:]]:
        p = lib.render_params(W, H, B, spp)
        t.render(cam, p, want_rgb8=False, want_radiance=False)
        t.render(cam, p, want_rgb8=False, want_radiance=False)
        st = t.stats()
        print(f"{which} spp {spp:5d}: {st.rays_traced / st.kernel_ms / 1e3:8.0f} Mrays/s  kernel {st.kernel_ms:9.2f} ms", flush=True)
