#!/usr/bin/env python3
"""Where a 1-spp frame's time goes: kernel time of C2 (camera inside) at 1 spp over frame sizes and bounce counts; the intercept
of the size series is what a launch costs before its first and after its last ray (scene staging, ramp, the last paths)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpupathtracer_amd import lib, scenes
scene = scenes.cornell_wahoo_scene()
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    for bounces in (8, 1):
        for w, h in ((64, 64), (256, 256), (512, 512), (800, 800), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
            cam = scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
            ks = []
            for i in range(8):
                t.render(cam, lib.render_params(w, h, bounces, 1, 100 + i), want_rgb8=False, want_radiance=False)
                ks.append(t.stats().kernel_ms)
            st = t.stats()
            print(f"{bounces} bounces {w}x{h}: kernel min {min(ks[2:]):.3f} ms median {sorted(ks[2:])[3]:.3f}  rays {st.rays_traced}  {st.rays_traced / min(ks[2:]) / 1e3:.0f} Mrays/s", flush=True)
