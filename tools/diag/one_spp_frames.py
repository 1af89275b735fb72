#!/usr/bin/env python3
"""Twelve 1-spp path-traced 1080p frames of C2 from one camera (for a kernel trace: tools/diag/frame_gaps.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpupathtracer_amd import lib, scenes
scene = scenes.cornell_wahoo_scene()
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    for i in range(12):
        t.render(cam, lib.render_params(1920, 1080, 8, 1, 100 + i), want_rgb8=False, want_radiance=False)
    st = t.stats()
    print(f"last frame: kernel {st.kernel_ms:.3f} ms, {st.rays_traced / st.kernel_ms / 1e3:.0f} Mrays/s")
