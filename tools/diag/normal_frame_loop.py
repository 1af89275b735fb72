"""200 frames of the reference's own workload (primary hit + abs(normal), 1 spp, shipped scene, default camera, 1080p) for
`rocprofv3 --kernel-trace --stats -- python3 tools/diag/normal_frame_loop.py`: trace kernel 85.6 us, combine 6.2 us, counter copy 3.8 us,
clear 2.6 us per frame on one MI355X (round 2)."""
import os, sys
sys.path.insert(0, os.getcwd())
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T
w, h = 1920, 1080
p = lib.render_params(w, h, 1, 1, 1234, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)
with lib.Tracer(0) as t:
    t.upload_scene(scenes.reference_scene(scenes.load_mesh("rocketman")))
    cam = scenes.default_camera(w, h)
    for _ in range(200):
        t.render(cam, p, want_rgb8=False, want_radiance=False)
    print("kernel_ms", t.stats().kernel_ms)
