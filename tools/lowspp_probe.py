#!/usr/bin/env python3
"""Low-spp frames (the viewer's regime: kernel.cu:266,342 renders 1 spp per frame): kernel time of a 1080p C2 frame as a
function of spp and of the chunk a wave takes from the work queue per atomic (FF_QUEUE_CHUNK items; default: 256 samples of work),
best of N frames each.
Usage: lowspp_probe.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
scene = scenes.cornell_wahoo_scene()
for ipf in (None, "8", "16", "32", "64", "128", "256", "512", "1024"):
    if ipf is None:
        os.environ.pop("FF_QUEUE_CHUNK", None)
    else:
        os.environ["FF_QUEUE_CHUNK"] = ipf
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        row = []
        for spp in (1, 2, 4, 8, 16):
            p = lib.render_params(1920, 1080, 8, spp)
            ks = []
            for _ in range(frames):
                t.render(cam, p, want_rgb8=False, want_radiance=False)
                ks.append(t.stats().kernel_ms)
            rays = t.stats().rays_traced
            row.append(f"spp {spp}: {min(ks):6.2f} ms {rays / min(ks) / 1e3:6.0f} Mrays/s")
        p = lib.render_params(1920, 1080, 1, 1, 1234, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)
        ks = []
        for _ in range(frames):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            ks.append(t.stats().kernel_ms)
        row.append(f"normal shade: {min(ks):6.3f} ms")
        print(f"queue chunk {ipf or 'default':>7s} | " + " | ".join(row), flush=True)
