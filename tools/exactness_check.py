#!/usr/bin/env python3
"""BVH kernel vs brute-force kernel on the full C2 frame at several spp: bit-exact radiance and equal ray counts expected."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpupathtracer_amd import lib, scenes, types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene = scenes.cornell_wahoo_scene()
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    out = {}
    for name, mode in (("bvh", T.TRACE_BVH), ("brute", T.TRACE_BRUTE_FORCE)):
        _, rad = t.render(cam, lib.render_params(1920, 1080, 8, spp, trace_mode=mode), want_rgb8=False)
        out[name] = (rad, t.stats().rays_traced, t.stats().kernel_ms)
        print(name, "rays", out[name][1], "kernel_ms", round(out[name][2], 1))
a, b = out["bvh"][0], out["brute"][0]
diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
print("pixels differing:", int(diff.sum()), "ray count equal:", out["bvh"][1] == out["brute"][1])
ys, xs = np.nonzero(diff)
for y, x in list(zip(ys, xs))[:10]:
    print("  pixel", x, y, "bvh", a[y, x], "brute", b[y, x])
