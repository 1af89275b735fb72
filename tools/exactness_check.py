#!/usr/bin/env python3
"""BVH kernel (host SAH trees and device LBVH trees) vs brute-force kernel on the full C2 frame: bit-exact radiance and
equal ray counts expected."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpupathtracer_amd import lib, scenes, types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene = scenes.cornell_wahoo_scene()
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
out = {}
for name, mode, builder in (("bvh", T.TRACE_BVH, T.BUILD_HOST_SAH), ("lbvh", T.TRACE_BVH, T.BUILD_GPU_LBVH), ("brute", T.TRACE_BRUTE_FORCE, T.BUILD_HOST_SAH)):
    with lib.Tracer(0) as t:
        t.set_builder(builder)
        t.upload_scene(scene)
        _, rad = t.render(cam, lib.render_params(1920, 1080, 8, spp, trace_mode=mode), want_rgb8=False)
        out[name] = (rad, t.stats().rays_traced, t.stats().kernel_ms)
        print(name, "rays", out[name][1], "kernel_ms", round(out[name][2], 1), flush=True)
l = out["lbvh"][0]
print("lbvh vs brute: pixels differing:", int((l.view(np.uint32) != out["brute"][0].view(np.uint32)).any(axis=2).sum()),
      "ray count equal:", out["lbvh"][1] == out["brute"][1])
a, b = out["bvh"][0], out["brute"][0]
diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
print("pixels differing:", int(diff.sum()), "ray count equal:", out["bvh"][1] == out["brute"][1])
ys, xs = np.nonzero(diff)
for y, x in list(zip(ys, xs))[:10]:
    print("  pixel", x, y, "bvh", a[y, x], "brute", b[y, x])
