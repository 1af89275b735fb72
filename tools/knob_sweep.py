#!/usr/bin/env python3
"""Same-box sweep of the BVH kernel's scheduling knobs (environment variables read at ff_create) on the headline frame at
reduced spp: Mrays/s by kernel time, best of N frames.  Usage: knob_sweep.py SPP "NAME=v1,v2,..." ["NAME2=..."]  (cartesian)"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes

spp = int(sys.argv[1])
axes = []
for a in sys.argv[2:]:
    name, vals = a.split("=")
    axes.append([(name, v) for v in vals.split(",")])
which = os.environ.get("FF_SWEEP_SCENE", "c2")
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
scene = scenes.sphere_stress_scene(5) if which == "c4" else scenes.cornell_wahoo_scene()
p = lib.render_params(1920, 1080, 8, spp)
for combo in itertools.product(*axes):
    for name, v in combo:
        if v == "-":
            os.environ.pop(name, None)
        else:
            os.environ[name] = v
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        ks = []
        for _ in range(3):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            ks.append(t.stats().kernel_ms)
        rays = t.stats().rays_traced
    print(" ".join(f"{n}={v}" for n, v in combo), f"| {min(ks):8.2f} ms {rays / min(ks) / 1e3:7.0f} Mrays/s", flush=True)
