#!/usr/bin/env python3
"""The 4-wide collapse rule A/B on one box: optimal roles (least summed node area, csrc/ff_build.hip) against the fixed rule of
rounds 1-2 (FF_COLLAPSE_PARITY=1: every binary node at even depth, slots = grandchildren), per scene and builder: 4-wide nodes,
node visits / triangle tests per ray, frame time.  Usage: collapse_ab.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes, types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inside = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
cases = {"c2": (scenes.cornell_wahoo_scene, inside), "c3": (scenes.blooper_scene, scenes.posed_camera(1920, 1080, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)),
         "c4": (lambda: scenes.sphere_stress_scene(5), inside)}
p = lib.render_params(1920, 1080, 8, spp)
for name, (make, cam) in cases.items():
    scene = make()
    for bname, builder in (("sah", T.BUILD_HOST_SAH), ("lbvh", T.BUILD_GPU_LBVH), ("ploc", T.BUILD_GPU_PLOC)):
        for rule in ("parity", "optimal"):
            if rule == "parity":
                os.environ["FF_COLLAPSE_PARITY"] = "1"
            else:
                os.environ.pop("FF_COLLAPSE_PARITY", None)
            with lib.Tracer(0) as t:
                t.set_builder(builder)
                t.upload_scene(scene)
                bs = t.build_stats()
                t.set_collect_stats(True)
                t.render(cam, p, want_rgb8=False, want_radiance=False)
                st = t.stats()
                t.set_collect_stats(False)
                ks = []
                for _ in range(3):
                    t.render(cam, p, want_rgb8=False, want_radiance=False)
                    ks.append(t.stats().kernel_ms)
            print(f"{name} {bname:4s} {rule:7s} | nodes4 {st.scene_bytes_nodes // 112:7d}  build {bs.build_ms:8.2f} ms | visits/ray {st.nodes_visited / st.rays_traced:6.3f} "
                  f"tris/ray {st.tris_tested / st.rays_traced:6.3f} | {min(ks):8.2f} ms {st.rays_traced / min(ks) / 1e3:7.0f} Mrays/s", flush=True)
