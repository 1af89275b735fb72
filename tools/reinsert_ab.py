#!/usr/bin/env python3
"""Device builders with 0..N parallel reinsertion passes (FF_GPU_REINSERT) against the host tree: build time, node visits / triangle
tests per ray, trace rate.  Usage: reinsert_ab.py [spp] [c2,c3,c4] [passes,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FF_NO_PRIMARY_CACHE", "1")
from gpupathtracer_amd import lib, scenes, types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2", "c3"]
counts = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 4, 8]
inside = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
for name, scene in [(n, s) for n, s in (("c2", scenes.cornell_wahoo_scene()), ("c3", scenes.blooper_scene()), ("c4", scenes.sphere_stress_scene(5))) if n in which]:
    cam = scenes.posed_camera(1920, 1080, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0) if name == "c3" else inside
    p = lib.render_params(1920, 1080, 8, spp if name != "c4" else max(spp // 2, 1))
    cases = [("sah", T.BUILD_HOST_SAH, 0)]
    for b, bn in ((T.BUILD_GPU_LBVH, "lbvh"), (T.BUILD_GPU_PLOC, "ploc")):
        cases += [(f"{bn} reinsert={k}", b, k) for k in counts]
    for label, builder, k in cases:
        os.environ["FF_GPU_REINSERT"] = str(k)
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)
            t.upload_scene(scene)  # (warm)
            bs = t.build_stats()
            t.set_collect_stats(True)
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            st = t.stats()
            t.set_collect_stats(False)
            ks = []
            for _ in range(3):
                t.render(cam, p, want_rgb8=False, want_radiance=False)
                ks.append(t.stats().kernel_ms)
        print(f"{name} {label:18s} | build {bs.build_ms:8.2f} ms depth {bs.bvh_max_depth:3d} nodes4 {st.scene_bytes_nodes // 112:7d} | visits/ray {st.nodes_visited / st.rays_traced:6.3f} "
              f"tris/ray {st.tris_tested / st.rays_traced:6.3f} | {min(ks):8.2f} ms {st.rays_traced / min(ks) / 1e3:7.0f} Mrays/s", flush=True)
