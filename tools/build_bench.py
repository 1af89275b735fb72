#!/usr/bin/env python3
"""Build / refit / rebuild times of the two BVH builders and the render rate on the trees they produce (one GPU).
Usage: python tools/build_bench.py [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inside = lambda w, h: scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)


def fmt(bs):
    return f"total {bs.total_ms:8.2f} ms (copy {bs.copy_ms:7.2f}, build {bs.build_ms:8.2f})  nodes {bs.bvh_nodes:7d} depth {bs.bvh_max_depth:2d}"


def run(name, scene, cam, params):
    gi = [i for i, s in enumerate(scene._specs) if s[0] == T.GEOM_TRIANGLEMESH]
    big = max(gi, key=lambda i: len(scene._specs[i][4]))
    tris = scene._specs[big][4]
    for builder, label in ((T.BUILD_HOST_SAH, "host SAH   "), (T.BUILD_GPU_LBVH, "device LBVH"), (T.BUILD_GPU_PLOC, "device PLOC")):
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            t.upload_scene(scene)      # warm (allocations, code objects)
            t.upload_scene(scene)
            print(f"{name} {label} upload : {fmt(t.build_stats())}")
            t.render(cam, params, want_rgb8=False, want_radiance=False)
            t.render(cam, params, want_rgb8=False, want_radiance=False)
            st = t.stats()
            t.set_collect_stats(True); t.render(cam, params, want_rgb8=False, want_radiance=False); c = t.stats(); t.set_collect_stats(False)
            print(f"{name} {label} render : {st.rays_traced / st.kernel_ms / 1e3:8.0f} Mrays/s  nodes/ray {c.nodes_visited / c.rays_traced:.2f} tris/ray {c.tris_tested / c.rays_traced:.2f}")
            moved = np.array(tris, copy=True); moved[:, 1:9:3] *= np.float32(1.01)
            t.update_mesh(big, moved, T.UPDATE_REFIT); t.update_mesh(big, moved, T.UPDATE_REFIT)
            print(f"{name} {label} refit  : {fmt(t.build_stats())}")
            if builder != T.BUILD_HOST_SAH:
                t.update_mesh(big, moved, T.UPDATE_REBUILD); t.update_mesh(big, moved, T.UPDATE_REBUILD)
                print(f"{name} {label} rebuild: {fmt(t.build_stats())}")
            t.update_transforms(scene)
            print(f"{name} {label} transf.: {fmt(t.build_stats())}")


run("C2", scenes.cornell_wahoo_scene(), inside(1920, 1080), lib.render_params(1920, 1080, 8, spp))
run("C3", scenes.blooper_scene(), scenes.posed_camera(1920, 1080, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0), lib.render_params(1920, 1080, 8, spp))
run("C4", scenes.sphere_stress_scene(5), inside(1920, 1080), lib.render_params(1920, 1080, 8, max(spp // 2, 1)))
