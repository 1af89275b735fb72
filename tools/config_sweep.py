#!/usr/bin/env python3
"""Measure the other BASELINE configs (C3 blooper scene, C4 million-triangle sphere, C5 4K) on one GPU at reduced spp."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes

def run(name, scene, w, h, bounces, spp, cam):
    with lib.Tracer(0) as t:
        t0 = time.perf_counter(); t.upload_scene(scene); up = time.perf_counter() - t0
        info = lib.scene_info(scene)
        p = lib.render_params(w, h, bounces, spp)
        t.render(cam, p, want_rgb8=False, want_radiance=False)  # warm
        t.set_collect_stats(True); t.render(cam, p, want_rgb8=False, want_radiance=False); c = t.stats(); t.set_collect_stats(False)
        t.render(cam, p, want_rgb8=False, want_radiance=False); st = t.stats()
        t0 = time.perf_counter(); rgb8, rad = t.render(cam, p); host = (time.perf_counter() - t0) * 1e3
        print(f"{name}: {w}x{h} b{bounces} spp{spp} tris {info.num_triangles} nodes {info.bvh_nodes} depth {info.bvh_max_depth} upload {up*1e3:.0f} ms | "
              f"{st.rays_traced/st.kernel_ms/1e3:.0f} Mrays/s kernel {st.kernel_ms:.1f} ms, host-buffer call {host:.1f} ms | "
              f"nodes/ray {c.nodes_visited/c.rays_traced:.2f} tris/ray {c.tris_tested/c.rays_traced:.2f} mean radiance {rad.mean():.4f}")

inside = lambda w, h: scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
run("C2", scenes.cornell_wahoo_scene(), 1920, 1080, 8, 64, inside(1920, 1080))
run("C3", scenes.blooper_scene(), 1920, 1080, 8, 64, scenes.posed_camera(1920, 1080, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0))
run("C4", scenes.sphere_stress_scene(5), 1920, 1080, 8, 32, inside(1920, 1080))
run("C5", scenes.cornell_wahoo_scene(), 3840, 2160, 16, 16, inside(3840, 2160))
