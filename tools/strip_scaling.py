#!/usr/bin/env python3
"""Kernel time of one rank's share of the benchmark frame (strips dealt round-robin) against full-frame time / ranks:
what strong scaling can reach before any communication."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes, dist

os.environ.setdefault("FF_NO_PRIMARY_CACHE", "1")  # (like bench.py: every timed frame pays for its own pre-pass)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene = scenes.cornell_wahoo_scene()
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
p = lib.render_params(1920, 1080, 8, spp)
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    t.render(cam, p, want_rgb8=False, want_radiance=False)
    t.render(cam, p, want_rgb8=False, want_radiance=False)
    full = t.stats().kernel_ms
    print(f"1 rank : kernel {full:8.2f} ms")
    for n in ((8,) if os.environ.get("FF_ONLY8") else (2, 4, 8)):
        rows = int(os.environ["FF_STRIP_ROWS"]) if os.environ.get("FF_STRIP_ROWS") else dist.strip_rows_for(n)
        worst = 0.0
        times = []
        for part in range(n):
            t.render_strips(cam, p, rows, part, n, want_rgb8=False, want_radiance=False)
            t.render_strips(cam, p, rows, part, n, want_rgb8=False, want_radiance=False)
            worst = max(worst, t.stats().kernel_ms)
            times.append(t.stats().kernel_ms)
        print(f"{n} ranks: slowest rank's kernel {worst:8.2f} ms = {full / n / worst * 100:.1f} % of ideal ({rows}-row strips); ranks: " + " ".join(f"{x:.1f}" for x in times))
