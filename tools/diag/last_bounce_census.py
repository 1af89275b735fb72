#!/usr/bin/env python3
"""VERDICT r3 item 3 (any-hit for last-bounce queries that hold an emitter): how many such queries are there?

Paths are identical up to their last segment whatever `bounces` is (the RNG is keyed on pixel, sample and bounce), so
   last-bounce queries of a B-bounce frame = rays(B) - rays(B - 1),
of which FfStats::rays_cut_short held no emitter after the analytic records and ended there; the rest hold the emitter plane and
walk the meshes to find out whether anything lies in front of it.  An any-hit walk could only shorten THOSE, and only the ones
that are in fact blocked; the others must complete their walk as they do now.  Prints the census for the headline scene."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpupathtracer_amd import lib, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
with lib.Tracer(0) as t:
    t.upload_scene(scenes.cornell_wahoo_scene())
    out = {}
    for b in (7, 8):
        t.render(cam, lib.render_params(1920, 1080, b, spp, 1234), want_rgb8=False, want_radiance=False)
        st = t.stats()
        out[b] = (st.rays_traced, st.rays_cut_short, st.rays_answered)
r8, cut8, ans8 = out[8]
last = r8 - out[7][0]
holders = last - cut8
print(f"C2 1080p, {spp} spp: path segments {r8}; last-bounce queries (bounce index 7) {last} = {100.0 * last / r8:.2f} % of all segments")
print(f"  ended after the planes (no emitter held): {cut8} = {100.0 * cut8 / r8:.2f} %")
print(f"  hold the emitter and walk the meshes:     {holders} = {100.0 * holders / r8:.2f} % of all segments, {100.0 * holders / (r8 - ans8):.2f} % of the traversed ones")
