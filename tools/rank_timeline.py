#!/usr/bin/env python3
"""Throughput over the wall clock of ONE rank's share of the headline frame (strips dealt round-robin to `ranks` parts):
where the launch loses time against full-frame time / ranks - the ramp or the dry end of the work queue.
Usage: rank_timeline.py [ranks [bucket_us [spp]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bucket = int(sys.argv[2]) if len(sys.argv) > 2 else 500
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
os.environ["FF_DEBUG_TIMELINE_US"] = str(bucket)
from gpupathtracer_amd import lib, scenes, dist
from gpupathtracer_amd import types as T

scene = scenes.cornell_wahoo_scene()
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
p = lib.render_params(1920, 1080, 8, spp)
rows = dist.strip_rows_for(ranks)
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    for _ in range(2):
        t.render_strips(cam, p, rows, 0, ranks, want_rgb8=False, want_radiance=False)
    st = t.stats()
    plain, rays, flags = st.kernel_ms, st.rays_traced, st.flags
    t.set_collect_stats(True)
    t.render_strips(cam, p, rows, 0, ranks, want_rgb8=False, want_radiance=False)
    inst = t.stats().kernel_ms
    t.set_collect_stats(False)
    us, counts = t.debug_timeline()
    last = max(i for i in range(1024) if counts[i])
    rate = counts[: last + 1] / (us * 1e-6) / 1e9
    peak = float(sorted(rate)[len(rate) // 2])  # the plateau: the median bucket
    print(f"part 0 of {ranks} ({rows}-row strips), {spp} spp: plain {plain:.2f} ms, {rays / plain / 1e6:.2f} Grays/s, tail items: {bool(flags & T.FF_STATS_TAIL_ITEMS)}; "
          f"instrumented {inst:.2f} ms in {last + 1} buckets of {us} us, plateau {peak:.2f} Grays/s")
    lost = sum(max(0.0, 1.0 - r / peak) for r in rate) * us / 1e3
    head = sum(max(0.0, 1.0 - r / peak) for r in rate[: len(rate) // 4]) * us / 1e3
    print(f"   time lost against the plateau rate: {lost:.2f} ms of the instrumented launch, {head:.2f} ms of it in the first quarter (ramp), the rest at the end")
    print("   last 24 buckets, share of the plateau: " + " ".join(f"{r / peak:.2f}" for r in rate[-24:]))
    print("   first 8 buckets: " + " ".join(f"{r / peak:.2f}" for r in rate[:8]))
