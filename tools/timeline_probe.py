#!/usr/bin/env python3
"""Throughput of one launch over its own wall clock: rays completed per bucket (FF_DEBUG_TIMELINE_US, instrumented kernel), for
C2 frames of a few sizes and sample counts.  Separates the ramp, the plateau and the drain of a short launch, and prints next
to them the plain (uninstrumented) kernel time of the same frame.
Usage: timeline_probe.py [bucket_us [WxHxSPP ...]]   (FF_ITEMS_PER_FETCH / FF_BLOCK_THREADS apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bucket = int(sys.argv[1]) if len(sys.argv) > 1 else 50
os.environ["FF_DEBUG_TIMELINE_US"] = str(bucket)
from gpupathtracer_amd import lib, scenes

scene = scenes.cornell_wahoo_scene()
cases = [(1920, 1080, 1), (3840, 2160, 1), (960, 540, 4), (1920, 1080, 4), (1920, 1080, 16), (1920, 1080, 64)]
if len(sys.argv) > 2:
    cases = [tuple(int(v) for v in c.split("x")) for c in sys.argv[2:]]
with lib.Tracer(0) as t:
    t.upload_scene(scene)
    for w, h, spp in cases:
        cam = scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
        p = lib.render_params(w, h, 8, spp)
        ks = []
        for _ in range(5):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            ks.append(t.stats().kernel_ms)
        rays = t.stats().rays_traced
        t.set_collect_stats(True)
        t.render(cam, p, want_rgb8=False, want_radiance=False)
        inst_ms = t.stats().kernel_ms
        t.set_collect_stats(False)
        us, counts = t.debug_timeline()
        last = max(i for i in range(1024) if counts[i]) if counts.any() else 0
        rate = counts[: last + 1] / (us * 1e-6) / 1e9  # Grays/s per bucket
        peak = float(rate.max())
        plateau = [i for i in range(last + 1) if rate[i] >= 0.8 * peak]
        print(f"{w}x{h} spp {spp}: plain {min(ks):.2f} ms = {rays / min(ks) / 1e6:.2f} Grays/s | {t.kernel_name()} | instrumented {inst_ms:.2f} ms, "
              f"{int(counts.sum())} rays in {last + 1} buckets of {us} us; peak {peak:.2f} Grays/s; >=80% of peak from {plateau[0] * us} to {(plateau[-1] + 1) * us} us")
        step = max(1, (last + 1) // 40)
        print("   Grays/s per bucket" + (f" (every {step}th)" if step > 1 else "") + ": " + " ".join(f"{r:.1f}" for r in rate[::step]), flush=True)
