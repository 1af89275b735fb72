#!/usr/bin/env python3
"""Work-queue knobs against frame kinds: kernel ms (best of N) for the number of queue counters (FF_QUEUE_COUNTERS) and the
chunk a wave takes per atomic (FF_QUEUE_CHUNK), one process per setting.
Frames: the reference's own frame (primary hit + abs(normal), 1 spp) on the shipped scene at 800x800 / 1080p and on the C2 box
at 1080p; path-traced C2 at 1080p with 1 / 4 / 16 / 64 spp; the 256-spp frame from the reference's default camera.
Usage: queue_sweep.py [counters,counters,... [chunk,chunk,...]]   ("default" is allowed in both lists)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from gpupathtracer_amd import lib, scenes
    from gpupathtracer_amd import types as T

    def best(t, cam, p, n):
        ks = []
        for _ in range(n):
            t.render(cam, p, want_rgb8=False, want_radiance=False)
            ks.append(t.stats().kernel_ms)
        return min(ks)

    out = []
    dbg = lambda w, h: lib.render_params(w, h, 1, 1, 1234, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)
    with lib.Tracer(0) as t:
        t.upload_scene(scenes.reference_scene(scenes.load_mesh("rocketman")))
        for w, h in ((800, 800), (1920, 1080)):
            out.append(f"ref{h} {best(t, scenes.default_camera(w, h), dbg(w, h), 30):.3f}")
    with lib.Tracer(0) as t:
        t.upload_scene(scenes.cornell_wahoo_scene())
        cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
        out.append(f"c2dbg {best(t, cam, dbg(1920, 1080), 30):.3f}")
        for spp, n in ((1, 12), (4, 8), (16, 5), (64, 3)):
            out.append(f"spp{spp} {best(t, cam, lib.render_params(1920, 1080, 8, spp), n):.2f}")
        out.append(f"defcam256 {best(t, scenes.default_camera(1920, 1080), lib.render_params(1920, 1080, 8, 256), 2):.2f}")
    print(" | ".join(out), flush=True)
else:
    counters = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "4", "8", "16"]
    chunks = sys.argv[2].split(",") if len(sys.argv) > 2 else ["default", "8", "16", "32", "64", "128", "256"]
    for n in counters:
        for chunk in chunks:
            env = dict(os.environ)
            if n != "default":
                env["FF_QUEUE_COUNTERS"] = n
            if chunk != "default":
                env["FF_QUEUE_CHUNK"] = chunk
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(f"counters {n:>7s} chunk {chunk:>7s}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
