#!/usr/bin/env python3
"""First contact of the job-pool kernel with a GPU: small frames with FF_POOL=1 against FF_POOL=0 (the lane-owned kernel), bit for
bit, with ray counts.  Run under `timeout`: a scheduling bug in a persistent kernel is a hang, not a wrong pixel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpupathtracer_amd import lib, scenes  # noqa: E402
from gpupathtracer_amd import types as T  # noqa: E402


def frames(pool, cases):
    os.environ["FF_POOL"] = pool
    out = []
    with lib.Tracer(0) as t:
        for scene, cam, params in cases:
            t.upload_scene(scene)
            t0 = time.perf_counter()
            rgb8, rad = t.render(cam, params)
            st = t.stats()
            out.append((rgb8.copy(), rad.view(np.uint32).copy(), st.rays_traced, st.rays_answered, st.rays_cut_short, t.kernel_name(), (time.perf_counter() - t0) * 1e3))
            print(f"  pool={pool} {params.width}x{params.height} b{params.bounces} spp{params.spp}: {st.rays_traced} rays, {out[-1][-1]:.1f} ms, {t.kernel_name()}", flush=True)
    return out


def main():
    sizes = [(64, 36, 4, 2), (96, 64, 8, 70), (320, 180, 8, 16)]
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        sizes += [(1920, 1080, 8, 16)]
    cases = []
    for w, h, b, spp in sizes:
        cam = scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
        cases.append((scenes.cornell_wahoo_scene(), cam, lib.render_params(w, h, b, spp, 7)))
    cases.append((scenes.cornell_glass_scene(), scenes.posed_camera(96, 64, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0), lib.render_params(96, 64, 6, 9, 3)))
    cases.append((scenes.blooper_scene(), scenes.posed_camera(128, 72, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0), lib.render_params(128, 72, 8, 5, 3)))
    a = frames("0", cases)
    b = frames("1", cases)
    ok = True
    for i, (x, y) in enumerate(zip(a, b)):
        same = np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2:5] == y[2:5]
        ok = ok and same
        print(f"case {i}: {'same bits and counts' if same else 'DIFFERENT'}  rays {x[2]} / {y[2]}  answered {x[3]} / {y[3]}  cut {x[4]} / {y[4]}  {x[5]} vs {y[5]}  {x[6]:.1f} / {y[6]:.1f} ms")
        if not same:
            d = (x[1] != y[1]).any(axis=2)
            print(f"   pixels that differ: {int(d.sum())} of {d.size}")
    print("POOL SMOKE", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
