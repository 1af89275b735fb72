#!/usr/bin/env python3
"""Re-create case N of tools/fuzz_parity.py's sequence (same seed) and show where BVH and brute force differ.
Usage: fuzz_case.py SEED CASE [builder 0|1|2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401
import numpy as np
import fuzz_parity as fz
from gpupathtracer_amd import lib, types as T

seed, case = int(sys.argv[1]), int(sys.argv[2])
builder = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
first = int(os.environ.get("FIRST_CASE", case))
with lib.Tracer(0) as t:
    t.set_builder(builder)
    for c in range(case + 1):
        scene = fz.rand_scene(rng, crowd=int(rng.integers(30, 110)) if rng.random() < 0.15 else 0)
        w, h, cam = fz.rand_view(rng)
        p = fz.rand_params(rng, w, h, T.TRACE_BVH)
        if c < first:
            continue
        t.upload_scene(scene)
        p.trace_mode = T.TRACE_BVH
        a8, a = t.render(cam, p)
        name = t.kernel_name()
        p.trace_mode = T.TRACE_BRUTE_FORCE
        b8, b = t.render(cam, p)
        d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
        ys, xs = np.nonzero(d)
        print("case", c, "geoms", len(scene), w, h, "bounces", p.bounces, "spp", p.spp, "shade", p.shade_mode, name, "differing pixels", len(ys), flush=True)
        for y, x in list(zip(ys, xs))[:5]:
            print("   pixel", x, y, "bvh", a[y, x], "brute", b[y, x])
            p.trace_mode = T.TRACE_BVH
            again = t.render(cam, p)[1]
            print("   rendered again: bvh", again[y, x], "equal to brute now:", np.array_equal(again.view(np.uint32), b.view(np.uint32)))
            tile = t.render_tile(cam, p, int(x), int(y), 1, 1)[1]
            print("   as a 1x1 tile:", tile[0, 0])
