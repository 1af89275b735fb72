#!/usr/bin/env python3
"""tools/fuzz_parity.py's exact sequence (three tracers: host SAH, LBVH, PLOC; brute force on the first) with details on a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401
import numpy as np
import fuzz_parity as fz
from gpupathtracer_amd import lib, types as T

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
with lib.Tracer(0) as sah, lib.Tracer(0) as lbvh, lib.Tracer(0) as ploc:
    lbvh.set_builder(T.BUILD_GPU_LBVH)
    ploc.set_builder(T.BUILD_GPU_PLOC)
    for c in range(cases):
        scene = fz.rand_scene(rng, crowd=int(rng.integers(30, 110)) if rng.random() < 0.15 else 0)
        w, h, cam = fz.rand_view(rng)
        p = fz.rand_params(rng, w, h, T.TRACE_BVH)
        out = {}
        for name, tr, mode in (("sah", sah, T.TRACE_BVH), ("lbvh", lbvh, T.TRACE_BVH), ("ploc", ploc, T.TRACE_BVH), ("brute", sah, T.TRACE_BRUTE_FORCE)):
            if name != "brute":
                tr.upload_scene(scene)
            p.trace_mode = mode
            out[name] = tr.render(cam, p)[1]
        for name, tr in (("sah", sah), ("lbvh", lbvh), ("ploc", ploc)):
            d = (out[name].view(np.uint32) != out["brute"].view(np.uint32)).any(axis=2)
            if d.any():
                ys, xs = np.nonzero(d)
                y, x = int(ys[0]), int(xs[0])
                print(f"case {c} {name}: {len(ys)} pixels differ; first ({x},{y}) bvh {out[name][y, x]} brute {out['brute'][y, x]}  geoms {len(scene)} {w}x{h} shade {p.shade_mode}")
                p.trace_mode = T.TRACE_BVH
                again = tr.render(cam, p)[1]
                tr.set_collect_stats(True)
                tr.render(cam, p)
                tr.set_collect_stats(False)
                print("   instrumented launch: stack overflows", tr.debug_counters()[26], "stack entries per lane / depths:", [int(v) for v in tr.download_bvh4(len(scene))[1][:, 2].clip(0)][-3:])
                print("   same tracer, rendered again:", again[y, x], "matches brute:", np.array_equal(again.view(np.uint32), out["brute"].view(np.uint32)))
                one = tr.render_tile(cam, p, x, y, 1, 1)[1]
                print("   as a 1x1 tile:", one[0, 0])
                n4a, t4a = tr.download_bvh4(len(scene))
                n2a, tra, t2a = tr.download_bvh(len(scene))
                tr.upload_scene(scene)
                n4b, t4b = tr.download_bvh4(len(scene))
                n2b, trb, t2b = tr.download_bvh(len(scene))
                print("   device data equal after re-upload: nodes4", np.array_equal(n4a.view(np.uint8)[:0] if False else n4a.view(np.uint8), n4b.view(np.uint8)), "table4", np.array_equal(t4a, t4b),
                      "binary nodes", np.array_equal(n2a.view(np.uint8), n2b.view(np.uint8)), "tris", np.array_equal(tra.view(np.uint8), trb.view(np.uint8)), "table", np.array_equal(t2a, t2b))
                if not np.array_equal(t4a, t4b):
                    bad = np.nonzero((t4a != t4b).any(axis=1))[0]
                    print("   table4 rows that differ:", [(int(i), [int(v) for v in t4a[i]], [int(v) for v in t4b[i]]) for i in bad[:6]])
                if not np.array_equal(n4a.view(np.uint8), n4b.view(np.uint8)):
                    used = np.zeros(len(n4a), bool)
                    for r in t4a:
                        if r[1] > 0:
                            used[r[0]:r[0] + r[1]] = True
                    diff = (n4a.view(np.uint8).reshape(len(n4a), -1) != n4b.view(np.uint8).reshape(len(n4b), -1)).any(axis=1)
                    print("   differing 4-wide nodes:", int(diff.sum()), "of which in use:", int((diff & used).sum()))
                up = tr.render(cam, p)[1]
                print("   uploaded again:", up[y, x], "matches brute:", np.array_equal(up.view(np.uint32), out["brute"].view(np.uint32)))
                tr.upload_scene(scene) if False else None
                _, table4 = tr.download_bvh4(len(scene))
                big = max(table4, key=lambda r: r[1])
                print("   biggest mesh row", [int(v) for v in big], "total nodes4", int(table4[:, 1].clip(0).sum()))
print("done")
