#!/usr/bin/env python3
"""Do two builds of the same mesh give the same tree?  Device builders with and without their treelet / reinsertion passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gpupathtracer_amd import lib, scenes, types as T
scene = scenes.cornell_wahoo_scene()
for builder, bn in ((T.BUILD_GPU_LBVH, "lbvh"), (T.BUILD_GPU_PLOC, "ploc")):
    for tp in ("0", "2"):
        for rp in ("0", "1", "8"):
            os.environ["FF_TREELET_PASSES"] = tp
            os.environ["FF_GPU_REINSERT"] = rp
            got = []
            for _ in range(3):
                with lib.Tracer(0) as t:
                    t.set_builder(builder)
                    t.upload_scene(scene)
                    nodes, tris, table = t.download_bvh(len(scene))
                    got.append((np.asarray(nodes).tobytes(), np.asarray(tris).tobytes()))
            print(f"{bn} treelet passes {tp} reinsertion passes {rp}: {'same tree three times' if got[0] == got[1] == got[2] else 'DIFFERENT trees'}", flush=True)
