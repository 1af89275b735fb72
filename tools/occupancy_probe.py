#!/usr/bin/env python3
"""Print the SIMD occupancy of each phase of the BVH kernel (instrumented launch).
Usage: occupancy_probe.py [spp] [c2|c3|c4] [sah|lbvh]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes, types as T

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "c2"
builder = T.BUILD_GPU_LBVH if len(sys.argv) > 3 and sys.argv[3] == "lbvh" else T.BUILD_HOST_SAH
cam = scenes.posed_camera(1920, 1080, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
if which == "c3":
    scene = scenes.blooper_scene()
    cam = scenes.posed_camera(1920, 1080, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)
elif which == "c4":
    scene = scenes.sphere_stress_scene(5)
elif which.startswith("crowd"):
    # the C2 box with N small objects scattered through the room (tools/crowd_bench.py's scene)
    import numpy as np
    rng = np.random.default_rng(1)
    cube = scenes.load_mesh("cube")
    s = scenes.Scene()
    s.add_mesh(scenes.load_mesh("wahoo"), (0, -2.4, 0), (0, 0, 0), (0.28, 0.28, 0.28), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0)))
    s.add_mesh(cube, (1.5, -2.0, 1.0), (0, 0, 0), (1, 1, 1), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75)))
    for _ in range(int(which[5:])):
        k = int(rng.integers(0, 3))
        pos, rot = tuple(rng.uniform(-2.2, 2.2, 3)), tuple(rng.uniform(-180, 180, 3))
        bx = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=tuple(float(v) for v in rng.uniform(0.3, 0.9, 3)))
        if k == 0:
            s.add_mesh(cube, pos, rot, tuple(float(v) for v in rng.uniform(0.1, 0.3, 3)), bx)
        elif k == 1:
            s.add_sphere(float(rng.uniform(0.08, 0.2)), pos, rot, (1, 1, 1), bx)
        else:
            s.add_plane(pos, rot, tuple(float(v) for v in rng.uniform(0.15, 0.5, 3)), bx)
    scene = scenes._box(s).finalize()
else:
    scene = scenes.cornell_wahoo_scene()
with lib.Tracer(0) as t:
    t.set_builder(builder)
    t.upload_scene(scene)
    t.set_collect_stats(True)
    t.render(cam, lib.render_params(1920, 1080, 8, spp), want_rgb8=False, want_radiance=False)
    c = t.debug_counters()
    st = t.stats()
rays, nodes, tris, planes = c[0], c[1], c[2], c[3]
ir, lr, tr, pr, sr = c[8], c[9], c[10], c[11], c[12]
budget = os.environ.get("FF_SETUP_THRESHOLD")
print(f"budget {budget} rays {rays}  kernel_ms {st.kernel_ms:.1f}  Mrays/s {rays / st.kernel_ms / 1e3:.0f}")
print(f"segments: rounds {sr}  occupancy {rays / (64 * sr):.3f}")
print(f"planes  : per ray {planes / rays:.2f}   queries without mesh candidates (big scenes: mesh entries per ray): {c[14] / rays:.3f}   exact plane tests per ray: {c[15] / rays:.4f}   screening rounds/segment-round {pr / max(sr, 1):.2f}")
print(f"inner   : per ray {nodes / rays:.2f}  rounds/segment-round {ir / sr:.2f}  occupancy {nodes / (64 * ir):.3f}")
print(f"leaves  : rounds/segment-round {lr / sr:.2f}")
print(f"tris    : per ray {tris / rays:.2f}  rounds/segment-round {tr / sr:.2f}  occupancy {tris / (64 * tr):.3f}")
tot = c[4] + c[5] + c[6] + c[7] + c[13]
if tot:
    names = ["resolve", "shade", "acquire", "begin", "traverse"]
    vals = [c[4], c[5], c[6], c[7], c[13]]
    print("wave-cycle shares: " + "  ".join(f"{n} {v / tot:.3f}" for n, v in zip(names, vals)))
tt = c[16] + c[17] + c[18]
if tt:
    print(f"traverse split: mesh-start {c[16] / tt:.3f}  inner {c[17] / tt:.3f}  leaf {c[18] / tt:.3f}   "
          f"cycles per inner round {c[17] / max(ir, 1):.0f}  per triangle round {c[18] / max(tr, 1):.0f}")
tl = c[28] + c[29] + c[30]
if tl:
    print(f"stack entries pushed beyond the LDS levels (spilled), per ray: {c[26] / rays:.4f}")
    print(f"leaf split: record wait {c[28] / tl:.3f}  triangle tests {c[29] / tl:.3f}  pop {c[30] / tl:.3f}   cycles per triangle round: wait {c[28] / max(tr, 1):.0f}  "
          f"test {c[29] / max(tr, 1):.0f};  per leaf round: pop {c[30] / max(lr, 1):.0f}")
tb = c[19] + c[20] + c[21]
if tb:
    print(f"begin split: quad boxes {c[19] / tb:.3f}  quad screens {c[20] / tb:.3f}  mesh boxes {c[21] / tb:.3f}   cycles per segment round {tb / max(sr, 1):.0f}")
loop = c[4] + c[5] + c[6] + c[7] + c[13]
print(f"stamped loop time per wave: mean {loop / 4096 / 1e6:.2f} M ticks, slowest wave {c[22] / 1e6:.2f} M; kernel {st.kernel_ms:.2f} ms = {st.kernel_ms * 2.4:.2f} M cycles at 2.4 GHz")
M = (1 << 64) - 1
if c[23] and c[24] and c[25]:
    t_start, t_empty, t_end = M - c[24], M - c[23], c[25]
    print(f"wall clock: launch {(t_end - t_start) / 100:.0f} us, of which the queue was empty for the last {(t_end - t_empty) / 100:.0f} us (the tail)")
