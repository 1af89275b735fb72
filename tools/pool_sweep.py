#!/usr/bin/env python3
"""Same-box sweep of the trace kernels' knobs: every configuration is a set of FF_* switches (read at ff_create), rendered on one
box in one process, the configurations interleaved `--reps` times; prints the median kernel rate of each.

    pool_sweep.py [--scene c2|c3|c4] [--spp 128] [--reps 3] [--size 1920x1080] "FF_POOL=0" "FF_POOL=1" "FF_POOL=1,FF_POOL_QUORUM=48" ...
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd import lib, scenes  # noqa: E402


def isolated(a):
    """Every run in a process of its own (the library is loaded once per process)."""
    import json
    import subprocess
    rates = {c: [] for c in a.configs}
    bits = {}
    for rep in range(a.reps):
        for c in a.configs:
            env = dict(os.environ)
            for kv in c.split(","):
                if "=" in kv:
                    k, v = kv.split("=", 1)
                    env[k] = v
            cmd = [sys.executable, os.path.abspath(__file__), "--one", "--scene", a.scene, "--spp", str(a.spp), "--bounces", str(a.bounces), "--size", a.size, "--camera", a.camera, "--reps", "1", c] + (["--keep-primary-hits"] if a.keep_primary_hits else [])
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("ONE ")]
            if not line:
                print(f"  rep {rep} [{c}] FAILED: {out.stdout[-300:]} {out.stderr[-300:]}", flush=True)
                continue
            d = json.loads(line[0][4:])
            rates[c].append(d["rate"])
            bits.setdefault(c, (d["crc"], d["rays"]))
            print(f"  rep {rep} [{c}] {d['rate']:.0f} Mrays/s  {d['ms']:.2f} ms  {d['kernel']}", flush=True)
    print(f"# {a.scene} {a.size} {a.bounces} bounces {a.spp} spp, median of {a.reps} interleaved runs, one process each (kernel time)")
    base = float(np.median(rates[a.configs[0]]))
    for c in a.configs:
        m = float(np.median(rates[c])) if rates[c] else float("nan")
        same = "" if c == a.configs[0] or c not in bits else ("  same radiance checksum and ray count" if bits[c] == bits[a.configs[0]] else "  DIFFERENT radiance checksum or ray count")
        print(f"{m:9.0f} Mrays/s  {100.0 * (m / base - 1.0):+6.2f} %  [{c}]{same}")


def main():
    # (like bench.py: every frame runs its own pre-pass unless a configuration says otherwise - FF_NO_PRIMARY_CACHE= with an empty value
    # is still "set"; use the --keep-primary-hits flag to measure the viewer-at-rest case)
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep-primary-hits", action="store_true", help="let the library keep the stored primary hits between frames (same camera and scene)")
    ap.add_argument("--scene", default="c2")
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--camera", default="inside", choices=["inside", "default"], help="default: the reference's camera (kernel.cu:312-321), 12.5 units outside the box")
    ap.add_argument("--check", action="store_true", help="compare every configuration's radiance bits and ray count with the first one's")
    ap.add_argument("--isolate", action="store_true", help="one process per run: needed when a configuration names another library (FF_LIB_PATH=...)")
    ap.add_argument("--one", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    if not a.keep_primary_hits:
        os.environ.setdefault("FF_NO_PRIMARY_CACHE", "1")
    if a.isolate:
        return isolated(a)
    w, h = (int(v) for v in a.size.split("x"))
    scene = {"c2": scenes.cornell_wahoo_scene, "c3": scenes.blooper_scene, "c4": scenes.sphere_stress_scene}[a.scene]()
    cam = scenes.posed_camera(w, h, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0) if a.scene == "c3" else scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    if a.camera == "default":
        cam = scenes.default_camera(w, h)
    params = lib.render_params(w, h, a.bounces, a.spp, 1234)
    if a.one:
        import json
        import zlib
        c = a.configs[0]
        with lib.Tracer(0) as t:
            t.upload_scene(scene)
            # (hits kept: a one-off frame, the frame that finds the camera at rest and runs the last pre-pass, then the measured one)
            for _ in range(2 if a.keep_primary_hits else 1):
                t.render(cam, params, want_rgb8=False, want_radiance=False)
            _, rad = t.render(cam, params, want_rgb8=False)
            st = t.stats()
            print("ONE " + json.dumps({"rate": st.rays_traced / st.kernel_ms / 1e3, "ms": st.kernel_ms, "rays": int(st.rays_traced), "crc": zlib.crc32(rad.tobytes()), "kernel": t.kernel_name()}))
        return
    rates = {c: [] for c in a.configs}
    occ = {}
    ref = None
    touched = set()
    for rep in range(a.reps):
        for c in a.configs:
            for k in touched:
                os.environ.pop(k, None)
            for kv in c.split(","):
                if "=" in kv:
                    k, v = kv.split("=", 1)
                    os.environ[k] = v
                    touched.add(k)
            with lib.Tracer(0) as t:
                t.upload_scene(scene)
                for _ in range(2 if a.keep_primary_hits else 1):  # warm-up (hits kept: a one-off frame, then the one that runs the last pre-pass)
                    t.render(cam, params, want_rgb8=False, want_radiance=False)
                if a.check and rep == 0:
                    _, rad = t.render(cam, params)
                    bits = rad.view(np.uint32).copy()
                else:
                    t.render(cam, params, want_rgb8=False, want_radiance=False)
                st = t.stats()
                ctr = t.debug_counters()
                rates[c].append(st.rays_traced / st.kernel_ms / 1e3)
                occ[c] = (ctr[1] / max(1, 64 * ctr[8]), ctr[14] / max(1, 64 * ctr[9]), ctr[8], ctr[9], ctr[16], ctr[17], ctr[18], ctr[11], ctr[12]) if ctr[8] else None
                if a.check and rep == 0:
                    if ref is None:
                        ref = (bits, st.rays_traced)
                    else:
                        same = np.array_equal(bits, ref[0]) and st.rays_traced == ref[1]
                        print(f"  check [{c}]: {'same bits and ray count' if same else 'DIFFERENT from the first configuration'}", flush=True)
                name = t.kernel_name()
            print(f"  rep {rep} [{c}] {rates[c][-1]:.0f} Mrays/s  {st.kernel_ms:.2f} ms  {name}", flush=True)
    print(f"# {a.scene} {w}x{h} {a.bounces} bounces {a.spp} spp, median of {a.reps} interleaved runs (kernel time)")
    base = float(np.median(rates[a.configs[0]]))
    for c in a.configs:
        m = float(np.median(rates[c]))
        o = occ.get(c)
        extra = f"  inner-round occupancy {o[0]:.3f} ({o[2] / 1e6:.1f} M rounds)  leaf-phase occupancy {o[1]:.3f} ({o[3] / 1e6:.1f} M)" if o else ""
        if o and o[4] + o[5] + o[6]:
            tot = o[4] + o[5] + o[6]
            extra += f"  time: setup {o[4] / tot:.3f} role {o[5] / tot:.3f} rest {o[6] / tot:.3f}; cycles per setup pass {o[4] / max(1, o[7]):.0f} ({o[7] / 1e6:.2f} M), per role visit {o[5] / max(1, o[8]):.0f} ({o[8] / 1e6:.2f} M)"
        print(f"{m:9.0f} Mrays/s  {100.0 * (m / base - 1.0):+6.2f} %  [{c}]{extra}")


if __name__ == "__main__":
    main()
