#!/usr/bin/env python3
"""Randomised exactness fuzz on the GPU: random scenes (meshes, planes, spheres under random rotations / non-uniform
scales, random materials), random cameras; the BVH kernel (host SAH trees, device LBVH and PLOC trees) must reproduce the
brute-force kernel (the reference's loop, kernel.cu:133-155) bit for bit: radiance, rgb8 and ray counts.
Usage: fuzz_parity.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

NORM = {"cube": 1.0, "sphere": 1.0, "sphereBlender": 1.0, "wahoo": 0.12, "rocketman": 0.4}
_MESHES = {}


def meshes():
    if not _MESHES:
        for n in NORM:
            _MESHES[n] = scenes.load_mesh(n)
    return _MESHES


def rand_bxdf(rng):
    k = rng.integers(0, 10)
    col = tuple(float(v) for v in rng.uniform(0.2, 1.0, 3))
    if k < 5:
        return scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=col)
    if k < 7:
        return scenes.make_bxdf(T.BXDF_MIRROR, specular=col)
    if k < 8:
        return scenes.make_bxdf(T.BXDF_GLASS, specular=(1, 1, 1), transmittance=col, ior=float(rng.uniform(1.1, 2.0)))
    return scenes.make_bxdf(T.BXDF_EMITTER, emissive=col, intensity=float(rng.uniform(1.0, 4.0)))


def rand_scene(rng, small=False, crowd=0):
    """1-3 meshes, 0-6 planes (axis-aligned, 45-degree and arbitrary rotations), 0-3 spheres, one large emitter above;
    crowd > 0 adds that many small cubes / spheres / quads (scenes of more than 32 geometries: several record chunks)."""
    s = scenes.Scene()
    names = ["cube", "sphere"] if small else list(NORM)
    for _ in range(crowd):
        k = int(rng.integers(0, 3))
        pos, rot = tuple(rng.uniform(-3, 3, 3)), tuple(rng.uniform(-180, 180, 3))
        if k == 0:
            s.add_mesh(meshes()["cube"], pos, rot, tuple(float(v) for v in rng.uniform(0.2, 0.7, 3)), rand_bxdf(rng))
        elif k == 1:
            s.add_sphere(float(rng.uniform(0.15, 0.5)), pos, rot, (1, 1, 1), rand_bxdf(rng))
        else:
            s.add_plane(pos, rot, tuple(float(v) for v in rng.uniform(0.3, 1.5, 3)), rand_bxdf(rng))
    for _ in range(int(rng.integers(1, 3 if small else 4))):
        name = names[int(rng.integers(0, len(names)))]
        sc = NORM[name] * rng.uniform(0.5, 2.0, 3)
        if rng.random() < 0.5:
            sc[:] = sc[0]
        s.add_mesh(meshes()[name], tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(-180, 180, 3)), tuple(float(v) for v in sc), rand_bxdf(rng))
    for _ in range(int(rng.integers(0, 7))):
        s.add_plane(tuple(rng.uniform(-3, 3, 3)), tuple(rng.choice([0.0, 90.0, 45.0, float(rng.uniform(-180, 180))], 3)),
                    tuple(float(v) for v in rng.uniform(1, 8, 3)), rand_bxdf(rng))
    for _ in range(int(rng.integers(0, 4))):
        s.add_sphere(float(rng.uniform(0.2, 1.5)), tuple(rng.uniform(-2.5, 2.5, 3)), tuple(rng.uniform(-180, 180, 3)),
                     tuple(float(v) for v in rng.uniform(0.5, 1.5, 3)), rand_bxdf(rng))
    s.add_plane((0, 4.5, 0), (90, 0, 0), (12, 12, 12), scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0))
    return s.finalize()


def rand_view(rng, max_w=200, max_h=150):
    w, h = int(rng.integers(24, max_w)), int(rng.integers(18, max_h))
    pos = rng.uniform(-3, 3, 3)
    pos[2] = abs(pos[2]) + 2.0
    cam = scenes.posed_camera(w, h, position=tuple(float(v) for v in pos), yaw=float(rng.uniform(-130, -50)), pitch=float(rng.uniform(-30, 30)))
    return w, h, cam


def rand_params(rng, w, h, mode):
    bounces, spp, seed = int(rng.integers(1, 9)), int(rng.integers(1, 5)), int(rng.integers(0, 1 << 30))
    if rng.random() < 0.2:
        return lib.render_params(w, h, 1, 1, seed, mode, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)
    shade = T.SHADE_DIFFUSE_PATH_SMOOTH if rng.random() < 0.3 else T.SHADE_DIFFUSE_PATH
    return lib.render_params(w, h, bounces, spp, seed, mode, shade, T.GRID_FULL, 0)


def run(cases, seed, verbose=True, crowd_fraction=0.15):
    """BVH (all tree kinds) against brute force.  Returns (mismatching renders, rays per kernel)."""
    rng = np.random.default_rng(seed)
    bad, total_rays = 0, 0
    with lib.Tracer(0) as sah, lib.Tracer(0) as lbvh, lib.Tracer(0) as ploc:
        lbvh.set_builder(T.BUILD_GPU_LBVH)
        ploc.set_builder(T.BUILD_GPU_PLOC)
        for c in range(cases):
            scene = rand_scene(rng, crowd=int(rng.integers(30, 110)) if rng.random() < crowd_fraction else 0)
            w, h, cam = rand_view(rng)
            p = rand_params(rng, w, h, T.TRACE_BVH)
            out = {}
            for name, tr, mode in (("sah", sah, T.TRACE_BVH), ("lbvh", lbvh, T.TRACE_BVH), ("ploc", ploc, T.TRACE_BVH), ("brute", sah, T.TRACE_BRUTE_FORCE)):
                if name != "brute":
                    tr.upload_scene(scene)
                p.trace_mode = mode
                rgb8, rad = tr.render(cam, p)
                out[name] = (rgb8, rad, tr.stats().rays_traced)
            total_rays += out["brute"][2]
            for name in ("sah", "lbvh", "ploc"):
                same = (np.array_equal(out[name][0], out["brute"][0]) and np.array_equal(out[name][1].view(np.uint32), out["brute"][1].view(np.uint32))
                        and out[name][2] == out["brute"][2])
                if not same:
                    bad += 1
                    d = (out[name][1].view(np.uint32) != out["brute"][1].view(np.uint32)).any(axis=2)
                    if verbose:
                        print(f"case {c} {name}: MISMATCH {int(d.sum())} pixels, rays {out[name][2]} vs {out['brute'][2]} "
                              f"({w}x{h} b{p.bounces} s{p.spp} seed {p.seed} geoms {len(scene)})", flush=True)
            if verbose and (c + 1) % 100 == 0:
                print(f"{c + 1} cases, {total_rays} rays, {bad} mismatches", flush=True)
    return bad, total_rays


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    nbad, rays = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"done: {n} cases, {rays} rays per kernel, {nbad} mismatches")
    sys.exit(1 if nbad else 0)
