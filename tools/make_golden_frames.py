#!/usr/bin/env python3
"""Generate tests/golden/frames.npz with the CPU oracle (oracle/ff_oracle.c) — run in the build container.

Every entry is an (input description, expected output) pair: the scene preset + camera pose + render params are
named in CASES below (rebuilt identically by tests/cases.py), the arrays are the oracle's framebuffers.
NORMAL_DEBUG cases are the reference's behaviour (kernel.cu:186-221); DIFFUSE_PATH cases are the build-defined
integrator (parity unpinned beyond the oracle itself, see oracle/ff_oracle.h).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import CASES, build_case  # noqa: E402
from oracle_lib import oracle_render  # noqa: E402


def main():
    out = {}
    for name in CASES:
        scene, cam, params = build_case(name)
        rgb8, rad = oracle_render(scene, cam, params, threads=8)
        out[name + "/rgb8"] = rgb8
        out[name + "/radiance"] = rad
        lit = int(rgb8.reshape(-1, 3).any(axis=1).sum())
        print(f"{name}: {params.width}x{params.height} lit={lit} mean={rad.mean():.6f}")
    path = os.path.join(ROOT, "tests", "golden", "frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
