"""Which kernel sources a measurement belongs to: the sha256 of the files that determine the trace kernels' machine code AND what
they are launched on (the builders, the 4-wide collapse, the LDS / stack-spill layout, the slice length and the per-frame launch
parameters all move nodes per ray, LDS hit rates and the PMC counters: ff_build.hip, ff_scene.cpp, ff_api.cpp).
tools/pmc_to_json.py stamps it into profiles/*_pmc.json; bench.py withholds the PMC-derived roofline fields when the stamp
does not match the tree it runs in (a counter file from other sources says nothing about this kernel)."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KERNEL_SOURCES = ("ff_kernels.hip", "ff_kernels.h", "ff_internal.h", "ff_state.h", "ff_build.hip", "ff_build.h", "ff_scene.cpp", "ff_api.cpp", "Makefile")


def kernel_source_hash():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_CSRC, name), "rb") as f:
            data = f.read()
        if name == "Makefile":  # only what reaches the compiler: the flag lines
            data = b"\n".join(l for l in data.splitlines() if l.startswith((b"FLAGS", b"ARCH", b"           -")))
        h.update(name.encode() + b"\0" + data + b"\0")
    return h.hexdigest()[:16]
