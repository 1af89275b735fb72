#!/usr/bin/env python3
"""Build-time check on the ISA of csrc/ff_build.hip: no agent-scope (sc1) store may reach an arrival
atomic (global_atomic_add) without an `s_waitcnt vmcnt(0)` in between.

The bottom-up builders hand boxes / costs from thread to thread through sc1 stores + an arrival counter; the release is "my
stores have completed before my arrival counts" (release_arrival), which in the binary is exactly that wait.  The scan is
linear in program text per kernel (a store seen on any earlier line counts until a full wait is seen), which is conservative
for the straight-line climb loops these kernels are.

usage: check_arrivals.py [ff_build.s]   (without an argument: compiles csrc/ff_build.hip with the Makefile's flags to a temp file)
exit status 0 = clean, 1 = a violation (printed with its kernel and line).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpupathtracer_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-fno-slp-vectorize", "--cuda-device-only", "-S"]


def compile_isa(out_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + FLAGS + ["-o", out_path, os.path.join(CSRC, "ff_build.hip")], stderr=subprocess.DEVNULL)


def scan(path):
    """Returns (violations, arrivals_checked): violations = [(kernel, line_no, store_line_no)]."""
    kernel = None
    dirty_at = None  # line of the oldest sc1 store not yet covered by a vmcnt(0) wait
    violations, arrivals = [], 0
    sym = re.compile(r"^(_ZN2ff[^:\s]*):")
    for no, line in enumerate(open(path, errors="replace"), 1):
        m = sym.match(line)
        if m:
            kernel, dirty_at = m.group(1), None
            continue
        if kernel is None:
            continue
        text = line.split(";", 1)[0].strip()
        if not text:
            continue
        if text.startswith("s_endpgm"):
            dirty_at = None
        elif text.startswith("s_waitcnt") and "vmcnt(0)" in text:
            dirty_at = None
        elif text.startswith("global_store") and " sc1" in text + " ":
            if dirty_at is None:
                dirty_at = no
        elif text.startswith("global_atomic_add"):
            arrivals += 1
            if dirty_at is not None:
                violations.append((kernel, no, dirty_at))
    return violations, arrivals


def main():
    if len(sys.argv) > 1:
        path, tmp = sys.argv[1], None
    else:
        tmp = tempfile.NamedTemporaryFile(suffix=".s", delete=False)
        tmp.close()
        path = tmp.name
        compile_isa(path)
    try:
        violations, arrivals = scan(path)
    finally:
        if tmp is not None:
            os.unlink(path)
    for kernel, no, store in violations:
        print(f"VIOLATION {kernel}: global_atomic_add at line {no} behind an sc1 store at line {store} with no s_waitcnt vmcnt(0) between them")
    print(f"check_arrivals: {arrivals} arrival atomics checked, {len(violations)} violations")
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main())
