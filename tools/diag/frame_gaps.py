#!/usr/bin/env python3
"""What one 1-spp frame's launches look like on the device: reads a rocprofv3 --kernel-trace CSV and prints, for the last frames
of the run, every kernel with its start relative to the frame's first kernel, its duration and the idle gap in front of it.
Usage: frame_gaps.py <dir with *_kernel_trace.csv> [frames]"""
import csv, glob, os, sys
d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame ends with combine_kernel
ends = [i for i, r in enumerate(rows) if "combine_kernel" in r["Kernel_Name"]]
for f in range(frames, 0, -1):
    hi = ends[-f]
    lo = ends[-f - 1] + 1
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = int(rows[lo - 1]["End_Timestamp"])
    print(f"frame -{f}: {hi - lo + 1} kernels, {(int(rows[hi]['End_Timestamp']) - t0) / 1e3:.1f} us from first start to last end; idle before it {(t0 - prev_end) / 1e3:.1f} us")
    for r in rows[lo:hi + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f} us  {r['Kernel_Name'][:70]}")
        prev_end = e
