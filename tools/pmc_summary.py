#!/usr/bin/env python3
"""Summarise the counter CSVs written by tools/pmc_profile.sh: per-kernel counter totals for the trace kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if "trace_" not in k or "kernel" not in k:
                    continue
                name = row["Counter_Name"]
                tot[k][name] += float(row["Counter_Value"])
                calls[k][name] += 1
    for k in tot:
        print(k)
        for name in sorted(tot[k]):
            n = calls[k][name]
            print(f"  {name:28s} total={tot[k][name]:.6g}  dispatches={n}  per_dispatch={tot[k][name] / max(n, 1):.6g}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
