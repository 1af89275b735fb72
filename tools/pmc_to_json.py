#!/usr/bin/env python3
"""Turn the counter CSVs of tools/pmc_passes.sh into the small JSON that bench.py's roofline object reads
(profiles/*_pmc.json): per-ray VALU wave-instructions, VALU lane occupancy and HBM bytes of the dominant trace kernel on one
workload.  The rays per launch come from the bench.py JSON line each pass printed into its log.

    python3 tools/pmc_to_json.py PMC_DIR OUT.json [note]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpupathtracer_amd.provenance import KERNEL_SOURCES, kernel_source_hash


def main(root, out_path, note=""):
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if "trace_" not in k or "kernel" not in k:
                    continue
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]] += 1
    bench = None
    for log in sorted(glob.glob(os.path.join(root, "*.log"))):
        for line in open(log, errors="replace"):
            line = line.strip()
            if line.startswith("{") and '"metric"' in line:
                bench = json.loads(line)
    if bench is None:
        raise SystemExit("no bench.py JSON line found in the pass logs")
    # the un-instrumented instantiation is the one the timed frames launch (the first warm-up frame runs the <true, ...> twin)
    kernels = [k for k in tot if "<false" in k] or list(tot)
    kernel = max(kernels, key=lambda k: tot[k].get("SQ_WAVE_CYCLES", 0.0))
    per = {n: tot[kernel][n] / max(calls[kernel][n], 1) for n in tot[kernel]}  # per dispatch
    rays = float(bench["roofline"]["rays_per_launch"])
    wl = bench["config"]["workload"]
    import re
    m = re.search(r"(\d+)x(\d+), (\d+) bounces, (\d+) spp, camera=(\w+), trace=(\w+)", wl)
    fetch = per.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0  # KB; gfx950 reports half of a wide read (MI355X guide, HBM section)
    write = per.get("WRITE_SIZE", 0.0) * 1024.0
    short = re.search(r"(trace_\w+<[^>]*>)", kernel).group(1)
    doc = {
        "source": "rocprofv3 --kernel-trace --pmc <group> (one group per pass, tools/pmc_passes.sh) on bench.py, MI355X; " + note,
        "workload": {"width": int(m.group(1)), "height": int(m.group(2)), "bounces": int(m.group(3)), "spp": int(m.group(4)),
                     "n_gpus": bench["n_gpus"], "camera": m.group(5), "trace": m.group(6), "scene": bench["config"].get("scene", "c2")},
        "kernel": short,
        # the sources this kernel was built from (bench.py withholds these numbers from a tree whose hash differs)
        "kernel_source_hash": kernel_source_hash(), "kernel_source_files": ["gpupathtracer_amd/csrc/" + n for n in KERNEL_SOURCES],
        "rays_per_launch": int(rays),
        "kernel_ms_per_launch_profiled": bench["roofline"]["kernel_ms_per_launch"],
        "valu_insts_per_ray": round(per.get("SQ_INSTS_VALU", 0.0) / rays, 4),
        "salu_insts_per_ray": round(per.get("SQ_INSTS_SALU", 0.0) / rays, 4),
        "lds_insts_per_ray": round(per.get("SQ_INSTS_LDS", 0.0) / rays, 4),
        "vmem_rd_insts_per_ray": round(per.get("SQ_INSTS_VMEM_RD", 0.0) / rays, 4),
        "lane_occupancy": round(per.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(64.0 * per.get("SQ_ACTIVE_INST_VALU", 1.0), 1.0), 4),
        "wait_any_frac": round(per.get("SQ_WAIT_ANY", 0.0) / max(per.get("SQ_WAVE_CYCLES", 1.0), 1.0), 4),
        "lds_array_busy_cycles_per_ray": round(per.get("SQ_LDS_IDX_ACTIVE", 0.0) / rays, 3),
        "FETCH_SIZE_KB": per.get("FETCH_SIZE"), "WRITE_SIZE_KB": per.get("WRITE_SIZE"),
        "correction": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read), WRITE_SIZE as reported",
        "hbm_bytes_per_launch": int(fetch + write),
        "hbm_bytes_per_ray": round((fetch + write) / rays, 4),
        "effective_clock_GHz": round(per["GRBM_GUI_ACTIVE"] / 8.0 / (bench["roofline"]["kernel_ms_per_launch"] * 1e-3) / 1e9, 3) if "GRBM_GUI_ACTIVE" in per else None,
        "counters_per_launch": {k: per[k] for k in sorted(per)},
    }
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("kernel", "valu_insts_per_ray", "lane_occupancy", "hbm_bytes_per_launch", "effective_clock_GHz")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
