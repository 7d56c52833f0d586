#!/usr/bin/env python3
"""Fold the FETCH_SIZE and WRITE_SIZE passes of tools/profile_bench.sh into one JSON:
   python tools/summarize_bench_traffic.py <fetch dir> <write dir> <bench line json> <command text>
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section) and is doubled; WRITE_SIZE is exact for these stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
                per[row["Dispatch_Id"]] += float(row["Counter_Value"])
        for did, v in per.items():
            acc[names[did]].append(v)
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
line = json.loads([ln for ln in open(sys.argv[3]) if ln.startswith("{")][-1])
out = {"command": sys.argv[4] if len(sys.argv) > 4 else "",
       "unit_note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced "
                    "read (MI355X_MICROARCH.md, HBM) and is doubled here",
       "workload": {"rows": line["config"]["rows"], "cols": line["config"]["cols"], "dtype": line["dtype"]},
       "timed_kernel": line["roofline"]["kernel"],
       "timed_kernel_ms_hip_events_this_run": line["roofline"]["kernel_ms"],
       "kernels": {}}
for name in sorted(set(fetch) | set(write)):
    if "mhlik::" not in name:
        continue
    f, w = fetch.get(name, []), write.get(name, [])
    if len(f) < 3:  # set-up kernels (data fill): not part of a step
        continue
    fm = sum(f) / len(f) if f else 0.0
    wm = sum(w) / len(w) if w else 0.0
    out["kernels"][name] = {"FETCH_SIZE_KiB_mean": fm, "launches_FETCH_SIZE": len(f), "WRITE_SIZE_KiB_mean": wm,
                            "launches_WRITE_SIZE": len(w), "hbm_bytes_per_launch_corrected": 2 * fm * 1024 + wm * 1024}
print(json.dumps(out, indent=1))
