#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc run per kernel:
   python tools/summarize_pmc.py <dir with *_counter_collection.csv> [out.json] [command text]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    per_dispatch = defaultdict(lambda: defaultdict(float))
    names = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            did = row["Dispatch_Id"]
            names[did] = row["Kernel_Name"]
            per_dispatch[did][row["Counter_Name"]] += float(row["Counter_Value"])
    for did, counters in per_dispatch.items():
        for k, v in counters.items():
            acc[names[did]][k].append(v)
out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
for name, counters in acc.items():
    rec = {k: sum(v) / len(v) for k, v in counters.items()}
    rec["dispatches"] = max(len(v) for v in counters.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("GRBM_GUI_ACTIVE", 0) > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        rec["mfma_utilisation"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * rec["GRBM_GUI_ACTIVE"] / 8)
    out["kernels"][name] = rec
text = json.dumps(out, indent=1)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    open(sys.argv[2], "w").write(text + "\n")
print(text)
