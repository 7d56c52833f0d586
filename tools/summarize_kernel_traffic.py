#!/usr/bin/env python3
"""HBM traffic per launch of every likelihood kernel, from the FETCH_SIZE and WRITE_SIZE passes of
tools/profile_kernels.sh (one rocprofv3 --pmc run each over tools/run_kernels.py).

usage: summarize_kernel_traffic.py <fetch_dir> <write_dir> <out.json>

Counters are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM section) and is doubled.  The first dispatch of a mixed-precision kernel after a reset
sees every block bright (the chain start, once per chain) and is listed on its own; "steady" is the mean of the
other dispatches."""
import csv
import glob
import json
import sys

N = 10_000_000
ALGO = {  # algorithmic HBM bytes per steady launch at tools/run_kernels.py's sizes
    "lik_tile_kernel<double": N * 64 * 8,
    "lik_tile_kernel<float": N * 64 * 4,
    "mp_tile_kernel": N * 64 * 4 + N * 64 * 8 // 16,      # float pass + 1/16 of the rows again in double
    "lik_batched_f32_kernel": N * 128 * 4,
    "lik_batched_x3_kernel": N * 128 * 4,
    "lik_batched_f64_kernel": N * 128 * 8,
    "mpb_dark_kernel": N * 128 * 4,
    "mpb_dark_x3_kernel": N * 128 * 4,
    "mpb_fix_dense_kernel": N * 128 * 12 // 16,            # one phase for all chains: 1/16 of the rows, float + double
    "mpb_fix_kernel": N * 128 * 12,                        # 64 chains over 16 phases: every row is some chain's
}


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024))
    return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}


def main():
    fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for name, f in fetch.items():
        if "mhlik::" not in name:
            continue
        w = write.get(name, [0.0] * len(f))
        mp = any(t in name for t in ("mp_tile", "mpb_"))
        runs = [f, w]
        first = None
        if mp and len(f) > 1:
            # a reset precedes dispatch 0 of each group of calls: those see every block bright
            big = [i for i in range(len(f)) if f[i] > 2 * min(f)]
            if big:
                first = {"dispatches": len(big), "fetch_bytes_corrected": sum(2 * f[i] for i in big) / len(big),
                         "write_bytes": sum(w[i] for i in big) / len(big)}
                runs = [[f[i] for i in range(len(f)) if i not in big], [w[i] for i in range(len(w)) if i not in big]]
        nf = len(runs[0])
        e = {"launches": nf, "FETCH_SIZE_bytes_raw": sum(runs[0]) / nf, "WRITE_SIZE_bytes": sum(runs[1]) / max(len(runs[1]), 1)}
        e["hbm_bytes_per_launch_corrected"] = 2 * e["FETCH_SIZE_bytes_raw"] + e["WRITE_SIZE_bytes"]
        for key, a in ALGO.items():
            if "mhlik::" + key in name:
                e["algorithmic_bytes"] = a
                e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch_corrected"] / a
        if first:
            e["first_call_after_reset"] = first
        kernels[name] = e
    doc = {
        "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE --output-format csv -- python3 tools/run_kernels.py 4 "
                   "(separate passes, tools/profile_kernels.sh)",
        "unit_note": "FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide "
                     "coalesced reads, MI355X_MICROARCH.md HBM section). Mixed-precision kernels: the dispatch after a "
                     "reset evaluates every block in both precisions (the chain start, once per chain) and is given "
                     "separately; the per-launch figures are the steady iterations.",
        "workload": "N=10M; d=64 for the streaming and single-chain MP kernels, d=128 K=64 for the batched ones; MP: "
                    "64-row blocks, 1/16 of the dark blocks re-examined per call",
        "kernels": kernels,
    }
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    for k, e in kernels.items():
        if "traffic_over_algorithmic" in e:
            print("%-50s %8.3f GB  x%.3f  write %.1f MB" % (k.split("(")[0][-50:], e["hbm_bytes_per_launch_corrected"] / 1e9,
                                                           e["traffic_over_algorithmic"], e["WRITE_SIZE_bytes"] / 1e6))


if __name__ == "__main__":
    main()
