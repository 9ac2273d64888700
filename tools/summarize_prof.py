#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel count / average duration from the
kernel trace, and per-kernel average FETCH_SIZE / WRITE_SIZE from the two PMC passes.
FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads (MI355X_MICROARCH.md, HBM): the corrected read traffic is 2 x FETCH_SIZE."""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(d):
    dur = defaultdict(list)
    for r in rows(os.path.join(d, "trace", "**", "*kernel_trace.csv")):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    pmc = {"fetch": defaultdict(list), "write": defaultdict(list)}
    for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for r in rows(os.path.join(d, kind, "**", "*counter_collection.csv")):
            if r.get("Counter_Name") == cname:
                pmc[kind][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    print("| kernel | launches | avg us | min us | total ms | FETCH_SIZE KB/launch (raw) | read MB/launch (x2 gfx950) | WRITE_SIZE KB/launch |")
    print("|---|---|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        # skip warm-up outliers: use the last 60 % of launches for the average
        tail = v[int(len(v) * 0.4):] or v
        f = pmc["fetch"].get(k, [])
        w = pmc["write"].get(k, [])
        ft = f[int(len(f) * 0.4):] or f
        wt = w[int(len(w) * 0.4):] or w
        fa = sum(ft) / len(ft) if ft else float("nan")
        wa = sum(wt) / len(wt) if wt else float("nan")
        print("| %s | %d | %.2f | %.2f | %.3f | %.1f | %.2f | %.1f |" % (
            k, len(v), sum(tail) / len(tail), min(v), sum(v) / 1e3, fa, 2 * fa * 1024 / 1e6, wa))
        if k.startswith("k_fast") and ft and wt:
            # machine-readable copy for bench.py's roofline.traffic (bytes per launch, FETCH_SIZE doubled)
            import json
            with open(os.path.join(d, "traffic.json"), "w") as fh:
                extra = {}
                try:      # VALU wave-instructions and clock of the same kernel from the SQ / GRBM passes (tools/r03_profile.sh)
                    with open(os.path.join(d, "fast_valu.json")) as vf:
                        extra = json.load(vf)
                except (OSError, ValueError):
                    pass
                json.dump({**extra, "kernel": "k_fast", "fetch_size_kb_raw": fa, "write_size_kb": wa,
                           "traffic_bytes_per_launch": int(2 * fa * 1024 + wa * 1024), "avg_launch_us": sum(tail) / len(tail),
                           "launches": len(v), "batch": 1024, "contexts": 1,
                           "source": "profiles/%s/summary.md (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate "
                                     "passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)"
                                     % os.path.basename(os.path.normpath(d)).replace("prof_", "")}, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
