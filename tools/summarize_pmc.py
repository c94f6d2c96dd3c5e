#!/usr/bin/env python3
"""Summarise the rocprofv3 counter passes of tools/pmc.sh into profiles/<tag>_pmc_summary.json."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
out = {}
for name in ["fetch", "write", "sq1", "sq2", "grbm"]:
    import os
    fs = sorted(glob.glob(f"gpurun_out/pmc/{name}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[-1])))
    agg = collections.defaultdict(list)
    for r in rows:
        if "trim_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
fetch, write = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
summary = {
    "command": "rocprofv3 --pmc <counter set> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 "
               "(one pass per counter set: tools/pmc.sh)",
    "kernel": "csdev::trim_kernel<true,false>",
    "pairs_per_launch": pairs,
    "counters": out,
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
    "hbm_note": "FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 "
                "wide coalesced reads (the scattered 4-byte quality loads are uncalibrated), WRITE_SIZE taken as is "
                "(64 MB = 8 M results x 8 B)",
    "algorithmic_bytes_per_launch": pairs * 616,
}
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}))
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
    if k in out:
        print(k, "%.4g" % out[k]["mean_per_launch"])
