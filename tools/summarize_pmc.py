#!/usr/bin/env python3
"""Summarise the rocprofv3 counter passes of tools/pmc.sh into profiles/<tag>_pmc_summary.json.

Per kernel (scan / resolve) and summed per launch (one launch = one scan + one resolve kernel)."""
import collections
import csv
import glob
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from cutseq_amd.build import kernel_source_hash  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def pairs_of_the_passes() -> int:
    """Pairs per launch as the profiled bench runs report it themselves (the JSON line each pass leaves in its log)."""
    seen = set()
    for log in glob.glob("gpurun_out/pmc/*.log"):
        for line in open(log, errors="replace"):
            if line.startswith("{") and "pairs_per_step_per_gpu" in line:
                seen.add(int(json.loads(line)["config"]["pairs_per_step_per_gpu"]))
    if len(seen) != 1:
        raise SystemExit(f"pairs per launch of the counter passes: {sorted(seen)} (expected exactly one value)")
    return seen.pop()


pairs = int(sys.argv[2]) if len(sys.argv) > 2 else pairs_of_the_passes()


def kernel_of(name: str) -> str:
    if "trim_kernel" not in name:
        return ""
    return "resolve" if name.rstrip(">(csdev::KArgs) ").endswith("1") or ", 1>" in name else "scan"


per = {"scan": {}, "resolve": {}}
for pass_dir in sorted(glob.glob("gpurun_out/pmc/*/")):
    fs = sorted(glob.glob(pass_dir + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        k = kernel_of(r["Kernel_Name"])
        if k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                dur[(k, r["Counter_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for (k, c), v in agg.items():
        per[k][c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
        if dur.get((k, c)):
            per[k][c]["mean_duration_ns"] = sum(dur[(k, c)]) / len(dur[(k, c)])
total = {}
for c in set(per["scan"]) | set(per["resolve"]):
    total[c] = sum(per[k][c]["mean_per_launch"] for k in per if c in per[k])
fetch, write = total["FETCH_SIZE"], total["WRITE_SIZE"]
# the clock the part held while the scan kernel ran: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md,
# "DVFS give-back"), the duration is the same dispatches' own, from the same pass
clock_hz, clock_src = None, None
grbm = per["scan"].get("GRBM_GUI_ACTIVE")
if grbm and grbm.get("mean_duration_ns"):
    clock_hz = grbm["mean_per_launch"] / 8.0 / (grbm["mean_duration_ns"] * 1e-9)
    clock_src = ("GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration of the scan kernel, same rocprofv3 pass "
                 f"({grbm['mean_per_launch']:.4g} / 8 / {grbm['mean_duration_ns'] / 1e6:.4f} ms)")
summary = {
    "kernel_source_sha256": kernel_source_hash(),
    "measured_clock_hz": clock_hz,
    "measured_clock_source": clock_src,
    "command": "rocprofv3 --pmc <counter set> --output-format csv -- python3 bench.py --steps 20 --warmup 10 --cpu-sample 0 "
               "--no-copy-probe (one pass per counter set of at most three SQ counters: tools/pmc.sh)",
    "kernels": {"scan": "csdev::trim_kernel<true,false,0>", "resolve": "csdev::trim_kernel<true,false,1>"},
    "pairs_per_launch": pairs,
    "per_kernel": per,
    "per_launch": total,
    "valu_insts_per_launch": total.get("SQ_INSTS_VALU"),
    "valu_insts_per_64_pair_tile": total.get("SQ_INSTS_VALU", 0) / (pairs / 64),
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
    "hbm_note": "FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 "
                "wide coalesced reads (the scattered 4-byte quality loads and the resolve kernel's row gather are "
                "uncalibrated: good to +-25 %), WRITE_SIZE taken as is",
    "algorithmic_bytes_per_launch": pairs * 616,
}
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "valu_insts_per_launch",
                                          "valu_insts_per_64_pair_tile")}))
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
    for kern in ("scan", "resolve"):
        if k in per[kern]:
            print(kern, k, "%.4g" % per[kern][k]["mean_per_launch"])
