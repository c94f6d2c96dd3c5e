#!/usr/bin/env python3
"""VALU/SALU/LDS wave-instructions per 64-pair tile for every variant of tools/ablate.py (after tools/pmc_ablate.sh)."""
import collections
import csv
import glob
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
f = sorted(glob.glob("gpurun_out/pmc_ablate/*/*counter_collection.csv"))[-1]
rows = list(csv.DictReader(open(f)))
# dispatches in launch order; each variant = 7 launches x 2 kernels
disp = collections.OrderedDict()
for r in rows:
    if "trim_kernel" not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
names = ["full", "full_nofilter", "no_polyA", "only_5prime", "only_3prime", "only_poly", "only_cuts", "only_qtrim"]
per_variant = len(ids) // len(names)
print(f"{'variant':16s} {'VALU/tile':>10s} {'SALU/tile':>10s} {'LDS/tile':>9s}   (scan + resolve kernel, per 64-pair tile)")
for i, name in enumerate(names):
    chunk = ids[i * per_variant:(i + 1) * per_variant][-4:]  # the last two launches (2 kernels each)
    tot = collections.Counter()
    for d in chunk:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            tot[c] += disp[d].get(c, 0.0)
    tiles = n / 64 * 2  # two launches
    print(f"{name:16s} {tot['SQ_INSTS_VALU'] / tiles:10.0f} {tot['SQ_INSTS_SALU'] / tiles:10.0f} {tot['SQ_INSTS_LDS'] / tiles:9.0f}")
