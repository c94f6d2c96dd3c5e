#!/usr/bin/env python3
"""VALU / SALU / LDS wave-instructions of the scan and the resolve kernel for every variant of tools/ablate.py, in its
order (after tools/pmc_ablate.sh; the last of a variant's seven launches)."""
import collections
import csv
import glob

rows = []
for f in glob.glob("gpurun_out/pmc_ablate/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trim_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), "scan" if "0>(" in r["Kernel_Name"] else "resolve", r["Counter_Name"],
                         float(r["Counter_Value"])))
rows.sort()
by = collections.OrderedDict()
for d, k, c, v in rows:
    by.setdefault((d, k), {})[c] = v
seq = list(by.items())
names = ["full", "full_nofilter", "no_polyA", "no_qtrim", "no_5prime", "no_3prime", "no_cuts", "only_5prime", "only_3prime",
         "only_3prime_mo10", "only_5prime_mo3", "only_poly", "only_cuts", "only_qtrim"]
scan = [x for x in seq if x[0][1] == "scan"]
res = [x for x in seq if x[0][1] == "resolve"]
assert len(scan) == len(res) == 7 * len(names), (len(scan), len(res))
for i, nm in enumerate(names):
    s, r = scan[i * 7 + 6][1], res[i * 7 + 6][1]
    print(f"{nm:18s} scan VALU {s.get('SQ_INSTS_VALU', 0) / 1e6:8.1f}M SALU {s.get('SQ_INSTS_SALU', 0) / 1e6:7.1f}M "
          f"LDS {s.get('SQ_INSTS_LDS', 0) / 1e6:6.1f}M | resolve VALU {r.get('SQ_INSTS_VALU', 0) / 1e6:7.1f}M")
