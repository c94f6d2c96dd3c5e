#!/bin/bash
# Kernel timeline of the pipelined bench (rocprofv3 kernel trace): start/end of every launch relative to the
# first, so the idle time between consecutive scan kernels and the overlap with the resolve kernels show.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tl
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-copy-probe "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
prev_scan_end = None
for r in rows:
    name = r["Kernel_Name"]
    kind = "scan" if "0>(" in name or ", 0>" in name else "resolve" if "trim_kernel" in name else name[:28]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "trim_kernel" not in name and "fill" not in name.lower():
        continue
    if t0 is None:
        t0 = s
    gap = ""
    if kind == "scan":
        if prev_scan_end is not None:
            gap = "  idle since previous scan kernel %.1f us" % ((s - prev_scan_end) / 1e3)
        prev_scan_end = e
    print("%-10s q%-3s start %9.1f us  dur %8.1f us%s" % (kind, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, gap))
PY
rm -rf $OUT/trace
