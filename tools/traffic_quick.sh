#!/bin/bash
# HBM traffic of the bench command: two counter passes (FETCH_SIZE, WRITE_SIZE), KiB per kernel launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tr; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-copy-probe > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "trim_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][-30:]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(sys.argv[2], k, "%.0f MiB per launch"%(sum(v)/len(v)/1024))
PY
done
