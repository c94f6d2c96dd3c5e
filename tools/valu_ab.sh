#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU per launch of the bench kernels for a list of library builds (same box, same call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  OUT=gpurun_out/valu_ab/$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT -- python3 bench.py --pairs 16000000 --serial --steps 3 --warmup 1 --cpu-sample 0 --no-copy-probe --tier-pairs 0 > $OUT/log.txt 2>&1
  python3 - $OUT $lib <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trim_kernel" in r["Kernel_Name"]:
            agg[(("scan" if "0>(" in r["Kernel_Name"] else "resolve"), r["Counter_Name"])].append(float(r["Counter_Value"]))
print(sys.argv[2], {f"{k[0]}.{k[1][9:]}": round(sum(v) / len(v) / 1e6, 1) for k, v in sorted(agg.items())})
PY
done
