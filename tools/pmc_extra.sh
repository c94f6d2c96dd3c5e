#!/bin/bash
# extra counter passes (instruction cache, branches, waits) for the bench kernels, three counters per pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcx
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --pairs 16000000 --serial --steps 3 --warmup 1 --cpu-sample 0 --no-copy-probe --tier-pairs 0 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run if SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM
run wt SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
run lv SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcx/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trim_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][-22:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{k[0]:24s} {k[1]:32s} {sum(v)/len(v):.4g}")
PY
