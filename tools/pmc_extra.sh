#!/bin/bash
# extra counter passes (instruction cache, branches, waits) for the bench kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcx
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-copy-probe > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
run lv SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcx/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trim_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:32s} {sum(v)/len(v):.4g}")
PY
