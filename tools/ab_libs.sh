#!/bin/bash
# A/B of kernel builds on ONE box in ONE call: tools/ab_libs.sh libA.so libB.so [rounds] -> M pairs/s per run, alternating
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for lib in $A $B; do
    v=$(CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --pairs ${AB_PAIRS:-16000000} --steps 25 --warmup 10 --cpu-sample 0 --no-copy-probe --tier-pairs 0 $AB_FLAGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg_each'])")
    echo "$lib $v"
  done
done
