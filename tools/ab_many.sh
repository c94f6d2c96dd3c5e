#!/bin/bash
# several library builds on one box in one call, alternating, R rounds: tools/ab_many.sh R lib1.so lib2.so ...
R=$1; shift
for i in $(seq $R); do
  for lib in "$@"; do
    v=$(CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --steps 25 --warmup 3 --cpu-sample 0 --no-copy-probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg_each'])")
    echo "$lib $v"
  done
done
