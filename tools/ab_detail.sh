#!/bin/bash
# one bench line per library: value, kernel times, deferred fractions (same box, same call)
for lib in "$@"; do
  CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --pairs ${AB_PAIRS:-16000000} --steps 25 --warmup 10 --cpu-sample 0 --no-copy-probe --tier-pairs 0 $AB_FLAGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['kernel_ms_avg_each'], 'dp', d['exact_dp_fraction'], 'refl', d['refiltered_fraction'])"
done
