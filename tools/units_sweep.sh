#!/bin/bash
# hand-out unit sizes under the pipelined bench: CUTSEQ_UNITS=big_shift,small_shift,big_pct (extra bench.py arguments pass through)
for u in ${UNITS:-"" "0,0,0" "1,0,50" "1,0,90" "2,0,50"}; do
  v=$(CUTSEQ_UNITS=$u python3 bench.py "$@" --cpu-sample 0 --no-copy-probe --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg_each'])")
  echo "units='$u' $v"
done
