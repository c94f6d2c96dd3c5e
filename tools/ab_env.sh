#!/bin/bash
# A/B of one environment switch on ONE box in ONE call: tools/ab_env.sh "VAR=a" "VAR=b" [rounds] -> M pairs/s per run, alternating
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for kv in "$A" "$B"; do
    v=$(env $kv python3 bench.py --pairs ${AB_PAIRS:-16000000} --steps 25 --warmup 10 --cpu-sample 0 --no-copy-probe --tier-pairs 0 $AB_FLAGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg_each'], d.get('parity_error'))")
    echo "$kv $v"
  done
done
