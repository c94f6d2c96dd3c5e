#!/bin/bash
# pipelined bench under a few resolve-grid settings, next to the serial form
for w in ${WAVES:-2 3 4 5 6}; do
  echo "resolve waves/CU $w"; CUTSEQ_RESOLVE_WAVES=$w timeout -k 10 120 python3 bench.py --cpu-sample 0 --no-copy-probe | python3 -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_avg_each'])"
done
echo serial; timeout -k 10 120 python3 bench.py --serial --cpu-sample 0 --no-copy-probe | python3 -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_avg_each'])"
