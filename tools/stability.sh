#!/bin/bash
# the default bench command a few times: is the pipelined rate stable from run to run?
for i in 1 2 3 4; do timeout -k 10 200 python3 bench.py --cpu-sample 0 --no-copy-probe "$@" | python3 -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg_each'])"; done
