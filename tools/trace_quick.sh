#!/bin/bash
# Quick per-kernel timing of the bench command (rocprofv3 kernel trace): gpurun_out/tq/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tq
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-copy-probe "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/trace
tail -1 $OUT/trace.log | cut -c1-400
cut -c1-200 $OUT/kernel_stats.csv | head -8
