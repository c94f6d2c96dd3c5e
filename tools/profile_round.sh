#!/bin/bash
# One GPU-box call that refreshes everything under profiles/ for the current kernel:
#   bench lines (config 3 + config 2), rocprofv3 kernel-trace stats of the same bench command,
#   the PMC passes (tools/pmc.sh) and the tier K/T/E table.  Outputs land in gpurun_out/round/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 200 python3 bench.py --workload config2 --pairs 10000000 --steps 20 --cpu-sample 1000000 > $OUT/bench_config2.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); head -30 "$t" > $OUT/kernel_trace_head.csv
rm -rf $OUT/trace
bash tools/pmc.sh > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc.log; exit 1; }
timeout -k 10 400 python3 tools/tiers.py > $OUT/tiers.json 2> $OUT/tiers.err || { echo "tiers failed"; tail -5 $OUT/tiers.err; exit 1; }
cat $OUT/bench.json; cat $OUT/kernel_stats.csv | head -5
