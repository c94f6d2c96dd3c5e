#!/bin/bash
# One GPU-box call that refreshes everything under profiles/ for the current kernels:
#   rocprofv3 kernel-trace stats of the bench command, the PMC passes (tools/pmc.sh) and their summary,
#   bench lines (config 3 with the fresh counter file, config 2 / 4 / 5), tier T and the end-to-end rates.
# Outputs land in gpurun_out/round/; copy them to profiles/<round>_* afterwards.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}
OUT=gpurun_out/round
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 10 --cpu-sample 0 --no-copy-probe --no-piece-check --tier-pairs 0 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); head -30 "$t" > $OUT/kernel_trace_head.csv
rm -rf $OUT/trace
bash tools/pmc.sh > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc.log; exit 1; }
mkdir -p profiles && python3 tools/summarize_pmc.py $TAG > $OUT/pmc_summary.log 2>&1 || { echo "summary failed"; cat $OUT/pmc_summary.log; exit 1; }
cp profiles/${TAG}_pmc_summary.json $OUT/pmc_summary.json
timeout -k 10 400 python3 bench.py --traffic-json $OUT/pmc_summary.json > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 200 python3 bench.py --serial --cpu-sample 0 --tier-pairs 0 > $OUT/bench_serial.json 2>> $OUT/bench.err || exit 1
timeout -k 10 200 python3 bench.py --workload config2 --pairs 10000000 --steps 20 --cpu-sample 1000000 > $OUT/bench_config2.json 2>> $OUT/bench.err || exit 1
timeout -k 10 200 python3 bench.py --workload config4 --pairs 16000000 --cpu-sample 500000 > $OUT/bench_config4.json 2>> $OUT/bench.err || exit 1
timeout -k 10 200 python3 bench.py --workload config5 --pairs 16000000 > $OUT/bench_config5.json 2>> $OUT/bench.err || exit 1
timeout -k 10 400 python3 tools/tiers.py > $OUT/tiers.json 2> $OUT/tiers.err || { echo "tiers failed"; tail -5 $OUT/tiers.err; exit 1; }
timeout -k 10 400 python3 tools/e2e_bench.py 8000000 > $OUT/e2e.json 2> $OUT/e2e.err || { echo "e2e failed"; tail -5 $OUT/e2e.err; exit 1; }
bash tools/text_profile.sh > $OUT/text_profile.log 2>&1 || { echo "text profile failed"; tail -5 $OUT/text_profile.log; exit 1; }
cp gpurun_out/text/text_plain_kernel_stats.csv gpurun_out/text/text_gz_kernel_stats.csv $OUT/
cat $OUT/bench.json; head -4 $OUT/kernel_stats.csv | cut -c1-160; cat $OUT/e2e.json
