#!/bin/bash
# rocprofv3 counter passes for the bench command (separate passes, kernel-trace only, at most three SQ
# counters per pass: larger sets over-report on this part).  Output: gpurun_out/pmc/<pass>/...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 20 --warmup 10 --cpu-sample 0 --no-copy-probe --no-piece-check --tier-pairs 0 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU
run sq4 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD
run sq5 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run grbm GRBM_GUI_ACTIVE
find $OUT -name "*counter_collection.csv" | head -20
