#!/bin/bash
# dynamic instruction counts per ablation variant (7 launches each, in tools/ablate.py order)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ablate
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 tools/ablate.py 2000000 > $OUT/log.txt 2>&1
echo rc=$?
