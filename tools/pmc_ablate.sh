#!/bin/bash
# dynamic instruction counts per ablation variant (7 launches each, in tools/ablate.py order)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ablate
rm -rf $OUT; mkdir -p $OUT
# three counters only: with the seven-counter set this script used first, rocprofv3 reported ~1.4x too
# many instructions for every variant (checked against a two-counter pass of the same launches)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -- python3 tools/ablate.py 2000000 > $OUT/log.txt 2>&1
echo rc=$?
