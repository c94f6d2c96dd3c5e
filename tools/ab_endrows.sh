#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" tools/ab/libab_ENDROWS.so tools/ab/libab_VERDICT.so; do
  OUT=gpurun_out/pmc_ablate; rm -rf $OUT; mkdir -p $OUT
  if [ -n "$lib" ]; then export CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -- python3 tools/ablate.py 2000000 > $OUT/log.txt 2>&1
  echo "== $lib"; python3 tools/ablate_summary.py | grep -E "full |only_5|only_3"
done
