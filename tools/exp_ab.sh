#!/bin/bash
# timing-only experiment builds (tools/ab/exp*.so: pieces of the kernel disabled, results wrong) on the ablation plans
for lib in cutseq_amd/libcutseq_hip.so "$@"; do
  echo "== $lib"
  if [ -n "$FRACTION" ]; then export CS_ADAPTER_FRACTION=$FRACTION CS_PARTIAL_FRACTION=${PARTIAL:-0}; fi
  CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/ablate.py 4000000 2>&1 | grep "^only_3prime \|^only_5prime \|^full "
done
