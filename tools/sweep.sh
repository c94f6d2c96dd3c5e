#!/bin/bash
# GPU tuning sweep: LDS column bytes per wave / tile rows vs kernel time (2M pairs)
for cfg in "2304 64" "2304 128" "2304 256"; do
  set -- $cfg
  echo "== COL_BYTES=$1 TILE_ROWS=$2"
  CUTSEQ_COL_BYTES=$1 CUTSEQ_TILE_ROWS=$2 timeout -k 10 200 python tools/ablate.py 2000000 2>&1 | grep -E "^full |no_polyA|only_5|only_3|only_cuts"
done
