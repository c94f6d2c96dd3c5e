#!/bin/bash
# A/B on one box: tools/ab/base.so vs tools/ab/new.so, alternating, kernel-only timings (tools/ablate.py)
for rep in 1 2 3; do
  for lib in base new; do
    echo "== $lib (rep $rep)"
    CUTSEQ_HIP_LIB=$PWD/tools/ab/$lib.so timeout -k 10 200 python tools/ablate.py ${1:-2000000} 2>/dev/null | grep -E "^full  |only_5prime|only_3prime" || exit 1
  done
done
