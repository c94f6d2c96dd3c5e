#!/bin/bash
# tools/lean_probe.py for scan kernels built at 5 / 6 / 7 waves per SIMD (tools/ab/scan_w{5,6,7}.so, built in the container:
#   for w in 5 6 7; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DCS_SCAN_WAVES=$w -o tools/ab/scan_w$w.so cutseq_amd/csrc/cutseq_hip.hip; done)
for w in 5 6 7; do
  echo "== scan kernel at $w waves/SIMD"
  CUTSEQ_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/scan_w$w.so timeout -k 10 200 python3 tools/lean_probe.py 4000000 2>&1 | grep -v "^$"
done
