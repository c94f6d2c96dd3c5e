#!/bin/bash
# Register / scratch / occupancy figures of the scan (Li0) and resolve (Li1) kernels of the coded, non-wide plan,
# and the biggest basic blocks of the scan kernel (tools/isa_blocks.py).  Cross-compiles: no GPU needed.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
  ${KRES_FLAGS} -o /tmp/isa/x.s cutseq_amd/csrc/cutseq_hip.hip 2>&1 | grep -A9 "trim_kernelILb1ELb0ELi[01]" |
  grep -E "Function Name|VGPRs|Scratch|Occupancy|Spill" | sed 's/.*remark: //; s/ \[-Rpass.*//'
python3 tools/isa_blocks.py /tmp/isa/x.s _ZN5csdev11trim_kernelILb1ELb0ELi0EEEvNS_5KArgsE ${1:-150}
