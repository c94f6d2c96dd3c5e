#!/bin/bash
# builds the second device-inflate probe, makes a single-member gzip -1 FASTQ of N pairs' R1 and runs the probe on it
set -o pipefail
cd "$GRAFT_REPO_ROOT"
N=${1:-1000000}
W=/dev/shm/cutseq_gpuinf
rm -rf $W && mkdir -p $W
hipcc -O3 --offload-arch=gfx950 -w -o /tmp/gpu_inflate_probe2 tools/micro/gpu_inflate_probe2.hip -lz || exit 1
python3 tools/make_fastq.py $N $W/syn > /dev/null 2>&1 || exit 1
gzip -dc $W/syn_R1.fastq.gz | gzip -1 > $W/single.gz
ls -la $W/single.gz
timeout -k 5 500 /tmp/gpu_inflate_probe2 $W/single.gz ${CHUNKS:-1024 256 64}
rm -rf $W
