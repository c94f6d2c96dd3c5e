#!/bin/bash
# The chunk decoder alone, 1 .. 32 threads (is a single-member run bound by the decoder, or by the memory behind it?).
# Optional argument: another pinflate.c (an older revision's: git show REV:cutseq_amd/csrc/pinflate.c > file) to measure beside it.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
W=/dev/shm/pinflate_bench; mkdir -p $W
B=$(mktemp -d -p "$PWD" .pib.XXXX)  # (binaries: /dev/shm is mounted noexec on the GPU boxes)
python3 tools/make_fastq.py 500000 $W/syn > /dev/null 2>&1
gzip -dc $W/syn_R1.fastq.gz | gzip -1 > $W/single.gz
gcc -O3 -std=gnu11 -w -o $B/now tools/micro/pinflate_bench.c cutseq_amd/csrc/pinflate.c -lpthread
if [ -n "$1" ]; then gcc -O3 -std=gnu11 -w -x c -o $B/old tools/micro/pinflate_bench.c "$1" -lpthread; fi
for t in 1 1 4 8 16 32; do
  echo -n "now  "; $B/now $W/single.gz $t
  [ -x $B/old ] && { echo -n "old  "; $B/old $W/single.gz $t; }
done
grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
rm -rf $W $B
