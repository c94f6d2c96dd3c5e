#!/bin/bash
# the plain-text CLI path in fresh processes, N times per setting: does it ever crash? (intermittent SIGSEGV hunt)
W=/dev/shm/cutseq_hunt; mkdir -p $W
python3 tools/make_fastq.py 4000000 $W/syn > /dev/null 2>&1
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from cutseq_amd import codec, fastq
for m in (1,2):
    src = codec.GzipSource(f"/dev/shm/cutseq_hunt/syn_R{m}.fastq.gz", None, fastq.ARENA.take, fastq.ARENA.give)
    with open(f"/dev/shm/cutseq_hunt/plain_R{m}.fastq","wb") as dst:
        for arr, nb in src.blocks():
            dst.write(memoryview(arr)[:nb]); fastq.ARENA.give(arr)
    src.close()
PY
mkdir -p gpurun_out; for setting in ${SETTINGS:-""}; do
  bad=0
  for i in $(seq ${N:-12}); do
    env $setting timeout -k 5 60 python3 -X faulthandler -m cutseq_amd.run -A TAKARAV3 --trim-polyA $W/plain_R1.fastq $W/plain_R2.fastq -o $W/o1.fastq $W/o2.fastq -s $W/s1.fastq $W/s2.fastq > $W/log.txt 2>&1 || { bad=$((bad+1)); cp $W/log.txt gpurun_out/crash_$bad.txt; }
  done
  echo "setting '$setting': $bad crashes of ${N:-12}"
done
rm -rf $W
