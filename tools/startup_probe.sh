#!/bin/bash
# Where does a short CLI run spend its start-up?  2 M pairs as a fresh process, with the text path's profile.
cd "$GRAFT_REPO_ROOT"
D=/dev/shm/cutseq_startup; rm -rf $D; mkdir -p $D
python3 tools/make_fastq.py 2000000 $D/syn > /dev/null 2>&1
python3 - <<PY
import gzip, shutil
for m in (1, 2):
    with gzip.open("$D/syn_R%d.fastq.gz" % m, "rb") as s, open("$D/plain_R%d.fastq" % m, "wb") as d:
        shutil.copyfileobj(s, d, 1 << 24)
PY
for i in 1 2; do
  /usr/bin/time -f "plain->plain wall %e s" env CUTSEQ_PROFILE=1 python3 -X importtime -m cutseq_amd.run $D/plain_R1.fastq $D/plain_R2.fastq -A TAKARAV3 --trim-polyA -o $D/o1.fastq $D/o2.fastq -s $D/s1.fastq $D/s2.fastq 2> $D/err.txt
  grep "wall\|cutseq_profile" $D/err.txt | cut -c1-1500
  grep "import time" $D/err.txt | sort -t'|' -k2 -n | tail -6
  rm -f $D/o?.fastq $D/s?.fastq
done
/usr/bin/time -f "gz->gz wall %e s" python3 -m cutseq_amd.run $D/syn_R1.fastq.gz $D/syn_R2.fastq.gz -A TAKARAV3 --trim-polyA -O $D/gz 2> $D/err.txt; grep wall $D/err.txt
rm -rf $D
