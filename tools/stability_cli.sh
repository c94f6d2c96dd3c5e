#!/bin/bash
# The CLI's main forms in fresh processes, N times each: same output every time?  (flaky races show as a differing
# checksum or a non-zero exit; outputs are compared decompressed)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
W=/dev/shm/cutseq_stab; rm -rf $W; mkdir -p $W gpurun_out
N=${N:-6}
python3 tools/make_fastq.py ${PAIRS:-1500000} $W/syn > /dev/null 2>&1 || exit 1
for m in 1 2; do gzip -dc $W/syn_R$m.fastq.gz > $W/plain_R$m.fastq; gzip -1 < $W/plain_R$m.fastq > $W/single_R$m.fastq.gz; done
sum_of() { for f in "$@"; do if [[ $f == *.gz ]]; then gzip -dc "$f"; else cat "$f"; fi; done | md5sum | cut -c1-12; }
run() {  # name, then the command line behind "cutseq"
  name=$1; shift
  ref=""; bad=0
  for i in $(seq $N); do
    rm -f $W/out*
    timeout -k 5 120 python3 -m cutseq_amd.run -A TAKARAV3 --trim-polyA "$@" > $W/log.txt 2>&1 || { bad=$((bad+1)); cp $W/log.txt gpurun_out/stab_${name}_$bad.txt; continue; }
    s=$(sum_of $W/out*R1* $W/out*R2*)
    if [ -z "$ref" ]; then ref=$s; elif [ "$s" != "$ref" ]; then bad=$((bad+1)); echo "  $name run $i: checksum $s != $ref"; fi
  done
  echo "$name: $bad bad of $N (checksum $ref)"
}
run plain_plain $W/plain_R1.fastq $W/plain_R2.fastq -o $W/out_R1.fastq $W/out_R2.fastq -s $W/outs_R1.fastq $W/outs_R2.fastq
run plain_gz $W/plain_R1.fastq $W/plain_R2.fastq -O $W/out
run gz_gz $W/syn_R1.fastq.gz $W/syn_R2.fastq.gz -O $W/out
run single_gz $W/single_R1.fastq.gz $W/single_R2.fastq.gz -O $W/out
CUTSEQ_DEVICES=0,0 run ranks2 --ranks 2 $W/syn_R1.fastq.gz $W/syn_R2.fastq.gz -O $W/out
CUTSEQ_DEVICES=0,0 run two_engines $W/plain_R1.fastq $W/plain_R2.fastq -O $W/out
rm -rf $W
