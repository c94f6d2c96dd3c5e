#!/bin/bash
# rocprofv3 kernel statistics of the text path (FASTQ text in -> finished text / gzip members out, all on the device)
# through the product CLI on 8 M synthetic pairs: plain -> plain and gz -> gz.  Outputs under gpurun_out/text/;
# copy text_plain_kernel_stats.csv / text_gz_kernel_stats.csv to profiles/<round>_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-8000000}
W=/dev/shm/cutseq_textprof
OUT=gpurun_out/text
rm -rf $OUT $W && mkdir -p $OUT $W
python3 tools/make_fastq.py $N $W/syn > /dev/null 2>&1 || { echo "make_fastq failed"; exit 1; }
python3 - <<PY || exit 1
import sys; sys.path.insert(0, ".")
from cutseq_amd import codec, fastq
for m in (1, 2):
    src = codec.GzipSource("$W/syn_R%d.fastq.gz" % m, None, fastq.ARENA.take, fastq.ARENA.give)
    with open("$W/plain_R%d.fastq" % m, "wb") as dst:
        for arr, nbytes in src.blocks():
            dst.write(memoryview(arr)[:nbytes]); fastq.ARENA.give(arr)
    src.close()
PY
run() {  # name, inputs..., -- outputs...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 -m cutseq_amd.run -A TAKARAV3 --trim-polyA "$@" > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/text_${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
run plain $W/plain_R1.fastq $W/plain_R2.fastq -o $W/o1.fastq $W/o2.fastq -s $W/s1.fastq $W/s2.fastq
run gz $W/syn_R1.fastq.gz $W/syn_R2.fastq.gz -o $W/o1.fastq.gz $W/o2.fastq.gz -s $W/s1.fastq.gz $W/s2.fastq.gz
rm -rf $W
for n in plain gz; do echo "== $n"; cut -c1-150 $OUT/text_${n}_kernel_stats.csv; done
