cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import tiers
from pathlib import Path
from concurrent.futures import ThreadPoolExecutor
from cutseq_amd import workloads
n=4000000
b=workloads.make_batch("config3", n)
w=Path("/dev/shm/su"); w.mkdir(exist_ok=True)
with ThreadPoolExecutor(16) as pool: tiers.write_inputs(w,b,n,pool)
PY
for i in 1 2; do
CUTSEQ_PROFILE=1 python -m cutseq_amd.run -A TAKARAV3 --trim-polyA /dev/shm/su/plain_R1.fastq /dev/shm/su/plain_R2.fastq -o /dev/shm/su/o1.fastq.gz /dev/shm/su/o2.fastq.gz -s /dev/shm/su/s1.fastq.gz /dev/shm/su/s2.fastq.gz 2>&1 | grep -E "cutseq_phase|cutseq_profile" | cut -c1-1500
rm -f /dev/shm/su/o*.gz /dev/shm/su/s*.gz
done
rm -rf /dev/shm/su
