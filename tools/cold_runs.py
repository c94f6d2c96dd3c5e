#!/usr/bin/env python3
"""What a user sees: the CLI as a FRESH process per run (interpreter, HIP, engines, page-locked buffers all included),
per input / output form and block size.  python3 tools/cold_runs.py [pairs] [block sizes ...]"""
import json
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
blocks = [int(x) for x in sys.argv[2:]] or [0]
work = Path("/dev/shm/cutseq_cold")
shutil.rmtree(work, ignore_errors=True)
work.mkdir(parents=True)
subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
for m in (1, 2):
    with open(work / f"plain_R{m}.fastq", "wb") as out:
        subprocess.run(["gzip", "-dc", str(work / f"syn_R{m}.fastq.gz")], stdout=out, check=True)
    head = subprocess.Popen(["head", "-n", str(4 * (n // 2)), str(work / f"plain_R{m}.fastq")], stdout=subprocess.PIPE)
    with open(work / f"single_R{m}.fastq.gz", "wb") as out:
        subprocess.run(["gzip", "-1"], stdin=head.stdout, stdout=out, check=True)
    head.wait()
forms = {
    "plain->plain": ([f"{work}/plain_R1.fastq", f"{work}/plain_R2.fastq", "-o", f"{work}/o1.fastq", f"{work}/o2.fastq", "-s", f"{work}/s1.fastq", f"{work}/s2.fastq"], n),
    "plain->gz": ([f"{work}/plain_R1.fastq", f"{work}/plain_R2.fastq", "-O", f"{work}/out"], n),
    "gz->gz": ([f"{work}/syn_R1.fastq.gz", f"{work}/syn_R2.fastq.gz", "-O", f"{work}/out"], n),
    "single-member gz->gz": ([f"{work}/single_R1.fastq.gz", f"{work}/single_R2.fastq.gz", "-O", f"{work}/out"], n // 2),
}
res = {}
for b in blocks:
    env = dict(os.environ)
    if b:
        env["CUTSEQ_CHUNK_READS"] = str(b)
    for name, (argv, pairs) in forms.items():
        best = None
        for rep in range(2):
            for f in list(work.glob("o[12].fastq")) + list(work.glob("s[12].fastq")) + list(work.glob("out*")):
                f.unlink()
            t0 = time.perf_counter()
            subprocess.run([sys.executable, "-m", "cutseq_amd.run", "-A", "TAKARAV3", "--trim-polyA"] + argv, check=True,
                           env=env, cwd=str(ROOT), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[f"{name} @ {b or 'auto'}"] = {"seconds": round(best, 3), "M_pairs_per_s": round(pairs / best / 1e6, 2)}
        print(f"{name:24s} block {b or 'auto':>7}: {best:.3f} s = {pairs / best / 1e6:.2f} M pairs/s", flush=True)
print(json.dumps(res))
shutil.rmtree(work, ignore_errors=True)
