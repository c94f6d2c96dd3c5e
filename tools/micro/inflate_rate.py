#!/usr/bin/env python3
"""How fast does the host turn .fastq.gz into text, without anything behind it?  GzipSource.blocks() on one file and on
two files at once (the pool is shared, as in a run), multi-member and single-member input.
    python3 tools/micro/inflate_rate.py [pairs]"""
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from cutseq_amd import codec, fastq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
work = Path("/dev/shm/cutseq_inflate")
work.mkdir(exist_ok=True)
subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
for m in (1, 2):  # single-member copies
    with open(work / f"single_R{m}.fastq.gz", "wb") as out:
        p1 = subprocess.Popen(["gzip", "-dc", str(work / f"syn_R{m}.fastq.gz")], stdout=subprocess.PIPE)
        p2 = subprocess.Popen(["gzip", "-1"], stdin=p1.stdout, stdout=out)
        p2.wait()
        p1.wait()


def drain(path, pool, box):
    src = codec.GzipSource(str(path), pool, fastq.ARENA.take, fastq.ARENA.give)
    tot = 0
    for arr, nbytes in src.blocks():
        tot += nbytes
        fastq.ARENA.give(arr)
    box.append((tot, dict(src.stats)))
    src.close()


pool = fastq._pool()
print(f"pool threads: {fastq.pool_size()}")
for kind in ("syn", "single"):
    files = [work / f"{kind}_R1.fastq.gz", work / f"{kind}_R2.fastq.gz"]
    for use in (files[:1], files):
        for rep in range(2):
            box = []
            t0 = time.perf_counter()
            ts = [threading.Thread(target=drain, args=(f, pool, box)) for f in use]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
            tot = sum(b[0] for b in box)
            print(f"{kind:6s} {len(use)} file(s): {tot / dt / 1e9:6.2f} GB/s of text ({dt:.3f} s) { {k: (round(v, 3) if isinstance(v, float) else v) for k, v in box[0][1].items()} }", flush=True)
import shutil
shutil.rmtree(work, ignore_errors=True)
