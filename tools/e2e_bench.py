#!/usr/bin/env python3
"""Tier E through the product CLI, in-process, with start-up amortised.

  python tools/e2e_bench.py [pairs] [workdir]

Writes a synthetic paired data set once (plain text and multi-member gzip), then runs
``cutseq_amd.run.main`` on it: plain -> plain twice (the second run has warm buffers and page cache),
gz -> gz once.  Prints one JSON object.  Not the headline metric: bench.py reports the HBM-resident rate.
"""
import json
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    # the steady-state regime: blocks of 262 144 records, what inputs of 25 M pairs and more get (smaller inputs get
    # smaller blocks -- a fresh process spends more time page-locking big buffers than it saves: tools/cold_runs.py)
    os.environ.setdefault("CUTSEQ_CHUNK_READS", "262144")
    work = Path(sys.argv[2] if len(sys.argv) > 2 else "/dev/shm/cutseq_e2e")
    work.mkdir(parents=True, exist_ok=True)
    free = shutil.disk_usage(work).free
    need = n * 2 * 340 * 3
    if free < need:
        raise SystemExit(f"{work}: {free >> 20} MiB free, {need >> 20} MiB needed for {n} pairs")
    out = {"pairs": n, "workdir": str(work), "cpus": len(os.sched_getaffinity(0))}
    t0 = time.perf_counter()
    subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
    out["make_gz_s"] = round(time.perf_counter() - t0, 2)
    from cutseq_amd import codec, fastq, run as cli

    t0 = time.perf_counter()
    for m in (1, 2):  # plain copies of the same records
        src = codec.GzipSource(str(work / f"syn_R{m}.fastq.gz"), None, fastq.ARENA.take, fastq.ARENA.give)
        with open(work / f"plain_R{m}.fastq", "wb") as dst:
            for arr, nbytes in src.blocks():
                dst.write(memoryview(arr)[:nbytes])
                fastq.ARENA.give(arr)
        src.close()
    out["gunzip_s"] = round(time.perf_counter() - t0, 2)

    def run(tag, inputs, outputs):
        for old in work.glob("o[12].fastq"), work.glob("s[12].fastq"), work.glob("gzout_*"):
            for f in old:  # a run writes NEW files: freeing the previous run's gigabytes of tmpfs pages is not its job
                f.unlink()
        t0 = time.perf_counter()
        try:
            cli.main(["-A", "TAKARAV3", "--trim-polyA"] + inputs + outputs)
        except SystemExit as exc:  # pragma: no cover
            if exc.code:
                raise
        dt = time.perf_counter() - t0
        out[tag] = {"seconds": round(dt, 3), "M_pairs_per_s": round(n / dt / 1e6, 3)}


    plain_in = [str(work / "plain_R1.fastq"), str(work / "plain_R2.fastq")]
    plain_out = ["-o", str(work / "o1.fastq"), str(work / "o2.fastq"), "-s", str(work / "s1.fastq"), str(work / "s2.fastq")]
    run("plain_cold", plain_in, plain_out)
    run("plain_warm", plain_in, plain_out)
    gz_in = [str(work / "syn_R1.fastq.gz"), str(work / "syn_R2.fastq.gz")]
    run("gz_to_gz", gz_in, ["-O", str(work / "gzout")])
    run("gz_to_gz_warm", gz_in, ["-O", str(work / "gzout")])
    run("plain_to_gz", plain_in, ["-O", str(work / "gzout")])
    # the usual sequencer output: ONE gzip member per file (gzip -1 of the first half of the records), which no
    # reader can enter in the middle
    n_single = n // 2
    t0 = time.perf_counter()
    procs = []
    for m in (1, 2):
        head = subprocess.Popen(["head", "-n", str(4 * n_single), str(work / f"plain_R{m}.fastq")], stdout=subprocess.PIPE)
        procs.append((head, subprocess.Popen(["gzip", "-1"], stdin=head.stdout, stdout=open(work / f"single_R{m}.fastq.gz", "wb"))))
    for head, gz in procs:
        gz.wait()
        head.wait()
    out["make_single_member_gz_s"] = round(time.perf_counter() - t0, 2)
    n_all, n = n, n_single
    single_in = [str(work / "single_R1.fastq.gz"), str(work / "single_R2.fastq.gz")]
    run("single_member_gz_to_gz", single_in, ["-O", str(work / "gzout")])
    out["single_member_gz_to_gz"]["pairs"] = n_single
    n = n_all
    shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
