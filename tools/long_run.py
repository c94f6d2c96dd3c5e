#!/usr/bin/env python3
"""A longer run through the CLI than the tier-E probes make (default 24 M pairs, multi-member gz in, gz out): the rate,
the peak resident set and the arena's footprint at the end -- leaks and slow-downs show here, not in 8 M pairs.
    python3 tools/long_run.py [pairs]"""
import json
import resource
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
work = Path("/dev/shm/cutseq_long")
shutil.rmtree(work, ignore_errors=True)
work.mkdir(parents=True)
free = shutil.disk_usage(work).free
need = n * 2 * 90 * 2 + (4 << 30)
if free < need:
    raise SystemExit(f"{work}: {free >> 20} MiB free, {need >> 20} MiB wanted")
t0 = time.perf_counter()
subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
print(f"input written in {time.perf_counter() - t0:.1f} s", flush=True)
from cutseq_amd import fastq, run as cli  # noqa: E402

out = {"pairs": n}
for rep in range(2):
    for f in work.glob("out*"):
        f.unlink()
    t0 = time.perf_counter()
    try:
        cli.main(["-A", "TAKARAV3", "--trim-polyA", str(work / "syn_R1.fastq.gz"), str(work / "syn_R2.fastq.gz"), "-O", str(work / "out")])
    except SystemExit as exc:
        if exc.code:
            raise
    dt = time.perf_counter() - t0
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    out[f"run{rep}"] = {"seconds": round(dt, 2), "M_pairs_per_s": round(n / dt / 1e6, 2), "peak_rss_GB": round(rss, 2),
                        "arena_free_GB": round(sum(k * len(v) for k, v in fastq.ARENA._free.items()) / 1e9, 2),
                        "pinned_free_GB": round(sum(k * len(v) for k, v in fastq.PINNED._free.items()) / 1e9, 2)}
    print(json.dumps(out[f"run{rep}"]), flush=True)
sizes = {f.name: f.stat().st_size for f in work.glob("out*")}
out["output_GB"] = round(sum(sizes.values()) / 1e9, 2)
print(json.dumps(out))
shutil.rmtree(work, ignore_errors=True)
