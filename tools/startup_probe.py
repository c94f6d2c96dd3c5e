#!/usr/bin/env python3
"""Where does a short CLI run spend its start-up?  2 M pairs as a fresh process (what tools/tiers.py's tier E times),
with the text path's per-thread profile and the interpreter's import times."""
import gzip
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
D = Path("/dev/shm/cutseq_startup")
shutil.rmtree(D, ignore_errors=True)
D.mkdir(parents=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(D / "syn")], check=True, stdout=subprocess.DEVNULL)
for m in (1, 2):
    with gzip.open(D / f"syn_R{m}.fastq.gz", "rb") as s, open(D / f"plain_R{m}.fastq", "wb") as d:
        shutil.copyfileobj(s, d, 1 << 24)


def run(tag, args, extra_env=None, importtime=False):
    for f in list(D.glob("o?.fastq")) + list(D.glob("s?.fastq")) + list(D.glob("gz_*")):
        f.unlink()
    env = dict(os.environ, CUTSEQ_PROFILE="1", **(extra_env or {}))
    cmd = [sys.executable] + (["-X", "importtime"] if importtime else []) + ["-m", "cutseq_amd.run"] + args
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print(f"{tag}: rc {p.returncode} wall {dt:.3f} s = {n / dt / 1e6:.2f} M pairs/s")
    for line in p.stderr.splitlines():
        if "cutseq_profile" in line or "cutseq_phase" in line:
            print("   ", line[:1400])
    if importtime:
        rows = [l for l in p.stderr.splitlines() if l.startswith("import time:") and l.split("|")[1].strip().isdigit()]
        rows.sort(key=lambda l: -int(l.split("|")[1]))
        for l in rows[:8]:
            print("   ", l)


plain = [str(D / "plain_R1.fastq"), str(D / "plain_R2.fastq"), "-A", "TAKARAV3", "--trim-polyA", "-o", str(D / "o1.fastq"),
         str(D / "o2.fastq"), "-s", str(D / "s1.fastq"), str(D / "s2.fastq")]
gz = [str(D / "syn_R1.fastq.gz"), str(D / "syn_R2.fastq.gz"), "-A", "TAKARAV3", "--trim-polyA", "-O", str(D / "gz")]
run("plain->plain", plain, importtime=True)
run("plain->plain", plain)
run("gz->gz", gz)
run("gz->gz", gz)
t0 = time.perf_counter()
subprocess.run([sys.executable, "-c", "import numpy, cutseq_amd.run"], cwd=str(ROOT))
print(f"python + imports alone: {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter()
subprocess.run([sys.executable, "-c", "from cutseq_amd import capi; print(capi.device_count())"], cwd=str(ROOT), stdout=subprocess.DEVNULL)
print(f"... + HIP device count: {time.perf_counter() - t0:.3f} s")
shutil.rmtree(D, ignore_errors=True)
