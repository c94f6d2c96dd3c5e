#!/usr/bin/env python3
"""--ranks N against one process on the same gzip input: wall time of the command line as a child
process, multi-member gzip in (members in lockstep between the mates: split by member index, nothing inflated twice),
gzip out.  On a one-GPU box both ranks share GPU 0 and the box's 16 host threads (8 per rank).

    python tools/ranks_bench.py [pairs] [ranks]
"""
import json
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import tiers  # noqa: E402
from cutseq_amd import workloads  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    work = Path("/dev/shm/cutseq_ranks_bench")
    work.mkdir(parents=True, exist_ok=True)
    out = {"pairs": n, "ranks": world, "host_threads": tiers.host_threads()}
    try:
        batch = workloads.make_batch("config3", n)
        with ThreadPoolExecutor(tiers.host_threads()) as pool:
            tiers.write_inputs(work, batch, n, pool)
        del batch
        env = dict(os.environ, PYTHONPATH=str(ROOT))
        for tag, extra, envx in (("one_process", [], {}), (f"ranks{world}", ["--ranks", str(world)], {"CUTSEQ_DEVICES": ",".join(["0"] * world)}),
                                 ("one_process_again", [], {}), (f"ranks{world}_again", ["--ranks", str(world)], {"CUTSEQ_DEVICES": ",".join(["0"] * world)})):
            for form, ins in (("gz_multi", ["multi_R1.fastq.gz", "multi_R2.fastq.gz"]), ("plain", ["plain_R1.fastq", "plain_R2.fastq"])):
                outs = [str(work / f"o{k}.fastq.gz") for k in range(4)]
                cmd = [sys.executable, "-m", "cutseq_amd.run", "-A", "TAKARAV3", "--trim-polyA", str(work / ins[0]), str(work / ins[1]),
                       "-o", outs[0], outs[1], "-s", outs[2], outs[3], "--json-file", str(work / "r.json")] + extra
                t0 = time.perf_counter()
                r = subprocess.run(cmd, cwd=str(ROOT), env=dict(env, **envx), capture_output=True, text=True)
                dt = time.perf_counter() - t0
                rep = json.load(open(work / "r.json")) if r.returncode == 0 else {}
                out[f"{tag}:{form}"] = {"seconds": round(dt, 3), "M_pairs_per_s": round(n / dt / 1e6, 2), "rc": r.returncode,
                                        "split": rep.get("engine", {}).get("ranks_split"),
                                        "sizes": [os.path.getsize(p) for p in outs] if r.returncode == 0 else r.stderr[-300:]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
