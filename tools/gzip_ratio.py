#!/usr/bin/env python3
"""Size of the device's gzip output against zlib level 1 (`gzip -1`, what xopen writes for the reference) and
libdeflate level 1 (the host path, CUTSEQ_GPU_DEFLATE=0) -- VERDICT r3 item 7.  Runs the CLI on the bench reads
(2 M synthetic pairs, TAKARAV3 + --trim-polyA) and on the reference's own 10 000 pairs, with the device's LZ77 stage on
and off (CUTSEQ_GPU_LZ=0: round 3's literal-only blocks); every output is inflated with zlib and compared.

    python tools/gzip_ratio.py [pairs]  ->  one JSON object (profiles/r04_gzip_ratio.json)
"""
import gzip
import json
import os
import shutil
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import tiers  # noqa: E402
from cutseq_amd import run as cli, workloads  # noqa: E402


def run_case(inputs, work, tag, env):
    for k, v in env.items():
        os.environ[k] = v
    outs = [str(work / f"{tag}_{k}.fastq.gz") for k in ("o1", "o2", "s1", "s2")]
    t0 = time.perf_counter()
    try:
        cli.main(["-A", "TAKARAV3", "--trim-polyA"] + inputs + ["-o", outs[0], outs[1], "-s", outs[2], outs[3]])
    except SystemExit as exc:
        if exc.code:
            raise
    dt = time.perf_counter() - t0
    for k in env:
        os.environ.pop(k, None)
    return outs, dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    work = Path("/dev/shm/cutseq_gzip_ratio")
    work.mkdir(parents=True, exist_ok=True)
    out = {}
    try:
        batch = workloads.make_batch("config3", n)
        with ThreadPoolExecutor(tiers.host_threads()) as pool:
            tiers.write_inputs(work, batch, n, pool)
        sets = {"bench reads": ([str(work / "plain_R1.fastq"), str(work / "plain_R2.fastq")], n),
                "reference's 10 000 pairs (tests/golden/fixture10k)": (
                    [str(ROOT / "tests" / "golden" / "fixture10k_R1.fq.gz"), str(ROOT / "tests" / "golden" / "fixture10k_R2.fq.gz")], 10_000)}
        for label, (inputs, pairs) in sets.items():
            row = {"pairs": pairs}
            texts = None
            for tag, env in (("device_lz", {}), ("device_literal_only", {"CUTSEQ_GPU_LZ": "0"}), ("host_libdeflate_1", {"CUTSEQ_GPU_DEFLATE": "0"})):
                outs, dt = run_case(inputs, work, tag, env)
                got = [gzip.decompress(open(p, "rb").read()) for p in outs]
                if texts is None:
                    texts = got
                row[tag] = {"bytes": sum(os.path.getsize(p) for p in outs), "seconds": round(dt, 3), "same_text": got == texts}
                for p in outs:
                    os.unlink(p)
            raw = sum(len(t) for t in texts)
            row["text_bytes"] = raw
            row["zlib_1 (gzip -1)"] = {"bytes": sum(len(zlib.compress(t, 1)) + 12 for t in texts)}
            row["zlib_6"] = {"bytes": sum(len(zlib.compress(t, 6)) + 12 for t in texts)}
            for k in ("device_lz", "device_literal_only", "host_libdeflate_1", "zlib_1 (gzip -1)", "zlib_6"):
                row[k]["ratio"] = round(raw / row[k]["bytes"], 3)
            row["device_lz_vs_gzip_1"] = round(row["device_lz"]["bytes"] / row["zlib_1 (gzip -1)"]["bytes"], 3)
            out[label] = row
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
