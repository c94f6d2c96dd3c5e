#!/usr/bin/env python3
"""Tier E of BASELINE config 5: 96-plex demultiplexing through the CLI (plain text in, one pair of files per barcode out).
    python3 tools/demux_e2e.py [pairs]"""
import json
import shutil
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from cutseq_amd import abi, fastq, plan as planmod, run as cli, workloads  # noqa: E402
from cutseq_amd.common import BarcodeConfig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
work = Path("/dev/shm/cutseq_demux")
shutil.rmtree(work, ignore_errors=True)
work.mkdir(parents=True)
codes = workloads.config5_barcodes()
scheme = workloads.config5_scheme(codes)
tp = planmod.compile_paired(BarcodeConfig(scheme), planmod.CutadaptConfig())
tp.has_umi = False
tp.r1.name_suffixes = ()
tp.r2.name_suffixes = ()
outs = [open(work / "in_R1.fastq", "wb"), open(work / "in_R2.fastq", "wb")]
step = 1 << 18
for lo in range(0, n, step):
    m = min(step, n - lo)
    b = workloads.make_batch("config5", m, lo)
    names1 = "".join(f"SIM:{lo + i} 1:N:0:IDX\n" for i in range(m)).encode()
    names2 = names1.replace(b" 1:N", b" 2:N")
    lens = np.array([len(x) for x in names1.split(b"\n")[:-1]], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens + 1)[:-1]]).astype(np.int64)
    res = np.zeros(m, dtype=abi.RESULT_DTYPE)
    res["stop"] = 150
    chunk = fastq.Chunk(m, b.stride, names1, offs, lens, b.seq1, b.qual1, b.len1, names2, offs, lens, b.seq2, b.qual2, b.len2)
    data, _ = fastq.format_chunk(chunk, tp, res, None, res.copy())
    outs[0].write(data[0][0])
    outs[1].write(data[0][1])
for o in outs:
    o.close()
(work / "bc.tsv").write_text("".join(f"bc{i:02d}\t{c}\n" for i, c in enumerate(codes)))
res = {}
for rep in range(2):
    for f in work.glob("out*"):
        f.unlink()
    t0 = time.perf_counter()
    try:
        cli.main(["-a", scheme, "--demux-barcodes", str(work / "bc.tsv"), "-O", str(work / "out"), str(work / "in_R1.fastq"), str(work / "in_R2.fastq")])
    except SystemExit as exc:
        if exc.code:
            raise
    dt = time.perf_counter() - t0
    res[f"run{rep}"] = {"seconds": round(dt, 3), "M_pairs_per_s": round(n / dt / 1e6, 3)}
print(json.dumps({"pairs": n, "plex": len(codes), **res}))
shutil.rmtree(work, ignore_errors=True)
