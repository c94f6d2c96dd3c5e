#!/usr/bin/env python3
"""Does the parallel decoder of ONE gzip member (csrc/pinflate.c, codec.GzipSource._parallel_member) care how the member
was written?  The same 2 M records as one deflate stream (gzip -1) and as pigz-style pieces joined by sync flushes
(pieces of 262 144 / 65 536 / 4 096 records: pigz itself flushes every 128 KB of text).  GB/s of text, 16 threads."""
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np  # noqa: E402
from cutseq_amd import codec, fastq, workloads  # noqa: E402
from tools import tiers  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
work = Path("/dev/shm/cutseq_pieces")
work.mkdir(exist_ok=True)
batch = workloads.make_batch("config3", n)
text = tiers.fastq_text(batch.seq1, batch.qual1, batch.len1, 1)
flat = memoryview(text.reshape(-1))
rec = text.shape[1]
pool = ThreadPoolExecutor(16)


def member(piece_records):
    step = piece_records * rec
    spans = [(lo, min(len(flat), lo + step)) for lo in range(0, len(flat), step)]

    def job(args):
        lo, hi = args
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        return c.compress(flat[lo:hi]) + c.flush(zlib.Z_FINISH if hi == len(flat) else zlib.Z_SYNC_FLUSH)
    parts = list(pool.map(job, spans))
    crc = zlib.crc32(flat)
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\xff" + b"".join(parts) + int(crc).to_bytes(4, "little") + int(len(flat) & 0xffffffff).to_bytes(4, "little")


os.environ["CUTSEQ_DEBUG_INFLATE"] = "1"
for label, pr in (("one stream", n), ("pieces of 1048576", 1 << 20), ("pieces of 262144", 262144), ("pieces of 65536", 65536), ("pieces of 4096", 4096), ("pieces of 400 (pigz: 128 KB)", 400)):
    path = work / "m.fastq.gz"
    path.write_bytes(member(pr))
    for rep in range(2):
        src = codec.GzipSource(str(path), fastq._pool(), fastq.ARENA.take, fastq.ARENA.give)
        t0 = time.perf_counter()
        got = 0
        for arr, nbytes in src.blocks():
            got += nbytes
            fastq.ARENA.give(arr)
        dt = time.perf_counter() - t0
        stats = dict(src.stats)
        src.close()
        assert got == len(flat)
    print(f"{label:32s} {os.path.getsize(path) / 1e6:8.1f} MB  {got / dt / 1e9:6.2f} GB/s of text   chunks {stats.get('chunks')} serial {stats.get('serial', 0)}  "
          + " ".join(f"{k} {v:.2f}" for k, v in sorted(stats.items()) if k.endswith("_s")), flush=True)
    path.unlink()
