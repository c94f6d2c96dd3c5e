#!/usr/bin/env python3
"""Where do the seconds of a plain -> plain run of the command line go?  (SURVEY.md 8d tier E)

  python tools/host_io_profile.py [pairs] [workdir]      (on the GPU box; files on tmpfs like the driver's tier legs)

Writes the bench reads as plain FASTQ, then, in this process (warm interpreter / HIP / buffers from the second run on):
  1. raw ceilings of the box's CPU share: memcpy with 1 / 4 / 8 / 16 threads, pread of the input files into one
     buffer with 1..16 threads, fallocate + parallel copies into a fresh tmpfs file (what the writers do);
  2. the readers alone (TextReader x 2, blocks dropped), the command line with its outputs on /dev/null-like sinks
     (CUTSEQ_DISCARD_OUTPUT=1: the writers drop their bytes), the full command line three times with CUTSEQ_PROFILE=1
     (seconds per thread and activity).
One JSON object on stdout; profiles/r05_host_io.md quotes it.
"""
import ctypes as C
import json
import os
import shutil
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["CUTSEQ_PROFILE"] = "1"
os.environ.setdefault("CUTSEQ_CHUNK_READS", "262144")

import numpy as np  # noqa: E402


def threads_available():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def memcpy_rates(nbytes=1 << 30):
    src = np.ones(nbytes, dtype=np.uint8)
    dst = np.empty(nbytes, dtype=np.uint8)
    dst[:] = 0
    out = {}
    for t in (1, 4, 8, 16):
        piece = nbytes // t
        with ThreadPoolExecutor(t) as pool:
            t0 = time.perf_counter()
            list(pool.map(lambda i: C.memmove(dst.ctypes.data + i * piece, src.ctypes.data + i * piece, piece), range(t)))
            out[str(t)] = round(nbytes / (time.perf_counter() - t0) / 1e9, 2)
    return out


def pread_rates(paths, block=8 << 20):
    total = sum(os.path.getsize(p) for p in paths)
    buf = np.empty(64 * block, dtype=np.uint8)
    buf[:] = 0
    mv = memoryview(buf)
    out = {}
    for t in (1, 2, 4, 8, 16):
        jobs = [(p, off) for p in paths for off in range(0, os.path.getsize(p), block)]
        fds = {p: os.open(p, os.O_RDONLY) for p in paths}

        def job(args):
            i, (p, off) = args
            slot = (i % 64) * block
            return os.preadv(fds[p], [mv[slot:slot + block]], off)
        with ThreadPoolExecutor(t) as pool:
            t0 = time.perf_counter()
            got = sum(pool.map(job, enumerate(jobs)))
            dt = time.perf_counter() - t0
        for fd in fds.values():
            os.close(fd)
        assert got == total
        out[str(t)] = round(total / dt / 1e9, 2)
    return out


def write_rates(work: Path, nbytes=2 << 30, piece=4 << 20):
    """fallocate + parallel copies into a shared mapping (the writers' way), and plain pwrite from 1 / 4 threads."""
    import mmap
    from cutseq_amd import textio
    src = np.ones(nbytes, dtype=np.uint8)
    out = {}
    for t in (4, 8, 16):
        path = work / "w.bin"
        fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666)
        t0 = time.perf_counter()
        ok = textio._fallocate(fd, 0, nbytes)
        t_alloc = time.perf_counter() - t0
        mm = mmap.mmap(fd, nbytes, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        dst = np.frombuffer(mm, dtype=np.uint8)
        with ThreadPoolExecutor(t) as pool:
            t1 = time.perf_counter()
            list(pool.map(lambda lo: C.memmove(dst.ctypes.data + lo, src.ctypes.data + lo, min(piece, nbytes - lo)), range(0, nbytes, piece)))
            t_copy = time.perf_counter() - t1
        del dst
        mm.close()
        os.close(fd)
        os.unlink(path)
        out[f"map_{t}"] = {"fallocate_ok": ok, "fallocate_GBps": round(nbytes / t_alloc / 1e9, 2), "copy_GBps": round(nbytes / t_copy / 1e9, 2),
                           "both_GBps": round(nbytes / (t_alloc + t_copy) / 1e9, 2)}
    for t in (1, 4):
        paths = [work / f"w{i}.bin" for i in range(t)]
        fds = [os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666) for p in paths]
        mv = memoryview(src)
        share = nbytes // t

        def job(i):
            at = 0
            while at < share:
                at += os.pwrite(fds[i], mv[i * share + at:i * share + min(share, at + (64 << 20))], at)
        with ThreadPoolExecutor(t) as pool:
            t0 = time.perf_counter()
            list(pool.map(job, range(t)))
            dt = time.perf_counter() - t0
        for fd, p in zip(fds, paths):
            os.close(fd)
            os.unlink(p)
        out[f"pwrite_{t}_files"] = round(nbytes / dt / 1e9, 2)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    work = Path(sys.argv[2] if len(sys.argv) > 2 else "/dev/shm/cutseq_hostio")
    work.mkdir(parents=True, exist_ok=True)
    out = {"pairs": n, "threads": threads_available(), "workdir": str(work)}
    from cutseq_amd import workloads
    from tools import tiers
    batch = workloads.make_batch("config3", n)
    with ThreadPoolExecutor(threads_available()) as pool:
        out["input_bytes"] = tiers.write_inputs(work, batch, n, pool)
    form = os.environ.get("HOSTIO_INPUT", "plain")  # plain | multi_gz | single_gz: what the readers are given
    keep = {"plain": "plain_R%d.fastq", "multi_gz": "multi_R%d.fastq.gz", "single_gz": "single_R%d.fastq.gz"}[form]
    for name in ("plain_R1.fastq", "plain_R2.fastq", "multi_R1.fastq.gz", "multi_R2.fastq.gz", "single_R1.fastq.gz",
                 "single_R2.fastq.gz"):
        if name not in (keep % 1, keep % 2):
            (work / name).unlink()
    del batch
    ins = [str(work / (keep % 1)), str(work / (keep % 2))]
    out["input_form"] = form
    if form == "plain":
        out["memcpy_GBps_by_threads"] = memcpy_rates()
        out["pread_GBps_by_threads"] = pread_rates(ins)
        out["tmpfs_write"] = write_rates(work)

    from cutseq_amd import fastq, run as cli, textio
    # the readers alone
    t0 = time.perf_counter()
    readers = [textio.TextReader(p, 262144) for p in ins]
    got = 0
    while True:
        bs = [r.get() for r in readers]
        if bs[0] is None:
            break
        got += bs[0].n
        for b in bs:
            b.release()
    dt = time.perf_counter() - t0
    out["readers_alone"] = {"seconds": round(dt, 3), "M_pairs_per_s": round(got / dt / 1e6, 2),
                            "GBps_of_file_bytes": round(sum(os.path.getsize(p) for p in ins) / dt / 1e9, 2)}
    outs = ["-o", str(work / "o1.fastq"), str(work / "o2.fastq"), "-s", str(work / "s1.fastq"), str(work / "s2.fastq")]

    gz_outs = ["-o", str(work / "o1.fastq.gz"), str(work / "o2.fastq.gz"), "-s", str(work / "s1.fastq.gz"), str(work / "s2.fastq.gz")]

    def run(tag, env=None, outs=outs):
        for f in list(work.glob("o[12].fastq*")) + list(work.glob("s[12].fastq*")):
            f.unlink()
        for k, v in (env or {}).items():
            os.environ[k] = v
        import io
        import contextlib
        err = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stderr(err):
            try:
                cli.main(["-A", "TAKARAV3", "--trim-polyA"] + ins + outs)
            except SystemExit as exc:
                if exc.code:
                    raise
        dt = time.perf_counter() - t0
        for k in (env or {}):
            del os.environ[k]
        prof = [json.loads(l) for l in err.getvalue().splitlines() if l.startswith('{"cutseq_profile"')]
        out[tag] = {"seconds": round(dt, 3), "M_pairs_per_s": round(n / dt / 1e6, 2), "profile": prof[-1]["cutseq_profile"] if prof else None}

    if form == "plain":
        run("full_cold")
        for rep in range(3):
            run(f"full_warm_{rep + 1}")
        run("discard_output", {"CUTSEQ_DISCARD_OUTPUT": "1"})
    for rep in range(3):
        run(f"{form}_to_gz_{rep + 1}", outs=gz_outs)
    if form != "plain":
        run("discard_output", {"CUTSEQ_DISCARD_OUTPUT": "1"}, outs=gz_outs)
    shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
