#!/usr/bin/env python3
"""The plain-output writers' loop in isolation: TWO files, each fed 8 pieces of 256 MB in order by its own thread, copies in
a shared pool of 16 threads (textio.StreamWriter._copy_mapped).  Variants: pages faulted in by the copies (round 4),
mapped up front per copy job (MADV_POPULATE_WRITE), the next piece's pages allocated while the pool copies (ahead),
one mapping per file for the whole run instead of one per piece (keep)."""
import ctypes as C
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cutseq_amd import textio  # noqa: E402

work = Path(sys.argv[1] if len(sys.argv) > 1 else "/dev/shm")
PIECES, N, COPY = 8, 256 << 20, 4 << 20
src = np.ones(N, dtype=np.uint8)
libc = C.CDLL(None, use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
pool = ThreadPoolExecutor(16)


def writer(path, populate, ahead, keep):
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666)
    pos = alloc_end = 0
    t_alloc = t_copy = t_map = 0.0
    whole = None
    if keep:
        os.ftruncate(fd, PIECES * N)
        whole = mmap.mmap(fd, PIECES * N, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        wbase = np.frombuffer(whole, dtype=np.uint8).ctypes.data
    for k in range(PIECES):
        t0 = time.perf_counter()
        if pos + N > alloc_end:
            assert textio._fallocate(fd, alloc_end, pos + N - alloc_end)
            alloc_end = pos + N
        t1 = time.perf_counter()
        if keep:
            d0 = wbase + pos
        else:
            mm = mmap.mmap(fd, N, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE, offset=pos)
            dst = np.frombuffer(mm, dtype=np.uint8)
            d0 = dst.ctypes.data
        s0 = src.ctypes.data

        def job(lo):
            if populate:
                libc.madvise(d0 + lo, COPY, 23)
            C.memmove(d0 + lo, s0 + lo, COPY)
        futs = [pool.submit(job, lo) for lo in range(0, N, COPY)]
        t2 = time.perf_counter()
        if ahead and k + 1 < PIECES:
            assert textio._fallocate(fd, alloc_end, N)
            alloc_end += N
        t3 = time.perf_counter()
        for f in futs:
            f.result()
        if not keep:
            del dst
            mm.close()
        t4 = time.perf_counter()
        t_alloc += (t1 - t0) + (t3 - t2)
        t_map += t2 - t1
        t_copy += t4 - t3
        pos += N
    if whole is not None:
        whole.close()
    os.close(fd)
    return t_alloc, t_map, t_copy


def run(tag, **kw):
    paths = [work / f"w3_{i}.bin" for i in range(2)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(2) as outer:
        parts = list(outer.map(lambda p: writer(p, **kw), paths))
    dt = time.perf_counter() - t0
    for p in paths:
        os.unlink(p)
    print(f"{tag:34s} {2 * PIECES * N / dt / 1e9:6.2f} GB/s   alloc {parts[0][0]:.3f}  map {parts[0][1]:.3f}  wait-copies {parts[0][2]:.3f}")


for rep in range(2):
    run("faults (round 4)", populate=False, ahead=False, keep=False)
    run("populate", populate=True, ahead=False, keep=False)
    run("populate + ahead", populate=True, ahead=True, keep=False)
    run("faults + ahead", populate=False, ahead=True, keep=False)
    run("populate + keep mapping", populate=True, ahead=False, keep=True)
    run("populate + ahead + keep mapping", populate=True, ahead=True, keep=True)
