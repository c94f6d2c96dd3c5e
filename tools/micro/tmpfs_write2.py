#!/usr/bin/env python3
"""Ways to put N bytes from a pinned-like buffer into a NEW tmpfs file (the plain-output writers' job, textio.py):
fallocate + parallel copies into a shared mapping (round 3..4), the same with the pages mapped up front by
madvise(MADV_POPULATE_WRITE) per piece in the pool, with MAP_POPULATE, and plain pwrite.  GB/s, 16 threads."""
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

MADV_POPULATE_WRITE = 23
work = Path(sys.argv[1] if len(sys.argv) > 1 else "/dev/shm")
nbytes, piece = 2 << 30, 4 << 20
src = np.ones(nbytes, dtype=np.uint8)
libc = C.CDLL(None, use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
pool = ThreadPoolExecutor(16)


def run(tag, populate=None, map_populate=False, files=1, fallocate=True):
    paths = [work / f"w{tag}{i}.bin" for i in range(files)]
    t0 = time.perf_counter()

    def one(path):
        fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666)
        ta = time.perf_counter()
        if fallocate:
            assert textio._fallocate(fd, 0, nbytes)
        else:
            os.ftruncate(fd, nbytes)
        tb = time.perf_counter()
        flags = mmap.MAP_SHARED | (mmap.MAP_POPULATE if map_populate else 0)
        mm = mmap.mmap(fd, nbytes, flags, mmap.PROT_READ | mmap.PROT_WRITE)
        tc = time.perf_counter()
        dst = np.frombuffer(mm, dtype=np.uint8)
        d0, s0 = dst.ctypes.data, src.ctypes.data

        def job(lo):
            n = min(piece, nbytes - lo)
            if populate:
                rc = libc.madvise(d0 + lo, n, MADV_POPULATE_WRITE)
                if rc != 0:
                    raise OSError(C.get_errno(), "madvise")
            C.memmove(d0 + lo, s0 + lo, n)
        list(pool.map(job, range(0, nbytes, piece)))
        td = time.perf_counter()
        del dst
        mm.close()
        os.close(fd)
        return tb - ta, tc - tb, td - tc
    if files == 1:
        parts = [one(paths[0])]
    else:
        with ThreadPoolExecutor(files) as outer:
            parts = list(outer.map(one, paths))
    dt = time.perf_counter() - t0
    for p in paths:
        os.unlink(p)
    print(f"{tag:28s} files {files}: {files * nbytes / dt / 1e9:6.2f} GB/s   alloc {parts[0][0]:.3f} s  mmap {parts[0][1]:.3f} s  copy {parts[0][2]:.3f} s")


for rep in range(2):
    run("fallocate+map", files=1)
    run("fallocate+map", files=2)
    run("fallocate+populate_write", populate=True, files=1)
    run("fallocate+populate_write", populate=True, files=2)
    run("truncate+populate_write", populate=True, files=1, fallocate=False)
    run("truncate+populate_write", populate=True, files=2, fallocate=False)
    run("fallocate+MAP_POPULATE", map_populate=True, files=1)
    run("truncate+map (faults alloc)", files=2, fallocate=False)
