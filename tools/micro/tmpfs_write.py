#!/usr/bin/env python3
"""How fast can one process put N GB of finished text into a NEW file on tmpfs (what tier E's plain output does)?
Variants: pwrite from T threads; ftruncate + mmap + parallel copies (the writers' form); fallocate first; MAP_POPULATE.
    python3 tools/micro/tmpfs_write.py [GB] [threads]"""
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(gb * (1 << 30)) & ~((1 << 22) - 1)
src = np.frombuffer(os.urandom(1 << 20) * (n >> 20), dtype=np.uint8)
PIECE = 4 << 20
path = "/dev/shm/_tmpfs_write_probe"
pool = ThreadPoolExecutor(T)


def fresh():
    if os.path.exists(path):
        os.unlink(path)
    return os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)


def report(name, t0):
    dt = time.perf_counter() - t0
    print(f"{name:44s} {dt:6.3f} s  {n / dt / 1e9:6.2f} GB/s", flush=True)


def pwrite_all(fd):
    def job(off):
        os.pwrite(fd, memoryview(src)[off:off + PIECE], off)
    list(pool.map(job, range(0, n, PIECE)))


def copy_all(mm):
    dst = np.frombuffer(mm, dtype=np.uint8)

    def job(off):
        dst[off:off + PIECE] = src[off:off + PIECE]
    list(pool.map(job, range(0, n, PIECE)))
    del dst


for rep in range(2):
    fd = fresh(); t0 = time.perf_counter(); pwrite_all(fd); report(f"pwrite, {T} threads", t0); os.close(fd)
    fd = fresh(); t0 = time.perf_counter(); os.ftruncate(fd, n); mm = mmap.mmap(fd, n); copy_all(mm); report(f"ftruncate + mmap + copies, {T} threads", t0); mm.close(); os.close(fd)
    fd = fresh(); t0 = time.perf_counter(); os.posix_fallocate(fd, 0, n); report("  (posix_fallocate alone)", t0); pwrite_all(fd); report(f"fallocate + pwrite, {T} threads", t0); os.close(fd)
    fd = fresh(); t0 = time.perf_counter(); os.posix_fallocate(fd, 0, n); mm = mmap.mmap(fd, n); copy_all(mm); report(f"fallocate + mmap + copies, {T} threads", t0); mm.close(); os.close(fd)
    fd = fresh(); t0 = time.perf_counter(); os.ftruncate(fd, n); mm = mmap.mmap(fd, n, flags=mmap.MAP_SHARED | mmap.MAP_POPULATE); report("  (ftruncate + MAP_POPULATE alone)", t0); copy_all(mm); report(f"MAP_POPULATE + copies, {T} threads", t0); mm.close(); os.close(fd)
    # fallocate, then ONE thread pwrites the front part (no page faults, no zeroing, but writes to a file serialise) while
    # the pool copies the rest into the mapping
    import threading
    for frac in (0.3, 0.5):
        fd = fresh(); t0 = time.perf_counter(); os.posix_fallocate(fd, 0, n); mm = mmap.mmap(fd, n)
        split = int(n * frac) & ~(PIECE - 1)

        def front():
            for off in range(0, split, PIECE):
                os.pwrite(fd, memoryview(src)[off:off + PIECE], off)
        th = threading.Thread(target=front); th.start()
        dst = np.frombuffer(mm, dtype=np.uint8)

        def job(off):
            dst[off:off + PIECE] = src[off:off + PIECE]
        list(pool.map(job, range(split, n, PIECE)))
        th.join()
        del dst
        report(f"fallocate + {int(frac * 100)} % pwrite | mmap copies, {T} threads", t0); mm.close(); os.close(fd)
    # two files at once, as the run writes R1 and R2 together
    fds = [fresh()]
    path2 = path + "2"
    if os.path.exists(path2):
        os.unlink(path2)
    fds.append(os.open(path2, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600))
    t0 = time.perf_counter()

    def job2(arg):
        fd, off = arg
        os.pwrite(fd, memoryview(src)[off:off + PIECE], off)
    list(pool.map(job2, [(fd, off) for off in range(0, n, PIECE) for fd in fds]))
    dt = time.perf_counter() - t0
    print(f"{'two files, pwrite, ' + str(T) + ' threads':44s} {dt:6.3f} s  {2 * n / dt / 1e9:6.2f} GB/s", flush=True)
    for fd in fds:
        os.close(fd)
    os.unlink(path2)
os.unlink(path)
