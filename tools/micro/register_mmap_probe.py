#!/usr/bin/env python3
"""Is zero-copy host I/O worth it?  Page-lock the mapping of a tmpfs file with hipHostRegister
and copy straight from / into it, against the product's path (parallel copies between the file's pages and a
page-locked block that stays registered).  Prints GB/s per step.

    python tools/micro/register_mmap_probe.py [GiB]
"""
import ctypes as C
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
H2D, D2H = 1, 2


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    n = int(gib * (1 << 30)) & ~((1 << 21) - 1)
    path = "/dev/shm/cutseq_register_probe.bin"
    with open(path, "wb") as fh:
        fh.write(os.urandom(1 << 20) * (n >> 20))
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), n) == 0
    pinned = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(pinned), n, 0) == 0
    pool = ThreadPoolExecutor(16)
    piece = 8 << 20

    def rate(t):
        return round(n / t / 1e9, 2)

    out = {}
    fd = os.open(path, os.O_RDWR)
    # the product's way in: parallel pread into a page-locked block, then one DMA copy
    mv = (C.c_char * n).from_address(pinned.value)
    t0 = time.perf_counter()
    list(pool.map(lambda lo: os.preadv(fd, [memoryview(mv)[lo:lo + piece]], lo), range(0, n, piece)))
    t1 = time.perf_counter()
    assert hip.hipMemcpy(dev, pinned, n, H2D) == 0
    t2 = time.perf_counter()
    out["pread_16_threads_GBps"], out["h2d_from_pinned_GBps"] = rate(t1 - t0), rate(t2 - t1)
    out["in_total_GBps (pread then copy, not overlapped)"] = rate(t2 - t0)
    # zero copy in: register the file's mapping, DMA straight from its pages
    mm = mmap.mmap(fd, n, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    addr = C.addressof(C.c_char.from_buffer(mm))
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(addr, n, 0)
    t1 = time.perf_counter()
    out["hipHostRegister_rc"] = rc
    if rc == 0:
        assert hip.hipMemcpy(dev, addr, n, H2D) == 0
        t2 = time.perf_counter()
        assert hip.hipMemcpy(dev, addr, n, H2D) == 0
        t3 = time.perf_counter()
        out["register_mapping_GBps"], out["h2d_from_mapping_GBps"] = rate(t1 - t0), rate(t3 - t2)
        out["in_total_zero_copy_GBps (register + copy)"] = rate((t1 - t0) + (t3 - t2))
        # way out: D2H into the registered mapping against D2H into pinned + parallel memcpy into the mapping
        t0 = time.perf_counter()
        assert hip.hipMemcpy(addr, dev, n, D2H) == 0
        t1 = time.perf_counter()
        out["d2h_into_mapping_GBps"] = rate(t1 - t0)
        t0 = time.perf_counter()
        hip.hipHostUnregister(addr)
        out["unregister_GBps"] = rate(time.perf_counter() - t0)
    t0 = time.perf_counter()
    assert hip.hipMemcpy(pinned, dev, n, D2H) == 0
    t1 = time.perf_counter()
    list(pool.map(lambda lo: C.memmove(addr + lo, pinned.value + lo, min(piece, n - lo)), range(0, n, piece)))
    t2 = time.perf_counter()
    out["d2h_into_pinned_GBps"], out["memcpy_into_mapping_16_threads_GBps"] = rate(t1 - t0), rate(t2 - t1)
    print(out)
    os.unlink(path)


if __name__ == "__main__":
    main()


def scaling():
    """hipHostRegister of 64 MB pieces of a tmpfs mapping from 1 / 2 / 4 / 8 threads at once, and of freshly
    fallocate-d pages (the output side)."""
    n = 2 << 30
    path = "/dev/shm/cutseq_register_probe2.bin"
    with open(path, "wb") as fh:
        fh.write(b"x" * (1 << 20) * (n >> 20))
    fd = os.open(path, os.O_RDWR)
    mm = mmap.mmap(fd, n, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    addr = C.addressof(C.c_char.from_buffer(mm))
    piece = 64 << 20
    res = {}
    for threads in (1, 2, 4, 8):
        pool = ThreadPoolExecutor(threads)
        t0 = time.perf_counter()
        rcs = list(pool.map(lambda lo: hip.hipHostRegister(addr + lo, piece, 0), range(0, n, piece)))
        t1 = time.perf_counter()
        list(pool.map(lambda lo: hip.hipHostUnregister(addr + lo), range(0, n, piece)))
        t2 = time.perf_counter()
        res[f"register_{threads}_threads_GBps"] = round(n / (t1 - t0) / 1e9, 1)
        res[f"unregister_{threads}_threads_GBps"] = round(n / (t2 - t1) / 1e9, 1)
        assert not any(rcs), rcs
        pool.shutdown()
    os.close(fd)
    os.unlink(path)
    # fresh pages: fallocate a new file, map, register (what an output range would cost)
    path = "/dev/shm/cutseq_register_probe3.bin"
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    t0 = time.perf_counter()
    os.posix_fallocate(fd, 0, n)
    t1 = time.perf_counter()
    mm2 = mmap.mmap(fd, n, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    addr2 = C.addressof(C.c_char.from_buffer(mm2))
    rc = hip.hipHostRegister(addr2, n, 0)
    t2 = time.perf_counter()
    res["fallocate_GBps"], res["register_fresh_pages_GBps"], res["rc_fresh"] = round(n / (t1 - t0) / 1e9, 1), round(n / (t2 - t1) / 1e9, 1), rc
    os.close(fd)
    os.unlink(path)
    print(res)


if __name__ == "__main__" and os.environ.get("PROBE_SCALING", "1") == "1":
    scaling()
