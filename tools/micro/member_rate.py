#!/usr/bin/env python3
"""One gzip member on one thread: libdeflate against the host library's byte-mode decoder (csh_inflate_stream).
    python3 tools/micro/member_rate.py [pairs]"""
import ctypes as C
import subprocess
import sys
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from cutseq_amd import build  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
work = Path("/dev/shm/cutseq_member_rate")
work.mkdir(exist_ok=True)
subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
raw = (work / "syn_R1.fastq.gz").read_bytes()
buf = np.frombuffer(raw + b"\0" * 64, dtype=np.uint8).copy()
out = np.empty(256 << 20, dtype=np.uint8)
out[:] = 1
H = C.CDLL(str(build.build_host()))
H.csh_inflate_stream.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
L = C.CDLL("libdeflate.so.0")
L.libdeflate_alloc_decompressor.restype = C.c_void_p
L.libdeflate_gzip_decompress_ex.restype = C.c_int
L.libdeflate_gzip_decompress_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
L.libdeflate_crc32.restype = C.c_uint32
L.libdeflate_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
d = L.libdeflate_alloc_decompressor()
best = [0.0, 0.0, 0.0]
for rep in range(10):
    nin, nout = C.c_size_t(), C.c_size_t()
    t = time.perf_counter()
    rc = L.libdeflate_gzip_decompress_ex(d, buf.ctypes.data, len(raw), out.ctypes.data, out.size, C.byref(nin), C.byref(nout))
    best[0] = max(best[0], nout.value / (time.perf_counter() - t) / 1e6)
    assert rc == 0
    end, no = C.c_int64(), C.c_int64()
    t = time.perf_counter()
    rc = H.csh_inflate_stream(buf.ctypes.data, len(raw), 80, out.ctypes.data, out.size, C.byref(end), C.byref(no))
    t1 = time.perf_counter()
    crc = L.libdeflate_crc32(0, out.ctypes.data, no.value)
    t2 = time.perf_counter()
    assert rc == 0 and no.value == nout.value
    best[1] = max(best[1], no.value / (t1 - t) / 1e6)
    best[2] = max(best[2], no.value / (t2 - t) / 1e6)
print(f"member of {nout.value >> 20} MB of text: libdeflate {best[0]:.0f} MB/s (CRC included), own byte-mode decoder {best[1]:.0f} MB/s, "
      f"with libdeflate's CRC-32 behind it {best[2]:.0f} MB/s")
import shutil
shutil.rmtree(work, ignore_errors=True)
