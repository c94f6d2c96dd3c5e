"""FASTQ chunk I/O for the CLI: gzip streams in, record-aligned SoA chunks to the engine,
formatted records out (tier E of SURVEY.md 8d; counterpart of ``cutadapt.files`` / dnaio /
xopen as the reference uses them, cutseq/run.py:434-441, 751-758).

Parsing and formatting are native (``csrc/cutseq_host.c``); (de)compression is zlib, which
releases the GIL, so reader, GPU submission, formatter and the per-file writer threads overlap.
"""
from __future__ import annotations

import ctypes as C
import gzip
import io
import queue
import threading
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from .synth import host_lib as _host_lib

CHUNK_READS = 1 << 18
_READ_BLOCK = 8 << 20


class FastqFormatError(ValueError):
    pass


class _FormatParams(C.Structure):
    _fields_ = [
        ("paired", C.c_int32), ("has_umi", C.c_int32), ("untrimmed_filter", C.c_int32),
        ("reverse_complement", C.c_int32), ("flag_too_short", C.c_uint8), ("flag_untrimmed", C.c_uint8),
        ("pad", C.c_uint8 * 2), ("suffix1", C.c_char_p * 2), ("suffix2", C.c_char_p * 2),
    ]


_bound = False


def _lib():
    global _bound
    L = _host_lib()
    if not _bound:
        i64, vp = C.c_int64, C.c_void_p
        L.csh_fastq_count.restype = i64
        L.csh_fastq_count.argtypes = [vp, i64, i64, C.c_int, C.POINTER(i64), C.POINTER(C.c_int32)]
        L.csh_fastq_parse.restype = i64
        L.csh_fastq_parse.argtypes = [vp, i64, i64, C.c_uint32, vp, vp, vp, vp, vp]
        L.csh_format_chunk.restype = i64
        L.csh_format_chunk.argtypes = [C.POINTER(_FormatParams), i64, C.c_uint32] + [vp] * 13 + [vp, vp, vp]
        _bound = True
    return L


def open_input(path: str):
    """Binary stream of the decompressed file (gzip by magic number, else plain)."""
    fh = open(path, "rb")
    magic = fh.read(2)
    fh.seek(0)
    if magic == b"\x1f\x8b":
        return gzip.GzipFile(fileobj=fh, mode="rb")
    return fh


@dataclass
class Chunk:
    n: int
    stride: int
    raw1: bytes
    name_off1: np.ndarray
    name_len1: np.ndarray
    seq1: np.ndarray
    qual1: np.ndarray
    len1: np.ndarray
    raw2: Optional[bytes] = None
    name_off2: Optional[np.ndarray] = None
    name_len2: Optional[np.ndarray] = None
    seq2: Optional[np.ndarray] = None
    qual2: Optional[np.ndarray] = None
    len2: Optional[np.ndarray] = None

    @property
    def paired(self) -> bool:
        return self.raw2 is not None


class _Stream:
    """Incremental reader: keeps the unparsed tail of one input file."""

    def __init__(self, path: str):
        self.path = path
        self.fh = open_input(path)
        self.buf = b""
        self.eof = False
        self.records_out = 0
        # decompression runs ahead in its own thread (zlib releases the GIL)
        self._q: "queue.Queue" = queue.Queue(maxsize=4)
        self._t = threading.Thread(target=self._pump, daemon=True)
        self._t.start()

    def _pump(self):
        try:
            while True:
                block = self.fh.read(_READ_BLOCK)
                self._q.put(block)
                if not block:
                    break
        except BaseException as exc:  # surfaced by the consumer
            self._q.put(exc)

    def fill(self, want_records: int):
        """Read until at least ``want_records`` complete records are buffered (or EOF).
        -> (records available (<= want), bytes they span, longest sequence)."""
        L = _lib()
        consumed, longest = C.c_int64(), C.c_int32()
        while True:
            n = L.csh_fastq_count(self.buf, len(self.buf), want_records, 1 if self.eof else 0,
                                  C.byref(consumed), C.byref(longest))
            if n >= want_records or self.eof:
                return int(n), int(consumed.value), int(longest.value)
            block = self._q.get()
            if isinstance(block, BaseException):
                raise block
            if not block:
                self.eof = True
            else:
                self.buf += block

    def take(self, n_records: int, n_bytes: int, stride: int):
        L = _lib()
        raw = self.buf[:n_bytes]
        seq = np.empty((n_records, stride), dtype=np.uint8)
        qual = np.empty((n_records, stride), dtype=np.uint8)
        lens = np.empty(n_records, dtype=np.uint16)
        noff = np.empty(n_records, dtype=np.int64)
        nlen = np.empty(n_records, dtype=np.int32)
        rc = L.csh_fastq_parse(raw, len(raw), n_records, stride, seq.ctypes.data, qual.ctypes.data, lens.ctypes.data,
                               noff.ctypes.data, nlen.ctypes.data)
        if rc < 0:
            raise FastqFormatError(
                f"{self.path}: malformed FASTQ record {self.records_out - rc} "
                "(expected '@' header, sequence, '+' line and a quality line of equal length)")
        self.buf = self.buf[n_bytes:]
        self.records_out += n_records
        return raw, noff, nlen, seq, qual, lens

    def close(self):
        self.fh.close()


def read_chunks(path1: str, path2: Optional[str] = None, chunk_reads: int = CHUNK_READS):
    """Yield record-aligned :class:`Chunk` objects (equal record counts for both mates)."""
    s1 = _Stream(path1)
    s2 = _Stream(path2) if path2 else None
    try:
        while True:
            n1, b1, l1 = s1.fill(chunk_reads)
            if s2 is None:
                if n1 == 0:
                    if s1.buf.strip():
                        raise FastqFormatError(f"{path1}: truncated FASTQ record at end of file")
                    return
                stride = max(4, (l1 + 3) // 4 * 4)
                yield Chunk(n1, stride, *s1.take(n1, b1, stride))
                continue
            n2, b2, l2 = s2.fill(chunk_reads)
            n = min(n1, n2)
            if n == 0:
                if n1 != n2 or s1.buf.strip() or s2.buf.strip():
                    raise FastqFormatError(
                        "Reads are improperly paired! There are more reads in one file than in the other, "
                        "or a record is truncated.")
                return
            if n < n1:
                n1, b1, l1 = s1.fill(n)
            if n < n2:
                n2, b2, l2 = s2.fill(n)
            stride = max(4, (max(l1, l2) + 3) // 4 * 4)
            a = s1.take(n, b1, stride)
            b = s2.take(n, b2, stride)
            yield Chunk(n, stride, *a, *b)
    finally:
        s1.close()
        if s2:
            s2.close()


def format_chunk(chunk: Chunk, plan, res1: np.ndarray, cap2: Optional[np.ndarray], res2: Optional[np.ndarray]):
    """-> (bytes[route][mate], counts[route]) with routes 0 trimmed, 1 short, 2 untrimmed."""
    L = _lib()
    fp = _FormatParams()
    fp.paired = 1 if chunk.paired else 0
    fp.has_umi = 1 if plan.has_umi else 0
    fp.untrimmed_filter = 1 if plan.untrimmed_filter else 0
    fp.reverse_complement = 1 if plan.reverse_complement else 0
    fp.flag_too_short, fp.flag_untrimmed = abi.CS_F_TOO_SHORT, abi.CS_F_UNTRIMMED
    suf1 = [s.encode() for s in plan.r1.name_suffixes] + [None, None]
    fp.suffix1[0], fp.suffix1[1] = suf1[0], suf1[1]
    if chunk.paired:
        suf2 = [s.encode() for s in plan.r2.name_suffixes] + [None, None]
        fp.suffix2[0], fp.suffix2[1] = suf2[0], suf2[1]
    # worst case per record: header (<= raw bytes in total) + '_' + two captures (<= 510) + sequence
    # + quality + "@\n\n+\n\n"
    per_rec = 2 * chunk.stride + 520
    cap_bytes = [len(chunk.raw1) + per_rec * chunk.n + 16,
                 (len(chunk.raw2) + per_rec * chunk.n + 16) if chunk.paired else 16]
    bufs = [[np.empty(cap_bytes[m], dtype=np.uint8) for m in range(2)] for _ in range(3)]
    out_ptrs = ((C.c_void_p * 2) * 3)()
    for r in range(3):
        for m in range(2):
            out_ptrs[r][m] = bufs[r][m].ctypes.data
    out_len = ((C.c_int64 * 2) * 3)()
    counts = (C.c_int64 * 3)()
    rc = L.csh_format_chunk(
        C.byref(fp), chunk.n, chunk.stride, chunk.raw1, chunk.name_off1.ctypes.data, chunk.name_len1.ctypes.data,
        chunk.seq1.ctypes.data, chunk.qual1.ctypes.data, res1.ctypes.data,
        cap2.ctypes.data if cap2 is not None else None,
        chunk.raw2 if chunk.paired else None,
        chunk.name_off2.ctypes.data if chunk.paired else None, chunk.name_len2.ctypes.data if chunk.paired else None,
        chunk.seq2.ctypes.data if chunk.paired else None, chunk.qual2.ctypes.data if chunk.paired else None,
        res2.ctypes.data if res2 is not None else None, out_ptrs, out_len, counts)
    if rc < 0:
        i = int(-rc - 1)
        n1 = chunk.raw1[chunk.name_off1[i]: chunk.name_off1[i] + chunk.name_len1[i]].decode(errors="replace")
        n2 = chunk.raw2[chunk.name_off2[i]: chunk.name_off2[i] + chunk.name_len2[i]].decode(errors="replace")
        raise ValueError(f"Input read IDs not identical: '{n1.split()[0] if n1.split() else n1}' != "
                         f"'{n2.split()[0] if n2.split() else n2}'")
    data = [[bufs[r][m][: out_len[r][m]].tobytes() for m in range(2)] for r in range(3)]
    return data, [int(c) for c in counts]


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        from .synth import usable_cpus
        _POOL = ThreadPoolExecutor(max_workers=max(2, usable_cpus(32)))
    return _POOL


def _gzip_member(data: bytes, level: int) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(data) + c.flush()


class OutputFile:
    """One output file.  ``.gz`` names are written as a sequence of gzip members (level 1 =
    cutadapt's default), one member per block, compressed in a shared thread pool and written
    in order by the file's own writer thread; anything else is written plain.  The decompressed
    byte stream is what parity is defined on."""

    def __init__(self, path: str, level: int = 1):
        self.path = path
        self.level = level
        self.gz = path.endswith(".gz")
        self.fh = open(path, "wb")
        self.q: "queue.Queue" = queue.Queue(maxsize=16)
        self.err: Optional[BaseException] = None
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    break
                self.fh.write(item.result() if self.gz else item)
        except BaseException as exc:  # pragma: no cover
            self.err = exc
        finally:
            self.fh.close()

    def write(self, data: bytes):
        if not data:
            return
        if self.gz:
            self.q.put(_pool().submit(_gzip_member, data, self.level))
        else:
            self.q.put(data)

    def close(self):
        self.q.put(None)
        self.t.join()
        if self.gz and self.fh.closed and Path(self.path).stat().st_size == 0:
            with open(self.path, "wb") as fh:  # an empty stream is still a valid gzip file
                fh.write(_gzip_member(b"", self.level))
        if self.err:
            raise self.err
