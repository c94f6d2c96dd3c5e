"""FASTQ chunk I/O for the CLI: gzip streams in, record-aligned SoA chunks to the engine,
formatted records out (tier E of SURVEY.md 8d; counterpart of ``cutadapt.files`` / dnaio /
xopen as the reference uses them, cutseq/run.py:434-441, 751-758).

Parsing and formatting are native (``csrc/cutseq_host.c``), (de)compression is ``codec.py``
(libdeflate / zlib); all of it releases the GIL, so reader, parser pool, GPU submission, formatter
and the per-file writer threads overlap.  The arrays that cross PCIe (sequence, quality, lengths,
results) come from a pinned arena when the caller asks for it (``read_chunks(pinned=True)``).
"""
from __future__ import annotations

import ctypes as C
import mmap
import os
import queue
import threading
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import abi, codec
from .synth import host_lib as _host_lib

CHUNK_READS = 1 << 16
_READ_BLOCK = 8 << 20
_MADV_HUGEPAGE = getattr(mmap, "MADV_HUGEPAGE", 14)


class FastqFormatError(ValueError):
    pass


class _FormatParams(C.Structure):
    _fields_ = [
        ("paired", C.c_int32), ("has_umi", C.c_int32), ("untrimmed_filter", C.c_int32),
        ("reverse_complement", C.c_int32), ("flag_too_short", C.c_uint8), ("flag_untrimmed", C.c_uint8),
        ("pad", C.c_uint8 * 2), ("suffix1", C.c_char_p * 2), ("suffix2", C.c_char_p * 2),
    ]


_bound = None  # the host library once this module's prototypes are bound (see _lib)


_bind_lock = threading.Lock()


def _lib():
    """The host helper library with this module's prototypes bound.  The reader threads of both mates come here
    at the same moment: the binding happens once, under a lock, on the object that is then published -- a thread
    must never get a library object whose functions still have default prototypes (a 64-bit buffer address
    would go through as a C int)."""
    global _bound
    if _bound is not None:
        return _bound
    with _bind_lock:
        if _bound is not None:
            return _bound
        L = _host_lib()
        i64, vp = C.c_int64, C.c_void_p
        L.csh_fastq_count.restype = i64
        L.csh_fastq_count.argtypes = [vp, i64, i64, C.c_int, C.POINTER(i64), C.POINTER(C.c_int32)]
        L.csh_fastq_parse.restype = i64
        L.csh_fastq_parse.argtypes = [vp, i64, i64, C.c_uint32, vp, vp, vp, vp, vp]
        L.csh_format_chunk.restype = i64
        L.csh_format_chunk.argtypes = [C.POINTER(_FormatParams), i64, C.c_uint32] + [vp] * 13 + [vp, vp, vp]
        L.csh_format_chunk_bins.restype = i64
        L.csh_format_chunk_bins.argtypes = ([C.POINTER(_FormatParams), i64, C.c_uint32] + [vp] * 13 +
                                            [vp, C.c_int32, vp, vp, vp] + [vp, vp, vp])
        _bound = L
    return L


class _Arena:
    """Recycles the big per-chunk buffers.  Fresh memory costs a page fault per 4 KB on first touch --
    more than the parsing and formatting themselves -- so buffers go back here when a chunk is done
    (:meth:`Chunk.release`) and are handed out again, already mapped."""

    def __init__(self, keep_per_size: int = 48, pinned: bool = False):
        self._lock = threading.Lock()
        self._free: dict = {}
        self._keep = keep_per_size
        self._pinned = pinned
        # pinned arena: cs_alloc_pinned_huge (page-locked memory on huge pages: 17 instead of 7.5 GB/s to get, same
        # copy bandwidth) unless CUTSEQ_PINNED_THP=0
        self.huge = os.environ.get("CUTSEQ_PINNED_THP", "1") != "0"
        self._all_pinned: list = []  # (address, array) of every pinned buffer ever handed out

    @staticmethod
    def _bucket(nbytes: int) -> int:
        step = 1 << 16 if nbytes <= (1 << 20) else (1 << 20 if nbytes <= (16 << 20) else 4 << 20)
        return max(step, (int(nbytes) + step - 1) // step * step)

    def take(self, nbytes: int) -> np.ndarray:
        size = self._bucket(nbytes)
        with self._lock:
            stack = self._free.get(size)
            if stack:
                return stack.pop()
        if not self._pinned:
            # transparent huge pages where the kernel grants them: one fault per 2 MB instead of one per 4 KB
            # (first-touch faults otherwise cost more than the parsing that fills the buffer)
            m = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
            if size >= (2 << 20):
                try:
                    m.madvise(_MADV_HUGEPAGE)
                except (OSError, ValueError, AttributeError):
                    pass
            return np.frombuffer(m, dtype=np.uint8)
        # page-locked host memory (hipHostMalloc through the C ABI): H2D / D2H copies of these buffers are
        # real DMA transfers that overlap the kernels of the other slot
        from . import capi
        L = capi.load()
        ptr = (L.cs_alloc_pinned_huge if self.huge else L.cs_alloc_pinned)(size)
        if not ptr:
            raise MemoryError(f"cs_alloc_pinned({size}) failed: {L.cs_last_error().decode(errors='replace')}")
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(size,))
        with self._lock:
            self._all_pinned.append((ptr, arr))
        return arr

    _KEEP_MAX = 96 << 20  # larger buffers (one huge gzip member) are unmapped instead of kept for the whole run

    def give(self, arr: np.ndarray) -> None:
        if arr.size > self._KEEP_MAX and not self._pinned:
            return
        with self._lock:
            stack = self._free.setdefault(arr.size, [])
            if len(stack) < self._keep:
                stack.append(arr)


    def free_pinned(self) -> None:
        """Release every pinned buffer (end of a run; nothing may still be using them)."""
        if not self._pinned:
            return
        from . import capi
        L = capi.load()
        with self._lock:
            self._free.clear()
            for ptr, _ in self._all_pinned:
                L.cs_free_pinned(ptr)
            self._all_pinned.clear()


ARENA = _Arena()
PINNED = _Arena(keep_per_size=64, pinned=True)


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

    _owned: tuple = ()  # (arena, buffer) pairs behind the arrays above
    bc: Optional[np.ndarray] = None  # demultiplexing runs: barcode index per record (filled by the device)

    @property
    def paired(self) -> bool:
        return self.raw2 is not None

    def release(self) -> None:
        """Hand the chunk's buffers back for reuse; the chunk must not be touched afterwards."""
        for arena, arr in self._owned:
            arena.give(arr)
        self._owned = ()


class _Failure:
    """An exception travelling through a queue to the thread that consumes it."""

    def __init__(self, exc: BaseException):
        self.exc = exc


@dataclass
class _Half:
    """One mate's share of a chunk, parsed."""

    n: int
    stride: int
    raw: bytes
    name_off: np.ndarray
    name_len: np.ndarray
    seq: np.ndarray
    qual: np.ndarray
    lens: np.ndarray
    owned: tuple = ()


class _StrideHint:
    """Row stride shared by the two parser threads: it only grows, so after the first chunk both mates
    are almost always parsed with the same stride straight away."""

    def __init__(self):
        self._lock = threading.Lock()
        self._value = 4

    def raise_to(self, longest: int) -> int:
        want = max(4, (longest + 3) // 4 * 4)
        with self._lock:
            if want > self._value:
                self._value = want
            return self._value


def _raw_pointer(raw):
    """Address of a bytes / memoryview / ndarray buffer for the native calls."""
    return raw if isinstance(raw, bytes) else np.frombuffer(raw, dtype=np.uint8).ctypes.data


def _parse(path: str, raw, n_records: int, stride: int, first_record: int, raw_owner=(), pinned: bool = False) -> _Half:
    """Native parse of ``n_records`` complete records in ``raw`` into fixed-stride rows."""
    L = _lib()
    dev = PINNED if pinned else ARENA  # what crosses PCIe: sequence, quality, lengths
    owned = [(dev, dev.take(n_records * stride)), (dev, dev.take(n_records * stride)), (dev, dev.take(n_records * 2)),
             (ARENA, ARENA.take(n_records * 8)), (ARENA, ARENA.take(n_records * 4))]
    bufs = [b for _, b in owned]
    seq = bufs[0][: n_records * stride].reshape(n_records, stride)
    qual = bufs[1][: n_records * stride].reshape(n_records, stride)
    lens = bufs[2][: n_records * 2].view(np.uint16)
    noff = bufs[3][: n_records * 8].view(np.int64)
    nlen = bufs[4][: n_records * 4].view(np.int32)
    rc = L.csh_fastq_parse(_raw_pointer(raw), len(raw), n_records, stride, seq.ctypes.data, qual.ctypes.data,
                           lens.ctypes.data, noff.ctypes.data, nlen.ctypes.data)
    if rc < 0:
        raise FastqFormatError(
            f"{path}: malformed FASTQ record {first_record - rc} "
            "(expected '@' header, sequence, '+' line and a quality line of equal length)")
    return _Half(n_records, stride, raw, noff, nlen, seq, qual, lens, tuple(owned) + tuple(raw_owner))


class _Reader:
    """One input file -> parsed halves of ``chunk_reads`` records.  One thread produces decompressed text
    (gzip: ``codec.GzipSource``, BGZF blocks inflate in the pool; plain files are read straight into the
    chunk buffer), one cuts it at record boundaries; the parse of each chunk runs in the shared pool, so
    several chunks of a file are parsed at once.  The consumer only pairs the halves up."""

    def __init__(self, path: str, chunk_reads: int, hint: _StrideHint, pinned: bool = False):
        self.path, self.chunk_reads, self.hint, self.pinned = path, chunk_reads, hint, pinned
        self.gz = codec.is_gzip(path)
        self.src = (codec.GzipSource(path, _pool(), ARENA.take, ARENA.give) if self.gz
                    else open(path, "rb", buffering=0))
        self._blocks: "queue.Queue" = queue.Queue(maxsize=4)
        self.halves: "queue.Queue" = queue.Queue(maxsize=3)
        self._stop = False
        self._threads = [threading.Thread(target=self._cut, daemon=True)]
        if self.gz:
            self._threads.append(threading.Thread(target=self._inflate, daemon=True))
        for t in self._threads:
            t.start()

    def _put(self, q, item) -> bool:
        while not self._stop:
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    def _inflate(self):
        gen = self.src.blocks()
        try:
            for block in gen:
                if self._stop or not self._put(self._blocks, block):
                    return
            self._put(self._blocks, None)
        except BaseException as exc:
            self._put(self._blocks, _Failure(exc))
        finally:
            gen.close()  # the generator's own clean-up waits for every pool task that still reads the mapped file

    def _fill(self, buf: np.ndarray, fill: int):
        """More text behind buf[:fill] -> (buf, fill, eof); the buffer grows when it has to."""
        if self.gz:
            block = self._blocks.get()
            if isinstance(block, _Failure):
                raise block.exc
            if block is None:
                return buf, fill, True
            text, nbytes = block
            if fill + nbytes > buf.size:
                bigger = ARENA.take(max(2 * buf.size, fill + nbytes))
                C.memmove(bigger.ctypes.data, buf.ctypes.data, fill)
                ARENA.give(buf)
                buf = bigger
            C.memmove(buf.ctypes.data + fill, text.ctypes.data, nbytes)
            if isinstance(text.base, mmap.mmap):  # an arena buffer (zlib fallback blocks are views of bytes objects)
                ARENA.give(text)
            return buf, fill + nbytes, False
        if buf.size - fill < _READ_BLOCK:
            bigger = ARENA.take(max(2 * buf.size, fill + _READ_BLOCK))
            C.memmove(bigger.ctypes.data, buf.ctypes.data, fill)
            ARENA.give(buf)
            buf = bigger
        got = self.src.readinto(memoryview(buf)[fill:fill + _READ_BLOCK])  # page cache -> chunk buffer, one copy
        return buf, fill + (got or 0), not got

    def _cut(self):
        L = _lib()
        # text is appended to an arena buffer; the buffer becomes the chunk's raw text, only the short tail
        # behind the last complete record moves on to the next one
        buf = ARENA.take(4 * _READ_BLOCK)
        fill = 0
        eof = False
        done = 0       # records handed out so far
        need = 0       # bytes worth buffering before the next count (from the previous chunk's density)
        consumed, longest = C.c_int64(), C.c_int32()
        pool = _pool()
        try:
            while not self._stop:
                while not eof and fill < max(need, 1):
                    buf, fill, eof = self._fill(buf, fill)
                n = L.csh_fastq_count(buf.ctypes.data, fill, self.chunk_reads, 1 if eof else 0, C.byref(consumed),
                                      C.byref(longest)) if fill else 0
                if n < self.chunk_reads and not eof:
                    need = fill + 1  # not there yet: wait for at least one more block
                    continue
                if n == 0:
                    if bytes(memoryview(buf)[:fill]).strip():
                        raise FastqFormatError(f"{self.path}: truncated FASTQ record at end of file")
                    ARENA.give(buf)
                    self._put(self.halves, None)
                    return
                used = int(consumed.value)
                nxt = ARENA.take(max(buf.size, 4 * _READ_BLOCK))
                C.memmove(nxt.ctypes.data, buf.ctypes.data + used, fill - used)
                raw = memoryview(buf)[:used]  # compares equal to bytes, slices without copying
                stride = self.hint.raise_to(int(longest.value))
                job = pool.submit(_parse, self.path, raw, int(n), stride, done, ((ARENA, buf),), self.pinned)
                buf, fill = nxt, fill - used
                done += int(n)
                need = used + (used >> 6)  # the next chunk will be about as long
                if not self._put(self.halves, job):
                    return
        except BaseException as exc:
            self._put(self.halves, _Failure(exc))

    def next_half(self) -> Optional[_Half]:
        item = self.halves.get()
        if isinstance(item, _Failure):
            raise item.exc
        return item.result() if item is not None else None

    def close(self):
        """Stop the reader.  The source (an mmap for gzip input) is closed only after the reader threads -- and with
        them every inflate task of the pool that holds a raw address into the map -- are done: closing it under
        them would unmap memory a libdeflate call is still reading."""
        self._stop = True
        deadline = time.monotonic() + 30.0
        while any(t.is_alive() for t in self._threads) and time.monotonic() < deadline:
            for q in (self._blocks, self.halves):  # unblock producers
                try:
                    while True:
                        q.get_nowait()
                except queue.Empty:
                    pass
            for t in self._threads:
                t.join(timeout=0.05)
        if any(t.is_alive() for t in self._threads):
            return  # a thread is stuck in I/O: leak the mapping rather than pull it away under the thread
        self.src.close()


def _restride(path: str, half: _Half, stride: int, first_record: int, pinned: bool) -> _Half:
    if half.stride == stride:
        return half
    again = _parse(path, half.raw, half.n, stride, first_record, raw_owner=half.owned[5:], pinned=pinned)
    for arena, arr in half.owned[:5]:
        arena.give(arr)
    return again


def read_chunks(path1: str, path2: Optional[str] = None, chunk_reads: int = CHUNK_READS, pinned: bool = False):
    """Yield record-aligned :class:`Chunk` objects (equal record counts for both mates).  ``pinned``: the
    arrays the GPU copies (sequence, quality, lengths) live in page-locked memory (needs the HIP library)."""
    hint = _StrideHint()
    r1 = _Reader(path1, chunk_reads, hint, pinned)
    r2 = _Reader(path2, chunk_reads, hint, pinned) if path2 else None
    done = 0
    try:
        while True:
            h1 = r1.next_half()
            if r2 is None:
                if h1 is None:
                    return
                yield Chunk(h1.n, h1.stride, h1.raw, h1.name_off, h1.name_len, h1.seq, h1.qual, h1.lens,
                            _owned=h1.owned)
                done += h1.n
                continue
            h2 = r2.next_half()
            if h1 is None and h2 is None:
                return
            if h1 is None or h2 is None or h1.n != h2.n:
                # both files are cut every chunk_reads records: a difference means unequal record counts
                raise FastqFormatError(
                    "Reads are improperly paired! There are more reads in one file than in the other, "
                    "or a record is truncated.")
            stride = max(h1.stride, h2.stride)
            h1, h2 = _restride(path1, h1, stride, done, pinned), _restride(path2, h2, stride, done, pinned)
            yield Chunk(h1.n, stride, h1.raw, h1.name_off, h1.name_len, h1.seq, h1.qual, h1.lens,
                        h2.raw, h2.name_off, h2.name_len, h2.seq, h2.qual, h2.lens, _owned=h1.owned + h2.owned)
            done += h1.n
    finally:
        r1.close()
        if r2:
            r2.close()


_tls = threading.local()


def _out_buffers(cap_bytes: Sequence[int]):
    """Six output buffers (route x mate) of this thread, grown on demand and reused between chunks."""
    bufs = getattr(_tls, "bufs", None)
    if bufs is None:
        bufs = _tls.bufs = [[np.empty(0, dtype=np.uint8) for _ in range(2)] for _ in range(3)]
    for r in range(3):
        for m in range(2):
            if bufs[r][m].size < cap_bytes[m]:
                bufs[r][m] = np.empty(int(cap_bytes[m] * 1.25) + 4096, dtype=np.uint8)
    return bufs


class Lease:
    """The first ``n`` bytes of an arena buffer on their way to a file; whoever writes them gives the buffer back."""

    __slots__ = ("arr", "n")

    def __init__(self, arr: np.ndarray, n: int):
        self.arr, self.n = arr, n

    def view(self) -> memoryview:
        return memoryview(self.arr)[: self.n]

    def release(self) -> None:
        ARENA.give(self.arr)


def format_chunk(chunk: Chunk, plan, res1: np.ndarray, cap2: Optional[np.ndarray], res2: Optional[np.ndarray],
                 copy: bool = True, lease=None):
    """-> (data[route][mate], counts[route]) with routes 0 trimmed, 1 short, 2 untrimmed.  ``data`` holds
    ``bytes``; with ``copy=False`` it holds memoryviews into this thread's reusable buffers, valid
    until the thread formats its next chunk -- except the streams flagged in ``lease[route][mate]``,
    which are formatted straight into arena buffers and come back as :class:`Lease` objects."""
    L = _lib()
    fp = _format_params(chunk, plan)
    # worst case per record: what the input record held (header, sequence, quality: all inside raw)
    # + '_' + two captures (<= 510) + "@\n\n+\n\n"
    cap_bytes = [len(chunk.raw1) + 528 * chunk.n + 16, (len(chunk.raw2) + 528 * chunk.n + 16) if chunk.paired else 16]
    bufs = _out_buffers(cap_bytes)
    leased = [[None, None] for _ in range(3)]
    out_ptrs = ((C.c_void_p * 2) * 3)()
    for r in range(3):
        for m in range(2):
            if lease is not None and lease[r][m] and not copy:
                leased[r][m] = ARENA.take(cap_bytes[m])
                out_ptrs[r][m] = leased[r][m].ctypes.data
            else:
                out_ptrs[r][m] = bufs[r][m].ctypes.data
    out_len = ((C.c_int64 * 2) * 3)()
    counts = (C.c_int64 * 3)()
    rc = L.csh_format_chunk(
        C.byref(fp), chunk.n, chunk.stride, _raw_pointer(chunk.raw1), chunk.name_off1.ctypes.data,
        chunk.name_len1.ctypes.data, chunk.seq1.ctypes.data, chunk.qual1.ctypes.data, res1.ctypes.data,
        cap2.ctypes.data if cap2 is not None else None,
        _raw_pointer(chunk.raw2) if chunk.paired else None,
        chunk.name_off2.ctypes.data if chunk.paired else None, chunk.name_len2.ctypes.data if chunk.paired else None,
        chunk.seq2.ctypes.data if chunk.paired else None, chunk.qual2.ctypes.data if chunk.paired else None,
        res2.ctypes.data if res2 is not None else None, out_ptrs, out_len, counts)
    if rc < 0:
        for row in leased:
            for arr in row:
                if arr is not None:
                    ARENA.give(arr)
        i = int(-rc - 1)
        n1 = bytes(chunk.raw1[chunk.name_off1[i]: chunk.name_off1[i] + chunk.name_len1[i]]).decode(errors="replace")
        n2 = bytes(chunk.raw2[chunk.name_off2[i]: chunk.name_off2[i] + chunk.name_len2[i]]).decode(errors="replace")
        raise ValueError(f"Input read IDs not identical: '{n1.split()[0] if n1.split() else n1}' != "
                         f"'{n2.split()[0] if n2.split() else n2}'")
    views = [[(Lease(leased[r][m], int(out_len[r][m])) if leased[r][m] is not None
               else memoryview(bufs[r][m])[: out_len[r][m]]) for m in range(2)] for r in range(3)]
    if copy:
        return [[bytes(v) for v in row] for row in views], [int(c) for c in counts]
    return views, [int(c) for c in counts]


def _format_params(chunk: Chunk, plan) -> _FormatParams:
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
    return fp


def format_chunk_bins(chunk: Chunk, plan, res1, cap2, res2, bc: np.ndarray, n_bins: int):
    """Demultiplexed formatting: -> (binned[mate] uint8 array, bin_off[mate][n_bins + 1], bin_counts[n_bins],
    views[route][mate] (routes 1 and 2 in use), counts[route]).  The arrays in ``binned`` are arena buffers
    the caller gives back."""
    L = _lib()
    fp = _format_params(chunk, plan)
    cap_bytes = [len(chunk.raw1) + 528 * chunk.n + 16, (len(chunk.raw2) + 528 * chunk.n + 16) if chunk.paired else 16]
    bufs = _out_buffers(cap_bytes)
    out_ptrs = ((C.c_void_p * 2) * 3)()
    for r in range(3):
        for m in range(2):
            out_ptrs[r][m] = bufs[r][m].ctypes.data
    binned = [ARENA.take(cap_bytes[0]), ARENA.take(cap_bytes[1])]
    binned_ptrs = (C.c_void_p * 2)(binned[0].ctypes.data, binned[1].ctypes.data)
    bin_off = np.zeros((2, n_bins + 1), dtype=np.int64)
    bin_counts = np.zeros(n_bins, dtype=np.int64)
    out_len = ((C.c_int64 * 2) * 3)()
    counts = (C.c_int64 * 3)()
    rc = L.csh_format_chunk_bins(
        C.byref(fp), chunk.n, chunk.stride, _raw_pointer(chunk.raw1), chunk.name_off1.ctypes.data,
        chunk.name_len1.ctypes.data, chunk.seq1.ctypes.data, chunk.qual1.ctypes.data, res1.ctypes.data,
        cap2.ctypes.data if cap2 is not None else None,
        _raw_pointer(chunk.raw2) if chunk.paired else None,
        chunk.name_off2.ctypes.data if chunk.paired else None, chunk.name_len2.ctypes.data if chunk.paired else None,
        chunk.seq2.ctypes.data if chunk.paired else None, chunk.qual2.ctypes.data if chunk.paired else None,
        res2.ctypes.data if res2 is not None else None, bc.ctypes.data, n_bins, binned_ptrs, bin_off.ctypes.data,
        bin_counts.ctypes.data, out_ptrs, out_len, counts)
    if rc < 0:
        for arr in binned:
            ARENA.give(arr)
        if -rc > chunk.n:
            raise ValueError("demultiplexed formatting failed (bad arguments)")
        raise ValueError(f"Input read IDs not identical in record {int(-rc)} of the chunk")
    views = [[memoryview(bufs[r][m])[: out_len[r][m]] for m in range(2)] for r in range(3)]
    return binned, bin_off, bin_counts, views, [int(c) for c in counts]


def finish_chunk(chunk: Chunk, plan, res1, cap2, res2, gz: Sequence[Sequence[Optional[bool]]], level: int = 1,
                 n_bins: int = 0):
    """Worker-thread job of the CLI: format one chunk and turn each wanted stream into what goes to disk:
    one gzip member (bytes), or the plain text in an arena buffer (:class:`Lease`).  ``gz[route][mate]`` is
    True / False for compressed / plain outputs and None where no file is open.  With ``n_bins`` (a
    demultiplexing run, ``chunk.bc`` filled) the trimmed route is split by barcode: streams 3 .. 3 + n_bins - 1.
    -> (blobs[stream][mate], counts per stream)."""
    if n_bins:
        binned, bin_off, bin_counts, views, counts = format_chunk_bins(chunk, plan, res1, cap2, res2, chunk.bc, n_bins)
        blobs = [[None, None] for _ in range(3 + n_bins)]
        try:
            for r in (1, 2):
                for m in range(2):
                    if gz[r][m] is not None and len(views[r][m]):
                        blobs[r][m] = codec.gzip_member(views[r][m], level) if gz[r][m] else bytes(views[r][m])
            for b in range(n_bins):
                for m in range(2 if chunk.paired else 1):
                    lo, hi = int(bin_off[m][b]), int(bin_off[m][b + 1])
                    if hi > lo and gz[3 + b][m] is not None:
                        piece = memoryview(binned[m])[lo:hi]
                        blobs[3 + b][m] = codec.gzip_member(piece, level) if gz[3 + b][m] else bytes(piece)
        finally:
            for arr in binned:
                ARENA.give(arr)
        return blobs, [0, counts[1], counts[2]] + [int(c) for c in bin_counts]
    lease = [[gz[r][m] is False for m in range(2)] for r in range(3)]
    views, counts = format_chunk(chunk, plan, res1, cap2, res2, copy=False, lease=lease)
    blobs = [[None, None] for _ in range(3)]
    for r in range(3):
        for m in range(2):
            v = views[r][m]
            if isinstance(v, Lease):
                if v.n:
                    blobs[r][m] = v
                else:
                    v.release()
            elif gz[r][m] is not None and len(v):
                blobs[r][m] = codec.gzip_member(v, level)
    return blobs, counts


_POOL = None
_THREADS = None  # -t/--threads of the CLI: upper bound of the host pool (None = every usable core)


def set_threads(n) -> None:
    """Bound the host thread pool (parse / format / inflate / deflate jobs) -- what ``-t`` means here: the reference
    hands it to ``make_runner(inpaths, cores=N)`` (cutseq/run.py:436, 753, 998-1003), where it is the number of
    worker processes; the per-read work of this engine runs on the GPU, the host threads feed it.  Must be called
    before the first chunk is read (the pool is created once)."""
    global _THREADS
    if _POOL is not None and n != _THREADS:
        raise RuntimeError("the host pool is already running")
    _THREADS = None if n is None else max(1, int(n))
    if _THREADS is not None:  # the library's own host threads (demultiplexing table build) honour the same bound
        os.environ["CUTSEQ_HOST_THREADS"] = str(_THREADS)


def pool_size() -> int:
    from .synth import usable_cpus
    # (the cap: one process with a worker per GPU of an eight-GPU node inflates for all of them -- 32 threads were 20 GB/s
    # of text, a quarter of what eight text engines take; the GPU boxes' 16-thread share is below either cap)
    cores = max(2, usable_cpus(128))
    return cores if _THREADS is None else min(cores, _THREADS)


def _pool():
    global _POOL
    if _POOL is None:
        with _bind_lock:  # the reader threads of both mates ask at the same moment: one pool, not two
            if _POOL is None:
                from concurrent.futures import ThreadPoolExecutor
                _POOL = ThreadPoolExecutor(max_workers=pool_size())
    return _POOL


class OutputFile:
    """One output file.  ``.gz`` names are written as a sequence of gzip members (level 1 =
    cutadapt's default), one member per block, compressed in a shared thread pool and written
    in order by the file's own writer thread; anything else is written plain.  The decompressed
    byte stream is what parity is defined on."""

    def __init__(self, path: str, level: int = 1):
        self.path = path
        self.level = level
        self.gz = path.lower().endswith(".gz")
        self.fh = open(path, "wb")
        self.q: "queue.Queue" = queue.Queue(maxsize=16)
        self.err: Optional[BaseException] = None
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                break
            if self.err is not None:
                # after a failure: keep draining so that producers never block on a dead consumer
                if isinstance(item, tuple):
                    try:
                        blob = item[0].result()[0][item[1]][item[2]]
                        if isinstance(blob, Lease):
                            blob.release()
                    except BaseException:
                        pass
                continue
            try:
                if isinstance(item, tuple):  # (future of finish_chunk, route, mate)
                    blob = item[0].result()[0][item[1]][item[2]]
                    if isinstance(blob, Lease):
                        try:
                            self.fh.write(blob.view())
                        finally:
                            blob.release()
                    elif blob:
                        self.fh.write(blob)
                else:
                    self.fh.write(item.result() if self.gz else item)
            except BaseException as exc:
                self.err = exc
        try:
            self.fh.close()
        except BaseException as exc:  # pragma: no cover
            self.err = self.err or exc

    def _check(self):
        if self.err is not None:
            raise self.err

    def write(self, data: bytes):
        self._check()
        if not data:
            return
        if self.gz:
            self.q.put(_pool().submit(codec.gzip_member, data, self.level))
        else:
            self.q.put(data)

    def write_job(self, future, route: int, mate: int):
        """Queue the (route, mate) stream of a :func:`finish_chunk` job; written when the job is done."""
        self._check()
        self.q.put((future, route, mate))

    def close(self):
        self.q.put(None)
        self.t.join()
        if self.err is None and self.gz and Path(self.path).stat().st_size == 0:
            with open(self.path, "wb") as fh:  # an empty stream is still a valid gzip file
                fh.write(codec.gzip_member(b"", self.level))
        self._check()
