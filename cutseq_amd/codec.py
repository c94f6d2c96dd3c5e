"""gzip codec of the CLI (tier E; counterpart of xopen as cutadapt drives it for the reference,
cutseq/run.py:434-441, 751-758).

Output: one gzip member per block, level 1 (cutadapt's default), compressed in the caller's
thread pool -- libdeflate when ``libdeflate.so.0`` can be loaded, zlib otherwise.
Input: a stream of decompressed blocks, written into recycled buffers (``take`` / ``give`` callbacks:
fresh memory costs a page fault per 4 KB, more than the inflate itself).
  * BGZF (block boundaries in the headers): blocks inflate in parallel in the pool;
  * any other gzip file: member by member with libdeflate (one call per member, 2-3x zlib's rate;
    multi-member files -- this tool's own output -- never hold more than one member in memory);
  * a member too large for that (> ``_MEMBER_CAP`` decompressed), or no libdeflate: zlib streaming.
The decompressed byte stream is what parity is defined on; the compressed bytes depend on the codec.
"""
from __future__ import annotations

import ctypes as C
import mmap
import os
import struct
import sys
import threading
import zlib
from typing import Iterator

# Largest decompressed member taken in one libdeflate call.  A member that does not fit the first buffer is tried
# ONCE more at this size and then streamed through zlib: a multi-gigabyte single-member file (the usual sequencer
# output) is not inflated from its start over and over, and no gigabyte buffers pile up in the arena.
_MEMBER_CAP = 256 << 20
_STREAM_BLOCK = 8 << 20

_lib = None
_lib_tried = False
_tls = threading.local()


_lib_lock = threading.Lock()


def libdeflate():
    """The shared library, or None (then everything goes through zlib).  Loaded and bound once, under a lock."""
    global _lib, _lib_tried
    if _lib_tried:
        return _lib
    with _lib_lock:
        if _lib_tried:
            return _lib
        for name in ("libdeflate.so.0", "libdeflate.so"):
            try:
                L = C.CDLL(name)
            except OSError:
                continue
            L.libdeflate_alloc_compressor.restype = C.c_void_p
            L.libdeflate_alloc_compressor.argtypes = [C.c_int]
            L.libdeflate_gzip_compress.restype = C.c_size_t
            L.libdeflate_gzip_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.libdeflate_gzip_compress_bound.restype = C.c_size_t
            L.libdeflate_gzip_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
            L.libdeflate_alloc_decompressor.restype = C.c_void_p
            L.libdeflate_gzip_decompress_ex.restype = C.c_int
            L.libdeflate_gzip_decompress_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
            L.libdeflate_crc32.restype = C.c_uint32
            L.libdeflate_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
            _lib = L
            break
        _lib_tried = True
    return _lib


def _view_address(mv: memoryview) -> int:
    import numpy as np
    return np.frombuffer(mv, dtype=np.uint8).ctypes.data


def gzip_member(data, level: int = 1) -> bytes:
    """``data`` (bytes-like) as one complete gzip member."""
    L = libdeflate()
    n = len(data)
    if L is None or n == 0:
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()
    comp = getattr(_tls, "comp", None)
    if comp is None or comp[0] != level:
        comp = _tls.comp = (level, L.libdeflate_alloc_compressor(level))
    bound = L.libdeflate_gzip_compress_bound(comp[1], n)
    out = getattr(_tls, "out", None)
    if out is None or len(out) < bound:
        out = _tls.out = bytearray(int(bound * 1.25) + 64)
    src = data if isinstance(data, bytes) else _view_address(memoryview(data))
    dst = (C.c_char * len(out)).from_buffer(out)
    got = L.libdeflate_gzip_compress(comp[1], src, n, dst, len(out))
    if got == 0:  # cannot happen with a bound-sized buffer; stay correct anyway
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()
    return bytes(memoryview(out)[:got])


def _decompressor():
    d = getattr(_tls, "decomp", None)
    if d is None:
        d = _tls.decomp = libdeflate().libdeflate_alloc_decompressor()
    return d


def _bgzf_block_size(buf, pos: int) -> int:
    """Total size of the BGZF block at ``pos`` (0 if the member there is not a BGZF block)."""
    if pos + 18 > len(buf) or buf[pos:pos + 4] != b"\x1f\x8b\x08\x04":
        return 0
    xlen = struct.unpack_from("<H", buf, pos + 10)[0]
    p, end = pos + 12, pos + 12 + xlen
    while p + 4 <= end and end <= len(buf):
        si1, si2, slen = buf[p], buf[p + 1], struct.unpack_from("<H", buf, p + 2)[0]
        if si1 == 66 and si2 == 67 and slen == 2:
            return struct.unpack_from("<H", buf, p + 4)[0] + 1
        p += 4 + slen
    return 0


def _np_take(nbytes: int):
    import numpy as np
    return np.empty(nbytes, dtype=np.uint8)


def _np_give(arr) -> None:
    pass


def _inflate_span(buf, lo: int, hi: int, out_hint: int, take=_np_take, give=_np_give):
    """Every member inside buf[lo:hi] (member-aligned), concatenated -> (uint8 array, bytes used)."""
    L, d = libdeflate(), _decompressor()
    base = _view_address(memoryview(buf))
    out = take(max(out_hint, 1 << 16))
    used = 0
    n_in, n_out = C.c_size_t(), C.c_size_t()
    pos = lo
    while pos < hi:
        while True:
            rc = L.libdeflate_gzip_decompress_ex(d, base + pos, hi - pos, out.ctypes.data + used, out.size - used,
                                                 C.byref(n_in), C.byref(n_out))
            if rc == 3:  # LIBDEFLATE_INSUFFICIENT_SPACE
                bigger = take(2 * out.size)
                C.memmove(bigger.ctypes.data, out.ctypes.data, used)
                give(out)
                out = bigger
                continue
            if rc != 0:
                give(out)
                raise OSError(f"corrupt gzip data (libdeflate error {rc})")
            break
        used += n_out.value
        pos += n_in.value
    return out, used



_PI = None
_PI_LOCK = threading.Lock()


def _pinflate():
    """libcutseq_host.so with the parallel-inflate entry points bound, or None (library not built)."""
    global _PI
    if _PI is None:
        with _PI_LOCK:
            if _PI is None:
                try:
                    from pathlib import Path
                    L = C.CDLL(str(Path(__file__).with_name("libcutseq_host.so")))
                    L.csh_deflate_find_block.restype = C.c_int64
                    L.csh_deflate_find_block.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
                    L.csh_inflate_chunk.restype = C.c_int
                    L.csh_inflate_chunk.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
                    L.csh_resolve_markers.restype = C.c_int64
                    L.csh_resolve_markers.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
                    L.csh_next_window.restype = C.c_int64
                    L.csh_next_window.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
                    L.csh_crc32_combine_many.restype = C.c_uint32
                    L.csh_crc32_combine_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
                    L.csh_find_gzip_magic.restype = C.c_int64
                    L.csh_find_gzip_magic.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
                    L.csh_inflate_stream.restype = C.c_int
                    L.csh_inflate_stream.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
                    _PI = L
                except (OSError, AttributeError):
                    _PI = False
    return _PI or None


def _gzip_header_size(buf, pos: int) -> int:
    """Bytes of the gzip member header at ``pos`` (RFC 1952), 0 if there is none."""
    n = len(buf)
    if pos + 18 > n or buf[pos:pos + 3] != b"\x1f\x8b\x08":
        return 0
    flg = buf[pos + 3]
    if flg & 0xE0:
        return 0
    p = pos + 10
    if flg & 4:  # FEXTRA
        if p + 2 > n:
            return 0
        p += 2 + struct.unpack_from("<H", buf, p)[0]
    for bit in (8, 16):  # FNAME, FCOMMENT: zero-terminated
        if flg & bit:
            z = buf.find(b"\0", p)
            if z < 0:
                return 0
            p = z + 1
    if flg & 2:  # FHCRC
        p += 2
    return p - pos if p < n else 0


class GzipSource:
    """Decompressed blocks of a gzip file, in order.  ``blocks()`` yields (buffer, nbytes) pairs: the first
    ``nbytes`` of the uint8 array are text; the consumer hands the array back with ``give`` when done."""

    def __init__(self, path: str, pool=None, take=_np_take, give=_np_give, post=None):
        """``post(address, nbytes) -> value``: run on every inflated block by the thread that inflated it (the bytes are
        still in its cache), e.g. a newline count; ``side(buffer)`` hands the value out once."""
        self.path = path
        self.pool = pool
        self.take, self.give = take, give
        self.post = post
        self._side = {}
        self.stats = {}  # diagnostics: chunks of a big member decoded in parallel / again serially
        self.fh = open(path, "rb")
        self.size = self.fh.seek(0, 2)
        self.fh.seek(0)
        self.map = mmap.mmap(self.fh.fileno(), 0, access=mmap.ACCESS_READ) if self.size else None
        self._open = True
        with _lib_lock:
            GzipSource._active += 1

    _active = 0  # sources open right now: they share the pool (how far each one speculates ahead)

    def _posted(self, out, produced):
        if self.post is not None and out is not None:
            self._side[out.ctypes.data] = self.post(out.ctypes.data, produced)

    def side(self, arr):
        return self._side.pop(arr.ctypes.data, None)

    def close(self):
        if self._open:
            self._open = False
            with _lib_lock:
                GzipSource._active -= 1
        if self.map is not None:
            try:
                self.map.close()
            except BufferError:  # a block handed out earlier still points into the map
                pass
        self.fh.close()

    def blocks(self, start: int = 0) -> Iterator[tuple]:
        """(buffer, nbytes) pairs from the member that begins at compressed offset ``start`` on."""
        for _, arr, nbytes in self.indexed_blocks(start):
            yield arr, nbytes

    def indexed_blocks(self, start: int = 0) -> Iterator[tuple]:
        """(compressed offset at which the block's first member starts, buffer, nbytes); the offset is -1 for a block
        in whose front the stream cannot be entered (the second and later blocks of one huge member; no libdeflate).  What the multi-process
        form of the CLI (``ranks.py``) builds its split points from."""
        if self.map is None:
            return
        if libdeflate() is None:
            if start:
                raise OSError(f"{self.path}: cannot start inside a gzip stream without libdeflate")
            for arr, nbytes in self._zlib_stream(0):
                yield -1, arr, nbytes
            return
        if _bgzf_block_size(self.map, start):
            yield from self._bgzf(start)
            return
        yield from self._members(start)

    # -- BGZF: boundaries are in the headers, spans of blocks inflate in parallel ------------------
    def _bgzf(self, start: int = 0) -> Iterator[tuple]:
        buf, n = self.map, self.size
        spans, lo, pos = [], start, start
        while pos < n:
            size = _bgzf_block_size(buf, pos)
            if size == 0:  # a foreign member in the middle: the rest goes member by member
                break
            pos += size
            if pos - lo >= (4 << 20):
                spans.append((lo, pos))
                lo = pos
        if pos > lo:
            spans.append((lo, pos))
        tail = pos
        def span(a, b):
            out, used = _inflate_span(buf, a, b, 4 * (b - a), self.take, self.give)
            self._posted(out, used)
            return out, used

        if self.pool is None:
            for a, b in spans:
                yield (a,) + span(a, b)
        else:
            from collections import deque
            pending = deque()
            for a, b in spans:
                pending.append((a, self.pool.submit(span, a, b)))
                if len(pending) >= 8:
                    a0, fut = pending.popleft()
                    yield (a0,) + fut.result()
            while pending:
                a0, fut = pending.popleft()
                yield (a0,) + fut.result()
        if tail < n:
            yield from self._members(tail)

    # -- generic gzip: one libdeflate call per member ----------------------------------------------
    def _inflate_member_at(self, pos: int, cap: int, grow: bool = True):
        """One member that starts at ``pos`` -> (rc, bytes consumed, buffer, bytes produced); rc 0 = fine, 3 = does
        not fit ``_MEMBER_CAP`` (``grow`` False: does not fit ``cap``), anything else = no gzip member starts here (or
        it is corrupt)."""
        base = _view_address(memoryview(self.map))
        H = _pinflate() if os.environ.get("CUTSEQ_OWN_INFLATE", "1") != "0" else None
        if H is not None:
            # the host library's byte-mode decoder (csrc/pinflate.c: whole matches in one table entry; 1.06 GB/s per
            # thread with the CRC behind it where libdeflate 1.10 does 0.93 -- tools/micro/member_rate.py); header,
            # trailer and CRC-32 are checked here
            hdr = _gzip_header_size(self.map, pos)
            if hdr == 0:
                return 1, 0, None, 0
            end_bit, produced = C.c_int64(), C.c_int64()
            out = self.take(cap)
            while True:
                rc = H.csh_inflate_stream(base + pos + hdr, self.size - pos - hdr, 0, out.ctypes.data, out.size,
                                          C.byref(end_bit), C.byref(produced))
                if rc == -2 and grow and out.size < _MEMBER_CAP:
                    self.give(out)
                    out = self.take(_MEMBER_CAP)
                    continue
                break
            if rc != 0:
                self.give(out)
                return (3 if rc == -2 else 1), 0, None, 0
            tail = pos + hdr + (end_bit.value + 7) // 8
            n_out = produced.value
            L = libdeflate()
            if tail + 8 <= self.size:
                want_crc, want_size = struct.unpack_from("<II", self.map, tail)
                crc = L.libdeflate_crc32(0, out.ctypes.data, n_out) if L is not None else zlib.crc32(memoryview(out)[:n_out])
            if tail + 8 > self.size or crc != want_crc or (n_out & 0xFFFFFFFF) != want_size:
                self.give(out)
                return 1, 0, None, 0
            self._posted(out, n_out)
            return 0, tail + 8 - pos, out, n_out
        L, d = libdeflate(), _decompressor()
        n_in, n_out = C.c_size_t(), C.c_size_t()
        out = self.take(cap)
        while True:
            rc = L.libdeflate_gzip_decompress_ex(d, base + pos, self.size - pos, out.ctypes.data, out.size, C.byref(n_in),
                                                 C.byref(n_out))
            if rc == 3 and grow and out.size < _MEMBER_CAP:
                self.give(out)
                out = self.take(_MEMBER_CAP)
                continue
            break
        if rc != 0:
            self.give(out)
            return rc, 0, None, 0
        self._posted(out, int(n_out.value))
        return 0, int(n_in.value), out, int(n_out.value)

    def _members(self, start: int = 0) -> Iterator[tuple]:
        if self.pool is None:
            yield from self._members_serial(start)
        else:
            yield from self._members_speculative(start)

    def _members_speculative(self, start: int) -> Iterator[tuple]:
        """Members inflate in parallel although their boundaries are only known once the member in front has been
        inflated: every ``1f 8b 08`` with clean flag bits behind the current member is tried as a member start in
        the pool; when the current member is done, the next one is usually finished or under way already.  A false
        candidate fails in its header or a few blocks in and is thrown away; order and content are those of the
        serial walk (the chain only ever follows ``start + bytes consumed``)."""
        buf, n = self.map, self.size
        H = _pinflate()
        mbase = _view_address(memoryview(buf)) if H is not None else 0
        # candidates in flight beyond the current member: enough to keep the whole pool busy from ONE file (single-end
        # runs; eight left half of a 16-thread pool idle: 3.0 GB/s of text from one file against 6.8 from two)
        # (two files of a paired run share the pool: half each -- more only queues speculation in front of the members the
        # other file is waiting for)
        window = max(8, getattr(self.pool, "_max_workers", 8) // max(1, GzipSource._active))
        tasks = {}  # member start -> future of _inflate_member_at
        cands = []  # ascending candidate starts behind the current member
        scan_from = start + 1
        cap = [min(32 << 20, _MEMBER_CAP)]  # output size to start with: the largest member seen so far

        def submit(pos):
            if pos not in tasks:
                tasks[pos] = self.pool.submit(self._inflate_member_at, pos, cap[0], False)

        running = []  # dropped candidates that had started already: they still read the mapped file

        def drop(pos):  # a candidate the chain walked past: give its buffer back when (if) it finishes
            fut = tasks.pop(pos, None)
            if fut is not None and not fut.cancel():
                running.append(fut)
                fut.add_done_callback(lambda f: f.exception() is None and f.result()[2] is not None and self.give(f.result()[2]))

        head = start
        try:
            while head < n:
                if buf[head] == 0 and not any(buf[head:min(n, head + (1 << 20))]):
                    return  # zero padding after the last member (tar-style): done
                # (member starts are looked for in the next 64 MB of compressed bytes only: mmap.find holds the GIL, and on
                # one huge member -- no magic anywhere -- it would walk the whole file while every other thread waits)
                ahead = min(n, head + _MEMBER_CAP // 4 + 2)
                while len(cands) < window and scan_from < ahead:
                    # (the host library's memchr walk, interpreter lock released; mmap.find without the library)
                    p = (H.csh_find_gzip_magic(mbase, scan_from, min(ahead, n - 2)) if H is not None
                         else buf.find(b"\x1f\x8b\x08", scan_from, ahead))
                    if p < 0:
                        scan_from = max(scan_from, ahead - 2)
                        break
                    scan_from = p + 1
                    if p + 18 <= n and (buf[p + 3] & 0xE0) == 0:
                        cands.append(p)
                if not cands and head not in tasks:
                    # No member starts anywhere near: one huge member, the usual sequencer output.  Inflating its first
                    # 32 MB only to learn that it does not fit costs the run 30 ms with every other thread idle; it
                    # cannot fit when its compressed bytes alone exceed the buffer (deflate never expands by more than
                    # a few bytes per block), or when it is the file's last member and the trailer says so.
                    span = ahead - head
                    isize = struct.unpack_from("<I", buf, n - 4)[0] if ahead == n and n - head >= 18 else 0
                    if span > cap[0] or isize > cap[0]:
                        yield from self._big_member(head)
                        return
                submit(head)
                for p in cands:
                    submit(p)
                rc, used, out, produced = tasks.pop(head).result()
                if rc == 3 and cap[0] < _MEMBER_CAP and any(
                        tasks[p].result()[0] in (0, 3) for p in cands if head < p <= head + _MEMBER_CAP // 4 and p in tasks):
                    # bigger than the members seen so far, and another member DOES start not far behind it (its own
                    # inflate got somewhere: the bytes 1f 8b 08 turn up by chance a couple of dozen times per gigabyte of
                    # compressed data, and such a place fails in its header or a few blocks in -- round 5: one of them
                    # within 64 MB of a huge member's start cost the run a quarter of a gigabyte of pointless inflate,
                    # 0.2 s of a 0.4 s run): once more with the largest buffer.  (No member anywhere near: one huge member,
                    # the usual sequencer output.)
                    cap[0] = _MEMBER_CAP
                    rc, used, out, produced = self._inflate_member_at(head, _MEMBER_CAP, False)
                if rc == 3:  # one very large member: its chunks decode in parallel (csrc/pinflate.c), or zlib streams it
                    for p in list(tasks):
                        drop(p)
                    yield from self._big_member(head)
                    return
                if rc != 0:
                    raise OSError(f"{self.path}: corrupt gzip data (libdeflate error {rc})")
                cap[0] = max(cap[0], min(_MEMBER_CAP, 1 << max(produced - 1, 1).bit_length()))
                yield head, out, produced
                head += used
                while cands and cands[0] <= head:
                    p = cands.pop(0)
                    if p < head:
                        drop(p)
                scan_from = max(scan_from, head + 1)
        finally:
            for p in list(tasks):
                drop(p)
            for fut in running:  # nothing may touch the map once the caller closes the source
                try:
                    fut.result()
                except Exception:  # noqa: BLE001 -- a false candidate; its error is nobody's business
                    pass

    def _members_serial(self, start: int = 0) -> Iterator[tuple]:
        L, d = libdeflate(), _decompressor()
        buf, n = self.map, self.size
        base = _view_address(memoryview(buf))
        n_in, n_out = C.c_size_t(), C.c_size_t()
        pos = start
        cap = 32 << 20  # grows to the largest member seen so far
        while pos < n:
            if buf[pos] == 0 and not any(buf[pos:min(n, pos + (1 << 20))]):
                return  # zero padding after the last member (tar-style): done
            out = self.take(cap)
            while True:
                rc = L.libdeflate_gzip_decompress_ex(d, base + pos, n - pos, out.ctypes.data, out.size, C.byref(n_in),
                                                     C.byref(n_out))
                if rc == 3 and out.size < _MEMBER_CAP:
                    self.give(out)
                    cap = _MEMBER_CAP
                    out = self.take(cap)
                    continue
                break
            if rc == 3:  # one very large member: stream IT through zlib, then go on member by member (a reader
                self.give(out)  # with a stop offset -- a rank's share, textio._members_until -- needs real offsets again)
                pos = yield from self._from_head(self._zlib_member(pos), pos)
                continue
            if rc != 0:
                self.give(out)
                raise OSError(f"{self.path}: corrupt gzip data (libdeflate error {rc})")
            at = pos
            pos += n_in.value
            self._posted(out, int(n_out.value))
            yield at, out, int(n_out.value)

    # -- one huge member: chunks of the compressed bytes decode side by side (csrc/pinflate.c) ------------------
    def _big_member(self, head: int) -> Iterator[tuple]:
        """The member at ``head`` is too large for one libdeflate call.  With a pool: decode it in parallel (and go on
        with whatever follows it); without one, or without the host library: stream everything from here through zlib."""
        H = _pinflate() if self.pool is not None else None
        hdr = _gzip_header_size(self.map, head) if H is not None else 0
        if H is None or hdr == 0 or os.environ.get("CUTSEQ_PARALLEL_INFLATE", "1") == "0":
            end = yield from self._from_head(self._zlib_member(head), head)
        else:
            end = yield from self._from_head(self._parallel_member(H, head + hdr), head)
        if end < self.size:
            yield from self._members(end)

    @staticmethod
    def _from_head(gen, head: int):
        """The blocks of one big member: the FIRST one carries the member's own start offset (a reader with a stop
        offset sees where it is), the others stay -1 (nobody can enter there)."""
        first = True
        while True:
            try:
                item = next(gen)
            except StopIteration as done:
                return done.value
            if first:
                item, first = (head,) + tuple(item[1:]), False
            yield item

    def _parallel_member(self, H, dstart: int):
        """Generator over the blocks of the deflate stream that starts at byte ``dstart``; returns the file offset behind
        the member's trailer.  Chunk i of the compressed bytes is decoded speculatively from the first block boundary the
        finder sees behind ``i * S`` -- into 16-bit symbols, references into the unknown 32 KB in front written as
        markers -- and ACCEPTED only if the chain of proven positions arrives at exactly that bit; otherwise the chunk is
        decoded again from the proven position, serially.  Markers are resolved once the window in front is known."""
        import numpy as np
        from collections import deque
        base = _view_address(memoryview(self.map)) + dstart
        n = self.size - dstart
        workers = max(2, getattr(self.pool, "_max_workers", 4))
        S = max(1 << 20, min(4 << 20, n // (4 * workers)))  # compressed bytes per chunk
        n_chunks = max(1, (n + S - 1) // S)
        sym_cap = [8 * S]  # symbols per chunk buffer; grows with the ratio seen

        def decode(start_bit, stop_bit):
            """-> (rc, end_bit, final, symbols (uint8 array viewed as uint16), n_symbols)"""
            cap = sym_cap[0]
            while True:
                raw = self.take(2 * cap)
                sym = raw[: 2 * cap].view(np.uint16)
                eb, no, fin = C.c_int64(), C.c_int64(), C.c_int32()
                rc = H.csh_inflate_chunk(base, n, start_bit, stop_bit, sym.ctypes.data, cap, C.byref(eb), C.byref(no), C.byref(fin))
                if rc == -2 and cap < (1 << 30):  # more text than expected in this chunk
                    self.give(raw)
                    cap *= 4
                    sym_cap[0] = max(sym_cap[0], cap)
                    continue
                if rc != 0:
                    self.give(raw)
                    return rc, 0, 0, None, 0
                return 0, eb.value, fin.value, raw, no.value

        import time
        T = self.stats  # (diagnostics: seconds per phase, summed over the threads)

        def tick(key, t0):
            t1 = time.perf_counter()
            T[key] = T.get(key, 0.0) + (t1 - t0)
            return t1

        def speculate(i):
            t0 = time.perf_counter()
            lo = i * S * 8
            start = 0 if i == 0 else H.csh_deflate_find_block(base, n, lo, min(n * 8, lo + 2 * S * 8))
            t0 = tick("find_s", t0)
            if start < 0:
                return start, (-1, 0, 0, None, 0)
            got = decode(start, (i + 1) * S * 8)
            tick("decode_s", t0)
            return start, got

        def resolve(raw, n_sym, window):
            t0 = time.perf_counter()
            sym = raw[: 2 * n_sym].view(np.uint16)
            out = self.take(max(n_sym, 1))
            markers = H.csh_resolve_markers(sym.ctypes.data, n_sym, window.ctypes.data if window is not None else None, out.ctypes.data)
            self.give(raw)
            t0 = tick("resolve_s", t0)
            if markers < 0:
                self.give(out)
                raise OSError(f"{self.path}: corrupt gzip data (a back-reference in front of the stream's start)")
            # (carry-less multiply: 8 GB/s; the zlib 1.2.11 Python links does 1 GB/s -- as long as the resolve pass itself)
            L = libdeflate()
            crc = L.libdeflate_crc32(0, out.ctypes.data, n_sym) if L is not None else zlib.crc32(memoryview(out)[:n_sym])
            t0 = tick("crc_s", t0)
            self._posted(out, n_sym)
            tick("post_s", t0)
            return out, n_sym, crc

        ahead = 2 * workers
        spec = {}      # chunk index -> future of speculate
        resolved = deque()  # futures of resolve, in stream order
        crcs, lens = [], []
        pos, window, final, i = 0, None, False, 0
        next_submit = 0
        try:
            while not final:
                while next_submit < n_chunks and next_submit < i + ahead:
                    spec[next_submit] = self.pool.submit(speculate, next_submit)
                    next_submit += 1
                if i < n_chunks:
                    t0 = time.perf_counter()
                    start, (rc, end_bit, fin, raw, n_sym) = spec.pop(i).result()
                    tick("main_wait_s", t0)
                else:
                    start, rc, raw = -1, -1, None
                self.stats["chunks"] = self.stats.get("chunks", 0) + 1
                if start != pos or rc != 0:
                    self.stats["serial"] = self.stats.get("serial", 0) + 1
                    if os.environ.get("CUTSEQ_DEBUG_INFLATE") == "1":  # diagnostic: which chunk lost its proof, and how
                        import sys
                        print(f"inflate: chunk {i} of {n_chunks} (S = {S}): proven bit {pos}, speculative start {start}, rc {rc}", file=sys.stderr)
                    # not proven (finder missed the boundary, saw a false one, or the chunk overflowed): from the proven
                    # position, serially
                    if raw is not None:
                        self.give(raw)
                    rc, end_bit, fin, raw, n_sym = decode(pos, (i + 1) * S * 8)
                    if rc != 0:
                        raise OSError(f"{self.path}: corrupt gzip data (deflate stream, error {rc})")
                nxt = np.empty(32768, dtype=np.uint8)
                sym = raw[: 2 * n_sym].view(np.uint16)
                if H.csh_next_window(sym.ctypes.data, n_sym, window.ctypes.data if window is not None else None, nxt.ctypes.data) < 0:
                    raise OSError(f"{self.path}: corrupt gzip data (a back-reference in front of the stream's start)")
                resolved.append(self.pool.submit(resolve, raw, n_sym, window))
                window, pos, final = nxt, end_bit, bool(fin)
                i += 1
                while resolved and (len(resolved) > ahead or resolved[0].done()):
                    out, nbytes, crc = resolved.popleft().result()
                    crcs.append(crc)
                    lens.append(nbytes)
                    if nbytes:
                        yield -1, out, nbytes
                    else:
                        self.give(out)
            while resolved:
                out, nbytes, crc = resolved.popleft().result()
                crcs.append(crc)
                lens.append(nbytes)
                if nbytes:
                    yield -1, out, nbytes
                else:
                    self.give(out)
        finally:
            for fut in spec.values():  # chunks decoded beyond the end of the member (or abandoned on an error)
                if not fut.cancel():
                    try:
                        raw = fut.result()[1][3]
                        if raw is not None:
                            self.give(raw)
                    except Exception:  # noqa: BLE001
                        pass
            for fut in resolved:
                try:
                    self.give(fut.result()[0])
                except Exception:  # noqa: BLE001
                    pass
        tail = dstart + (pos + 7) // 8
        if tail + 8 > self.size:
            raise OSError(f"{self.path}: truncated gzip member (no trailer)")
        want_crc, want_size = struct.unpack_from("<II", self.map, tail)
        total = sum(lens)
        k = len(crcs)
        crc = int(H.csh_crc32_combine_many((C.c_uint32 * max(k, 1))(*crcs), (C.c_uint32 * max(k, 1))(*lens), k))
        if crc != want_crc or (total & 0xFFFFFFFF) != want_size:
            raise OSError(f"{self.path}: corrupt gzip data (CRC-32 / size mismatch)")
        return tail + 8

    # -- fallback: zlib streaming, any member size -------------------------------------------------
    def _zlib_member(self, start: int):
        """Generator over the (-1, buffer, nbytes) blocks of the ONE member at ``start``, through zlib; returns the file
        offset behind its trailer, where the member-by-member walk goes on with real offsets (a rank whose share ends
        there must not read on into the next rank's members: ADVICE r4)."""
        end = start
        for item in self._zlib_stream(start, single=True):
            if isinstance(item, int):
                end = item
            else:
                yield -1, item[0], item[1]
        return end

    def _zlib_stream(self, start: int, single: bool = False) -> Iterator[tuple]:
        """(buffer, nbytes) blocks of everything from ``start`` to the end of the file; ``single``: of the member at
        ``start`` only, followed by one int -- the offset behind that member."""
        import numpy as np
        buf, n = self.map, self.size
        pos = start
        d = zlib.decompressobj(31)
        fresh = True  # the current decompressor has not seen a byte yet
        while pos < n:
            data = buf[pos:pos + (1 << 20)]
            pos += len(data)
            while data:
                if fresh and data[0] == 0 and not any(data) and not any(buf[pos:min(n, pos + (1 << 20))]):
                    if single:
                        yield n
                    return  # zero padding behind the last member
                try:
                    out = d.decompress(data, _STREAM_BLOCK)
                except zlib.error as exc:
                    raise OSError(f"{self.path}: corrupt gzip data ({exc})") from exc
                fresh = False
                if out:
                    yield np.frombuffer(out, dtype=np.uint8), len(out)
                if d.eof:
                    data = d.unused_data
                    if single:
                        yield pos - len(data)
                        return
                    d = zlib.decompressobj(31)
                    fresh = True
                else:
                    data = d.unconsumed_tail
        if not fresh:
            raise OSError(f"{self.path}: truncated gzip stream")
        if single:
            yield n


def is_gzip(path: str) -> bool:
    with open(path, "rb") as fh:
        return fh.read(2) == b"\x1f\x8b"


# ---------------------------------------------------------------------------------------------------------------
# The other containers xopen gives the reference for nothing (cutseq/run.py:434, 437, 751, 754: InputPaths /
# OutputFiles): bzip2, xz, zstandard (where a binding is importable) and "-" for standard input / output.  None of
# them is a fast path -- the stdlib codecs run on one thread -- they feed the same block reader / ordered writers.

_MAGIC = {b"\x1f\x8b": "gzip", b"BZh": "bz2", b"\xfd7zXZ\x00": "xz", b"\x28\xb5\x2f\xfd": "zst"}
STREAM_BLOCK = 4 << 20


def container_of_magic(head: bytes) -> str:
    for magic, name in _MAGIC.items():
        if head.startswith(magic):
            return name
    return "plain"


def container_of_name(path: str) -> str:
    """Output side: xopen goes by the file name's extension."""
    low = path.lower()
    for ext, name in ((".gz", "gzip"), (".bz2", "bz2"), (".xz", "xz"), (".zst", "zst")):
        if low.endswith(ext):
            return name
    return "plain"


def _zstd():
    try:
        import zstandard  # noqa: PLC0415
        return zstandard
    except ImportError:
        raise ValueError("zstandard (.zst) files need the 'zstandard' Python module, which is not installed") from None


def open_decoder(raw, container: str):
    """A binary file object with the decompressed bytes of ``raw`` (a binary file object)."""
    if container == "plain":
        return raw
    if container == "gzip":
        import gzip
        return gzip.GzipFile(fileobj=raw, mode="rb")
    if container == "bz2":
        import bz2
        return bz2.BZ2File(raw, "rb")
    if container == "xz":
        import lzma
        return lzma.LZMAFile(raw, "rb")
    if container == "zst":
        return _zstd().ZstdDecompressor().stream_reader(raw, read_across_frames=True)
    raise ValueError(container)


def make_compressor(container: str, level: int = 1):
    """-> (compress(bytes-like) -> bytes, flush() -> bytes) for a sequential output stream."""
    if container == "bz2":
        import bz2
        c = bz2.BZ2Compressor(max(1, min(9, level)))
        return c.compress, c.flush
    if container == "xz":
        import lzma
        c = lzma.LZMACompressor(preset=max(0, min(9, level)))
        return c.compress, c.flush
    if container == "zst":
        c = _zstd().ZstdCompressor(level=level).compressobj()
        return c.compress, c.flush
    raise ValueError(container)


class StreamSource:
    """Blocks of decompressed text from a sequential reader: standard input, bzip2 / xz / zstandard files, and plain
    or gzip files when a wrapper (FASTA) needs the bytes in order.  Same face as :class:`GzipSource`."""

    def __init__(self, fileobj, take=_np_take, give=_np_give, owner=None):
        self.f, self.take, self.give, self.owner = fileobj, take, give, owner

    def side(self, arr):
        return None

    def close(self):
        for f in (self.f, self.owner):
            if f is not None and f is not sys.stdin.buffer:
                try:
                    f.close()
                except Exception:  # noqa: BLE001
                    pass

    def blocks(self, start: int = 0) -> Iterator[tuple]:
        assert start == 0, "a sequential stream has no entry points"
        while True:
            arr = self.take(STREAM_BLOCK)
            mv = memoryview(arr)[:STREAM_BLOCK]
            got = 0
            while got < STREAM_BLOCK:
                n = self.f.readinto(mv[got:])
                if not n:
                    break
                got += n
            if got == 0:
                self.give(arr)
                return
            yield arr, got
            if got < STREAM_BLOCK:
                return


class _Peeked:
    """A binary reader with some bytes already taken off its front (sniffing standard input)."""

    def __init__(self, head: bytes, rest):
        self.head, self.rest = head, rest

    def readinto(self, b) -> int:
        if self.head:
            n = min(len(b), len(self.head))
            b[:n] = self.head[:n]
            self.head = self.head[n:]
            return n
        return self.rest.readinto(b)

    def read(self, n: int = -1) -> bytes:
        if self.head:
            out, self.head = (self.head, b"") if n < 0 or n >= len(self.head) else (self.head[:n], self.head[n:])
            if n < 0:
                return out + self.rest.read()
            return out
        return self.rest.read(n)

    def readable(self):
        return True

    def close(self):
        pass

    @property
    def closed(self):
        return False

    def seekable(self):
        return False

    def flush(self):
        pass


def sniff_input(path: str):
    """-> (container, first byte of the TEXT behind leading white space, opener) where ``opener()`` gives a fresh binary
    reader of the decompressed text.  ``path`` "-" is standard input (compression recognised by its magic bytes, as for
    files)."""
    if path == "-":
        raw = sys.stdin.buffer
        head = raw.read(8) or b""
        container = container_of_magic(head)
        peeked = _Peeked(head, raw)
        dec = open_decoder(peeked, container)
        text = dec.read(65536) or b""
        first = text.lstrip()[:1]
        stream = _Peeked(text, dec)
        return container, first, (lambda: stream)
    with open(path, "rb") as fh:
        head = fh.read(8)
    container = container_of_magic(head)

    def opener():
        return open_decoder(open(path, "rb"), container)

    with opener() as dec:
        text = dec.read(65536) or b""
    return container, text.lstrip()[:1], opener


class FastaSource:
    """FASTA text re-shaped into four-line records on its way in (csh_fasta_to_fastq in csrc/cutseq_host.c: names and
    joined sequence lines as they are, a quality line no cutoff trims) -- the device parses one record shape."""

    def __init__(self, inner, convert, take=_np_take, give=_np_give, label="input"):
        self.inner, self.convert, self.take, self.give, self.label = inner, convert, take, give, label

    def side(self, arr):
        return None

    def close(self):
        self.inner.close()

    def blocks(self, start: int = 0) -> Iterator[tuple]:
        # What goes to the converter always ENDS IN FRONT OF A RECORD START (a '>' at the start of a line): the record
        # that is still open when a block ends waits in `pending`, piece by piece, and is converted once -- when the
        # block with the next record start (or the end of the input) arrives.  (Round 4 prepended the open record to every
        # new block and converted it again from its start: sixty passes over a 250 MB contig)
        pending: list = []   # the open record's pieces (bytes)
        line0 = 0
        gen = self.inner.blocks(start)
        done = False
        while not done:
            item = next(gen, None)
            if item is None:
                done = True
                data, pending = b"".join(pending), []
            else:
                arr, nbytes = item
                block = bytes(memoryview(arr)[:nbytes])
                self.give(arr)
                ends_line = (pending[-1][-1:] == b"\n") if pending else True
                cut = block.rfind(b"\n>")
                cut = cut + 1 if cut >= 0 else (0 if block[:1] == b">" and ends_line and pending else -1)
                if cut < 0 or (cut == 0 and not pending):
                    pending.append(block)  # no record starts in here (or only the block's first one): the record stays open
                    continue
                data, pending = b"".join(pending) + block[:cut], [block[cut:]]
            if not data:
                continue
            out = self.take(3 * len(data) + 64)
            produced, consumed, err_line = self.convert(data, out, True)
            if produced == -2:
                self.give(out)
                raise ValueError(f"{self.label}: FASTA format error in line {line0 + err_line}: expected '>' at the start of a record")
            if produced < 0 or consumed != len(data):
                self.give(out)
                raise MemoryError("FASTA conversion buffer too small")
            line0 += data.count(b"\n")
            if produced:
                yield out, produced
            else:
                self.give(out)
