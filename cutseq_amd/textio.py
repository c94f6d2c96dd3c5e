"""Host side of the text path: blocks of whole FASTQ records in, output streams out.

What is left for the host once the device parses and formats (``cs_text_*``, ``textpath.py``): read (or inflate)
the input, cut it into blocks of exactly ``chunk_reads`` records -- two vectorised passes over the bytes, newline
count and k-th newline, no per-record work -- and write (or deflate) the finished output text, strictly in input
order.  Counterpart of ``runner.run(pipeline, Progress(), outfiles)`` with ``make_runner(cores=N)``
(cutseq/run.py:436, 473, 753, 794): reader, chunk fan-out to workers, ordered merge.

  TextReader    one thread per input file: page cache (several ``pread`` calls at once) or inflate pool -> pinned
                block buffer -> ``TextBlock(buf, nbytes, n_records)``
  TextWorker    one thread per GPU: a ``TrimEngine`` + ``TextEngine``, three batches in flight; grows the row stride
                or the text capacity when a batch needs it
  run_text_pipeline   pairs the mates' blocks, deals them round-robin, re-orders the results and feeds one
                ``StreamWriter`` per output file (plain: parallel ``pwrite``; ``.gz``: 4 MB gzip members from the pool)
"""
from __future__ import annotations

import ctypes as C
import mmap
import os
import queue
import threading
import time
from collections import deque
from typing import List, Optional

import numpy as np

from . import abi, codec, fastq, report, shard, textpath

CHUNK_READS = 1 << 18
_BLOCK = 8 << 20          # bytes per pread / per inflate hand-over
_GZ_PIECE = 4 << 20       # output text per gzip member
_COPY_PIECE = 4 << 20     # plain output: bytes per parallel copy into the file mapping
_MAP_MIN = 8 << 20        # ... used from this size on (smaller pieces go through pwrite)
_PAGE = mmap.ALLOCATIONGRANULARITY


PROFILE = os.environ.get("CUTSEQ_PROFILE") == "1"
_DISCARD = False
_prof = {}
_prof_lock = threading.Lock()


def _tick(key: str, t0: float) -> float:
    """Diagnostic (CUTSEQ_PROFILE=1): seconds per thread and activity, printed at the end of the run."""
    now = time.perf_counter()
    if PROFILE:
        name = f"{threading.current_thread().name}:{key}"
        with _prof_lock:
            _prof[name] = _prof.get(name, 0.0) + (now - t0)
    return now


def _host():
    L = fastq._lib()
    if not getattr(L, "_text_bound", False):
        L.csh_count_newlines.restype = C.c_int64
        L.csh_count_newlines.argtypes = [C.c_void_p, C.c_int64]
        L.csh_after_kth_newline.restype = C.c_int64
        L.csh_after_kth_newline.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.csh_fasta_to_fastq.restype = C.c_int64
        L.csh_fasta_to_fastq.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L._text_bound = True
    return L


class TextBlock:
    """``n_records`` complete records in ``buf[:nbytes]`` (a pinned arena buffer)."""

    __slots__ = ("buf", "nbytes", "n", "first_record")

    def __init__(self, buf: np.ndarray, nbytes: int, n: int, first_record: int):
        self.buf, self.nbytes, self.n, self.first_record = buf, nbytes, n, first_record

    def release(self) -> None:
        if self.buf is not None:
            fastq.PINNED.give(self.buf)
            self.buf = None


class TextReader(threading.Thread):
    """One input file -> :class:`TextBlock` objects of exactly ``chunk_reads`` records (the last one may be shorter),
    then ``None``.  Exceptions travel through the queue."""

    def __init__(self, path: str, chunk_reads: int, start: int = 0, skip_lines: int = 0, max_records: Optional[int] = None,
                 stop: Optional[int] = None):
        """``start`` / ``skip_lines`` / ``max_records`` / ``stop``: one rank's share of the file in the multi-process form
        (``ranks.py``): begin at byte ``start`` (plain text) or at the gzip member that starts there, drop
        ``skip_lines`` lines, stop behind ``max_records`` records (None = to the end of the file) or in front of the
        gzip member that starts at compressed offset ``stop``."""
        super().__init__(daemon=True, name=f"cutseq-read-{os.path.basename(path)}")
        self.path, self.chunk_reads = path, chunk_reads
        self.start_at, self.skip_lines, self.max_records, self.stop_at = start, skip_lines, max_records, stop
        self.blocks: "queue.Queue" = queue.Queue(maxsize=3)
        self._halt = False
        # What the file holds decides how it is read (dnaio / xopen go by content too, cutseq/run.py:434-441, 751-758):
        # gzip and plain FASTQ files take the parallel paths; standard input ("-"), bzip2 / xz / zstandard files and
        # FASTA text come through a sequential reader (codec.StreamSource), FASTA re-shaped into four-line records.
        self.container, first, self._opener = codec.sniff_input(path)
        self.fasta = first in (b">", b"#")  # (dnaio: a leading comment line means FASTA too)
        self.sequential = path == "-" or self.container in ("bz2", "xz", "zst") or (self.fasta and self.container == "plain")
        self.gz = self.container == "gzip" and not self.sequential
        if (self.sequential or self.fasta or (stop is not None and not self.gz)) and (
                start or skip_lines or max_records is not None or stop is not None):
            raise ValueError(f"{path}: only plain and gzip FASTQ files can be split between ranks")
        self.start()

    # -- plumbing ---------------------------------------------------------------------------------------
    def _put(self, item) -> bool:
        t0 = time.perf_counter()
        try:
            while not self._halt:
                try:
                    self.blocks.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False
        finally:
            _tick("blocked_on_consumer", t0)

    def get(self) -> Optional[TextBlock]:
        item = self.blocks.get()
        if isinstance(item, fastq._Failure):
            raise item.exc
        return item

    def close(self) -> None:
        self._halt = True
        deadline = time.monotonic() + 30.0
        while self.is_alive() and time.monotonic() < deadline:
            try:
                while True:
                    item = self.blocks.get_nowait()
                    if isinstance(item, TextBlock):
                        item.release()
            except queue.Empty:
                pass
            self.join(timeout=0.05)

    # -- sources: append text behind buf[:fill] -> (buf, fill, [(offset, nbytes, newlines)], eof) ------------------
    @staticmethod
    def _room(buf: np.ndarray, fill: int, want: int) -> np.ndarray:
        if buf.size - fill >= want:
            return buf
        bigger = fastq.PINNED.take(max(buf.size + buf.size // 2, fill + want))
        C.memmove(bigger.ctypes.data, buf.ctypes.data, fill)
        fastq.PINNED.give(buf)
        return bigger

    @staticmethod
    def _copy_in(dst: int, text: np.ndarray, nbytes: int) -> None:
        C.memmove(dst, text.ctypes.data, nbytes)
        if isinstance(text.base, fastq.mmap.mmap):
            fastq.ARENA.give(text)

    @staticmethod
    def _members_until(src, start: int, stop: int):
        """The blocks of the members in [start, stop) (compressed offsets; the share of a rank that splits by members)."""
        gen = src.indexed_blocks(start)
        try:
            for off, arr, nbytes in gen:
                if off >= stop:
                    src.give(arr)
                    return
                yield arr, nbytes
        finally:
            gen.close()

    @staticmethod
    def _fasta_convert(data: bytes, out: np.ndarray, final: bool):
        """-> (bytes written | -1 | -2, bytes of ``data`` consumed, line of a format error)"""
        consumed, records, err_line = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        produced = _host().csh_fasta_to_fastq(data, len(data), out.ctypes.data, out.size, 1 if final else 0, 0,
                                              C.byref(consumed), C.byref(records), C.byref(err_line))
        return int(produced), int(consumed.value), int(err_line.value)

    def _read_plain(self, fd: int, pos: int, buf: np.ndarray, fill: int, want: int):
        """``want`` bytes of the file from ``pos`` on, several preads (and newline counts) at once in the pool."""
        L = _host()
        pieces = max(1, min(16, (want + _BLOCK - 1) // _BLOCK))  # (a block of 262 144 short-read records: 11 pieces, one round)
        want = pieces * _BLOCK
        buf = self._room(buf, fill, want)
        base = buf.ctypes.data
        mv = memoryview(buf)

        def job(i):
            off = fill + i * _BLOCK
            got = 0
            while got < _BLOCK:  # a short read is not the end of the file (FUSE, network mounts): only 0 bytes is
                more = os.preadv(fd, [mv[off + got:off + _BLOCK]], pos + i * _BLOCK + got)
                if more <= 0:
                    break
                got += more
            return got, (int(L.csh_count_newlines(base + off, got)) if got else 0)

        pool = fastq._pool()
        results = [f.result() for f in [pool.submit(job, i) for i in range(pieces)]] if pieces > 1 else [job(0)]
        marks, eof, total = [], False, 0
        for i, (got, lines) in enumerate(results):
            if eof and got:  # a short piece in the middle: the file changed under us
                raise OSError(f"{self.path}: short read")
            if got:
                marks.append((fill + i * _BLOCK, got, lines))
                total += got
            if got < _BLOCK:
                eof = True
        return buf, fill + total, marks, eof

    def _run(self):
        L = _host()
        need = 4 * self.chunk_reads
        est = 400  # bytes per record, corrected by every block that goes out
        buf = fastq.PINNED.take(int(self.chunk_reads * est * 1.25) + 2 * _BLOCK)
        fill, lines, done = 0, 0, 0
        marks: List[tuple] = []  # (offset, nbytes, newlines) of every piece behind buf[:fill]
        eof = False
        fd, pos, gen = -1, self.start_at, None
        skip = self.skip_lines
        left = self.max_records  # records still to hand out (None: everything)
        copies: List = []  # inflated blocks on their way into `buf` (copied by the pool, awaited before `buf` is read)
        if self.gz and not self.fasta:
            src = codec.GzipSource(self.path, fastq._pool(), fastq.ARENA.take, fastq.ARENA.give,
                                   post=lambda addr, nbytes: int(L.csh_count_newlines(addr, nbytes)))
            gen = src.blocks(self.start_at) if self.stop_at is None else self._members_until(src, self.start_at, self.stop_at)
        elif self.gz or self.sequential:
            if self.gz:  # (FASTA in a gzip file: the members still inflate in the pool)
                src = codec.GzipSource(self.path, fastq._pool(), fastq.ARENA.take, fastq.ARENA.give)
            else:
                src = codec.StreamSource(self._opener(), fastq.ARENA.take, fastq.ARENA.give)
            if self.fasta:
                src = codec.FastaSource(src, self._fasta_convert, fastq.ARENA.take, fastq.ARENA.give, self.path)
            gen = src.blocks(0)
        else:
            fd = os.open(self.path, os.O_RDONLY)
        try:
            while not self._halt:
                if left is not None:
                    if left == 0:
                        fastq.PINNED.give(buf)
                        buf = None
                        self._put(None)
                        return
                    need = 4 * min(self.chunk_reads, left)
                while skip and not eof and lines < skip:  # (the share starts `skip` lines into its first gzip member)
                    item = next(gen, None)
                    if item is None:
                        eof = True
                        break
                    text, nbytes = item
                    buf = self._room(buf, fill, nbytes)
                    C.memmove(buf.ctypes.data + fill, text.ctypes.data, nbytes)
                    if isinstance(text.base, fastq.mmap.mmap):
                        fastq.ARENA.give(text)
                    lines += int(L.csh_count_newlines(buf.ctypes.data + fill, nbytes))
                    fill += nbytes
                if skip and lines >= skip:
                    cut = int(L.csh_after_kth_newline(buf.ctypes.data, fill, skip))
                    C.memmove(buf.ctypes.data, buf.ctypes.data + cut, fill - cut)
                    fill -= cut
                    lines -= skip
                    marks = [(0, fill, lines)]
                    skip = 0
                while lines < need and not eof:
                    if gen is not None:
                        item = next(gen, None)
                        if item is None:
                            eof = True
                            break
                        text, nbytes = item
                        if buf.size - fill < nbytes:
                            for f in copies:
                                f.result()
                            copies = []
                            buf = self._room(buf, fill, nbytes)
                        got = src.side(text)
                        if got is None:
                            got = int(L.csh_count_newlines(text.ctypes.data, nbytes))
                        copies.append(fastq._pool().submit(self._copy_in, buf.ctypes.data + fill, text, nbytes))
                        marks.append((fill, nbytes, got))
                        fill += nbytes
                        lines += got
                    else:
                        missing = max(_BLOCK, int((need - lines) / 4 * est * 1.05) - 0)
                        t0 = time.perf_counter()
                        buf, fill, more, eof = self._read_plain(fd, pos, buf, fill, missing)
                        _tick("read", t0)
                        pos += sum(m[1] for m in more)
                        lines += sum(m[2] for m in more)
                        marks += more
                for f in copies:
                    f.result()
                copies = []
                if lines >= need:
                    # the block ends right behind newline number `need`: find the piece that holds it, then the byte
                    t0 = time.perf_counter()
                    cum, cut = 0, -1
                    for off, nbytes, got in marks:
                        if cum + got >= need:
                            cut = off + int(L.csh_after_kth_newline(buf.ctypes.data + off, nbytes, need - cum))
                            break
                        cum += got
                    assert cut > 0
                    nxt = fastq.PINNED.take(max(buf.size, int(self.chunk_reads * est * 1.25) + 2 * _BLOCK))
                    carry = fill - cut
                    C.memmove(nxt.ctypes.data, buf.ctypes.data + cut, carry)
                    block = TextBlock(buf, cut, need // 4, done)
                    done += need // 4
                    if left is not None:
                        left -= need // 4
                    est = max(64, cut // (need // 4) + 1)
                    buf, fill, lines = nxt, carry, lines - need
                    marks = [(0, carry, lines)]
                    _tick("cut+carry", t0)
                    if not self._put(block):
                        block.release()
                        return
                    continue
                # End of input: what is left must be whole records; blank lines behind the last one are tolerated.
                # The last NON-BLANK line decides: it is line 4k + 3 (a quality line: the record ends with it, the
                # device takes the end of the text for its missing line end) or line 4k + 2 (a '+' line: the record
                # has an empty read, "@id\n\n+\n\n", and its empty quality line is the first blank line behind it).
                # Only whole blank lines beyond that are dropped -- stripping all trailing white space took the
                # empty quality line with it and turned a valid file into "truncated".
                end = fill
                view = memoryview(buf)
                while end > 0 and view[end - 1] in b"\n\r \t":
                    end -= 1
                if end == 0:
                    fastq.PINNED.give(buf)
                    buf = None
                    self._put(None)
                    return
                last = int(L.csh_count_newlines(buf.ctypes.data, end))  # index of the last non-blank line
                if last % 4 == 3:
                    tail_lines = last + 1
                elif last % 4 == 2 and end < fill:
                    # the '+' line's own line end, then the empty quality line (its '\n' is written if the file lacks it)
                    tail = bytes(view[end:fill])  # white space only
                    nl1 = tail.find(b"\n")
                    if nl1 < 0:
                        raise fastq.FastqFormatError(f"{self.path}: truncated FASTQ record at end of file")
                    nl2 = tail.find(b"\n", nl1 + 1)
                    if nl2 >= 0:
                        end += nl2 + 1
                    else:
                        buf = self._room(buf, fill, 1)
                        buf[fill] = 0x0A
                        end = fill + 1
                    tail_lines = last + 2
                else:
                    raise fastq.FastqFormatError(f"{self.path}: truncated FASTQ record at end of file")
                block = TextBlock(buf, end, tail_lines // 4, done)
                buf = None
                if self._put(block):
                    self._put(None)
                else:
                    block.release()
                return
        finally:
            for f in copies:
                try:
                    f.result()
                except BaseException:
                    pass
            if buf is not None:
                fastq.PINNED.give(buf)
            if gen is not None:
                gen.close()
                src.close()
            if fd >= 0:
                os.close(fd)

    def run(self):
        try:
            self._run()
        except BaseException as exc:
            self._put(fastq._Failure(exc))


class _Shared:
    """Output buffers of one batch on their way to several files: released when the last writer is done."""

    def __init__(self, bufs, consumers: int, on_release):
        self.bufs, self.left, self.on_release = bufs, consumers, on_release
        self.lock = threading.Lock()
        if consumers == 0:
            self._release()

    def _release(self):
        for b in self.bufs:
            fastq.PINNED.give(b)
        self.bufs = ()
        self.on_release()

    def done(self):
        with self.lock:
            self.left -= 1
            last = self.left == 0
        if last:
            self._release()


_libc = None


def _bind_libc():
    global _libc
    if _libc is None:
        try:
            _libc = C.CDLL(None, use_errno=True)
            _libc.fallocate.restype = C.c_int
            _libc.fallocate.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64]
        except (OSError, AttributeError):
            _libc = False
    return _libc


def _fallocate(fd: int, offset: int, length: int) -> bool:
    """Linux fallocate(2), mode 0 (extends the file) -> False when the file system has no such thing.  (Not
    os.posix_fallocate: glibc emulates a missing fallocate by touching every block.)"""
    if not _bind_libc():
        return False
    if _libc.fallocate(fd, 0, offset, length) == 0:
        return True
    err = C.get_errno()
    import errno
    if err in (errno.EOPNOTSUPP, errno.ENOSYS, errno.EINVAL, errno.ENODEV, errno.ESPIPE):
        return False  # this file (system) cannot do it: the caller truncates instead
    raise OSError(err, os.strerror(err))  # disk full, file too large, ...: better here than as SIGBUS in a mapping


class StreamWriter:
    """One output file, written strictly in the order of :meth:`put`.  ``.gz``: the text goes out as gzip members
    of about 4 MB, compressed in the pool (level 1 = cutadapt's default; any split of the text is a valid gzip
    file and decompresses to the same bytes); plain: big pieces are written with several ``pwrite`` calls at once."""

    def __init__(self, path: str, level: int = 1, precompressed: bool = False):
        """``precompressed``: what arrives are finished gzip members (the device compressed them): written as they are."""
        # xopen's rules (cutseq/run.py:437, 754: OutputFiles): the container goes by the name's extension, "-" is
        # standard output.  gzip and plain files keep their parallel paths; bzip2 / xz / zstandard streams are
        # compressed by this writer's own thread (stdlib codecs), standard output takes sequential writes.
        self.path, self.level = path, level
        self.container = codec.container_of_name(path) if path != "-" else "plain"
        self.gz = self.container == "gzip"
        self.precompressed = precompressed and self.gz
        self.codec = codec.make_compressor(self.container, level) if self.container in ("bz2", "xz", "zst") else None
        self.stdout = path == "-"
        if self.stdout:
            import sys
            sys.stdout.flush()
            self.fd = os.dup(1)
        else:
            self.fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666)
        self.pos = 0
        # (gzip members are small: plain pwrite, the pool has better things to do)
        self.mappable = self.container == "plain" and not self.stdout
        self._can_allocate = True    # plain output: fallocate works on this file (system)
        self.q: "queue.Queue" = queue.Queue()
        self.err: Optional[BaseException] = None
        self.t = threading.Thread(target=self._run, daemon=True, name=f"cutseq-write-{os.path.basename(path)}")
        self.t.start()

    def put(self, view: memoryview, shared: _Shared) -> None:
        if self.err is not None:
            shared.done()
            raise self.err
        if self.gz and not self.precompressed:
            pool = fastq._pool()
            futs = [pool.submit(codec.gzip_member, view[lo:lo + _GZ_PIECE], self.level) for lo in range(0, len(view), _GZ_PIECE)]
            self.q.put((futs, shared))
        else:
            self.q.put((view, shared))

    def _write_all(self, data) -> None:
        """``data`` at the end of the file.  Writes to ONE file serialise on its inode lock whatever the number of
        threads, so big plain pieces do not go through write(2): the file is extended, the new range mapped, and the
        pool copies into the mapping -- page faults and copies of different pages run side by side."""
        if self.codec is not None:
            data = self.codec[0](data)
        mv = memoryview(data)
        n = len(mv)
        if _DISCARD:  # diagnostic (tools/host_io_profile.py): everything but the writers' copies
            self.pos += n
            return
        if self.stdout:
            at = 0
            while at < n:
                at += os.write(self.fd, mv[at:])
            self.pos += n
            return
        if n >= _MAP_MIN and self.mappable:
            try:
                self._copy_mapped(mv, n)
                self.pos += n
                return
            except (OSError, ValueError, BufferError):
                self.mappable = False  # a file system without usable shared mappings: plain writes from here on
                os.ftruncate(self.fd, self.pos)
        at = 0
        while at < n:
            at += os.pwrite(self.fd, mv[at:], self.pos + at)
        self.pos += n

    def _copy_mapped(self, mv: memoryview, n: int) -> None:
        start, end = self.pos, self.pos + n
        # the new range gets its pages in ONE call where the file system can do that (tmpfs, ext4, xfs: 18 GB/s on the
        # GPU box against 3-6 GB/s when the copies below fault them in one by one; tools/micro/tmpfs_write2.py).
        # Tried again in round 5 and dropped (profiles/r05_host_io.md): allocating the NEXT range meanwhile (fallocate and
        # copies into another range of the same file slow each other down: 16 -> 8 GB/s for two files), and having every
        # copy job map its piece first (madvise MADV_POPULATE_WRITE: +27 % for the writers alone, nothing inside the
        # pipeline, where the same 16 threads also read).
        if not (self._can_allocate and _fallocate(self.fd, start, n)):
            self._can_allocate = False
            os.ftruncate(self.fd, end)
        base = start - start % _PAGE
        mm = mmap.mmap(self.fd, end - base, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE, offset=base)
        try:
            dst = np.frombuffer(mm, dtype=np.uint8)
            src = np.frombuffer(mv, dtype=np.uint8)
            d0, s0 = dst.ctypes.data + (start - base), src.ctypes.data
            pool = fastq._pool()
            futs = [pool.submit(C.memmove, d0 + lo, s0 + lo, min(_COPY_PIECE, n - lo)) for lo in range(0, n, _COPY_PIECE)]
            for f in futs:
                f.result()
            del dst, src
        finally:
            try:
                mm.close()
            except BufferError:  # (an exception above left a view alive: the mapping goes with the garbage collector)
                pass

    def _run(self):
        while True:
            t0 = time.perf_counter()
            item = self.q.get()
            t0 = _tick("idle", t0)
            if item is None:
                return
            payload, shared = item
            try:
                if self.err is None:
                    if isinstance(payload, list):
                        for f in payload:
                            blob = f.result()
                            t0 = _tick("wait_deflate", t0)
                            self._write_all(blob)
                            t0 = _tick("write", t0)
                    else:
                        self._write_all(payload)
                        _tick("write", t0)
                elif isinstance(payload, list):
                    for f in payload:  # the views must not outlive their buffer
                        try:
                            f.result()
                        except BaseException:
                            pass
            except BaseException as exc:
                self.err = exc
            finally:
                shared.done()

    def close(self) -> None:
        self.q.put(None)
        self.t.join()
        try:
            if self.err is None and self.gz and self.pos == 0:
                self._write_all(codec.gzip_member(b"", self.level))  # an empty stream is still a valid gzip file
            if self.err is None and self.codec is not None:
                tail, self.codec = self.codec[1](), None
                self._write_all(tail)
        finally:
            os.close(self.fd)
        if self.err is not None:
            raise self.err


class _Done:
    __slots__ = ("k", "n", "res", "out", "sizes", "counts")

    def __init__(self, k, n, res, out, sizes=None, counts=None):
        self.k, self.n, self.res, self.out, self.sizes, self.counts = k, n, res, out, sizes, counts


class TextWorker(threading.Thread):
    """One GPU.  Counterpart of one worker process of ``make_runner(inpaths, cores=N)`` (cutseq/run.py:436, 753)."""

    SLOTS = 3

    def __init__(self, tp, device: int, done: "queue.Queue", chunk_reads: int, compress: bool = False, bins: int = 0,
                 fasta: bool = False):
        super().__init__(daemon=True, name=f"cutseq-gpu{device}")
        self.tp, self.device, self.done, self.chunk_reads, self.compress = tp, device, done, chunk_reads, compress
        self.fasta = fasta  # the records leave as FASTA
        self.bins = bins  # demultiplexing: one route per barcode behind the three ordinary ones
        self.inbox: "queue.Queue" = queue.Queue(maxsize=self.SLOTS)
        self.engine = self.text = None
        self.stride, self.capacity, self.want_stride = 152, 0, 152
        self.stats = None
        self.error: Optional[BaseException] = None
        self.submitted = 0

    def _collect_stats(self):
        part = [s.as_dict() for s in self.engine.stats()]
        self.stats = part if self.stats is None else [shard.merge_stats([a, b]) for a, b in zip(self.stats, part)]

    def _ensure(self, inflight: deque, text_bytes: int, stride: Optional[int] = None):
        """The text engine, rebuilt for longer rows or bigger blocks -- after everything in flight came back."""
        stride = max(stride or self.stride, self.want_stride)
        if self.text is not None and stride <= self.stride and text_bytes <= self.capacity:
            return
        from .engine import TrimEngine
        while inflight:
            self._finish(inflight)
        if self.text is not None:
            self.text.close()
        if self.engine is None:
            self.engine = TrimEngine(self.tp, device=self.device, slots=0)
        self.stride = max(self.stride, stride)
        self.capacity = max(self.capacity, int(text_bytes * 1.25) + (1 << 20))
        self.text = textpath.TextEngine(self.engine, slots=self.SLOTS, max_text_bytes=self.capacity,
                                        max_records=self.chunk_reads, stride=self.stride, compress=self.compress,
                                        bins=self.bins, fasta=self.fasta)
        self.submitted = 0

    def _submit(self, inflight: deque, k: int, b1: TextBlock, b2: Optional[TextBlock]):
        slot = self.submitted % self.SLOTS
        self.submitted += 1
        if os.environ.get("CUTSEQ_DEBUG_BLOCKS") == "1":  # diagnostic: what goes to the device
            import sys
            L = _host()
            print(f"block {k}: slot {slot} records {b1.n}/{b2.n if b2 is not None else '-'} bytes {b1.nbytes}/"
                  f"{b2.nbytes if b2 is not None else '-'} newlines {L.csh_count_newlines(b1.buf.ctypes.data, b1.nbytes)}/"
                  f"{L.csh_count_newlines(b2.buf.ctypes.data, b2.nbytes) if b2 is not None else '-'} capacity {self.capacity}",
                  file=sys.stderr, flush=True)
        self.text.submit(slot, b1.buf, b1.nbytes, b2.buf if b2 is not None else None, b2.nbytes if b2 is not None else 0, b1.n)
        inflight.append((k, slot, b1, b2))

    def _finish(self, inflight: deque):
        k, slot, b1, b2 = inflight.popleft()
        t0 = time.perf_counter()
        try:
            res = self.text.wait(slot, first_record=b1.first_record)
            t0 = _tick("wait", t0)
        except textpath.ReadTooLong as exc:
            from .run import ReadTooLong
            raise ReadTooLong(f"{exc} in records {b1.first_record + 1}..{b1.first_record + b1.n}")
        except textpath.TextFormatError as exc:
            if exc.code == abi.CS_TEXT_ERR_IDS_DIFFER:
                raise ValueError(str(exc))
            raise fastq.FastqFormatError(str(exc))
        # Reads longer than the rows took the exact but slow long-read kernel.  A library whose reads are simply longer
        # than the rows so far (2 x 250, 2 x 300) should not stay there: longer rows from the next batch on.
        if max(res.n_long) * 64 > b1.n and self.stride < abi.CS_MAX_STRIDE:
            self.want_stride = max(self.want_stride, min(abi.CS_MAX_STRIDE, (int(res.max_len) + 3) // 4 * 4))
        sizes = counts = None
        if self.bins:
            sizes, _, counts = self.text.routes(slot)
        out = [fastq.PINNED.take(max(int(res.out_bytes[m]), 1)) for m in range(2 if b2 is not None else 1)]
        t0 = _tick("take", t0)
        self.text.fetch(slot, out[0], out[1] if b2 is not None else None)
        _tick("fetch", t0)
        b1.release()
        if b2 is not None:
            b2.release()
        self.done.put(_Done(k, b1.n, res, out, sizes, counts))

    def run(self):
        inflight: deque = deque()
        try:
            # the trimming engine right away (plan upload, code object load: 0.1-0.2 s), while the readers are still
            # getting their first blocks; the text engine follows when the first block says how big it has to be
            t0 = time.perf_counter()
            from .engine import TrimEngine
            self.engine = TrimEngine(self.tp, device=self.device, slots=0)
            _tick("ensure", t0)
            while True:
                t0 = time.perf_counter()
                item = self.inbox.get()
                t0 = _tick("idle", t0)
                if item is None:
                    break
                k, b1, b2 = item
                self._ensure(inflight, max(b1.nbytes, b2.nbytes if b2 is not None else 0))
                t0 = _tick("ensure", t0)
                if len(inflight) == self.SLOTS:
                    self._finish(inflight)
                t0 = time.perf_counter()
                self._submit(inflight, k, b1, b2)
                _tick("submit", t0)
            while inflight:
                self._finish(inflight)
            if self.engine is not None:
                self._collect_stats()
        except BaseException as exc:
            self.error = exc
        finally:
            try:
                if self.text is not None:
                    self.text.close()
                if self.engine is not None:
                    self.engine.close()
            finally:
                self.done.put(self)


_FASTA_EXT = (".fasta", ".fa", ".fna", ".csfasta", ".csfa")
_FASTQ_EXT = (".fastq", ".fq")


def format_of_name(name: str) -> Optional[str]:
    """dnaio's rule for output files: the format goes by the extension in front of a compression suffix; None when
    the name says nothing (the input's format then decides)."""
    low = name.lower()
    for ext in (".gz", ".bz2", ".xz", ".zst"):
        if low.endswith(ext):
            low = low[: -len(ext)]
            break
    if low.endswith(_FASTA_EXT):
        return "fasta"
    if low.endswith(_FASTQ_EXT):
        return "fastq"
    return None


def output_format(names, has_qualities: bool) -> bool:
    """-> True when the records leave as FASTA.  What OutputFiles(qualities=runner.input_file_format().has_qualities())
    does in the reference (cutseq/run.py:437-441, 754-758): a name with a FASTA / FASTQ extension fixes the format,
    any other name follows the input; FASTQ output of an input without qualities is an error (dnaio refuses it)."""
    kinds = {format_of_name(n) or ("fastq" if has_qualities else "fasta") for n in names}
    if "fastq" in kinds and not has_qualities:
        raise fastq.FastqFormatError(
            "Output format cannot be FASTQ since no quality values are available: the input is FASTA "
            "(name the output files .fasta / .fa)")
    if len(kinds) > 1:
        raise ValueError("the output files name different formats (FASTA and FASTQ): one format per run")
    return kinds == {"fasta"}


def run_text_pipeline(args, tp, devices, chunk_reads: int, shares=None) -> dict:
    """The CLI's run on the text path -> the run statistics ``report`` expects.  ``shares``: per input file the part
    of it this process takes (``ranks.py``)."""
    global _DISCARD
    _DISCARD = os.environ.get("CUTSEQ_DISCARD_OUTPUT") == "1"  # diagnostic: the writers drop their bytes
    paired = tp.paired
    in1 = args.input_file[0]
    in2 = args.input_file[1] if paired else None
    share = shares or [{}, {}]
    r1 = TextReader(in1, chunk_reads, **share[0])  # (a missing input file raises here, before anything else exists)
    r2 = None
    opened: List[StreamWriter] = []
    try:
        r2 = TextReader(in2, chunk_reads, **share[1]) if paired else None
        if PROFILE:
            from .run import _phase
            _phase("readers started")

        # every output a ".gz" file: the device compresses (deflate_kernels.hip.inc) and the writers pass the members
        # through; otherwise text comes back and ".gz" outputs are deflated in the host pool (CUTSEQ_GPU_DEFLATE=0 too)
        # demultiplexing (table form): the trimmed pairs of barcode b are route 3 + b, one pair of files each
        n_bins = len(tp.demux.barcodes) if tp.demux is not None else 0
        bin_files = list(getattr(args, "demux_files", None) or []) if n_bins else []
        names_all = [n for group in [args.output_file if not n_bins else [], args.short_file, args.untrimmed_file] + bin_files
                     for n in group if n]
        compress = bool(names_all) and all(n != "-" and codec.container_of_name(n) == "gzip" for n in names_all) and os.environ.get("CUTSEQ_GPU_DEFLATE", "1") != "0"
        if r2 is not None and r1.fasta != r2.fasta:
            raise fastq.FastqFormatError("the two input files are in different formats (one FASTA, one FASTQ)")
        fasta_out = output_format(names_all, has_qualities=not r1.fasta)

        def mk(names):
            group = []
            for n in names:
                group.append(StreamWriter(n, precompressed=compress) if n else None)
                if group[-1] is not None:
                    opened.append(group[-1])
            return group

        trimmed = mk(args.output_file) if not n_bins else [None] * len(args.output_file)
        if paired and tp.swap_outputs:
            trimmed = trimmed[::-1]
        outs = [trimmed, mk(args.short_file), mk(args.untrimmed_file)]
        for names in bin_files:
            files = mk(names)
            outs.append(files[::-1] if paired and tp.swap_outputs else files)
    except BaseException:
        r1.close()
        if r2 is not None:
            r2.close()
        for fh in opened:
            try:
                fh.close()
            except BaseException:
                pass
        raise
    totals = report.new_totals()
    t_start = time.perf_counter()
    if PROFILE:
        from .run import _phase
        _phase("readers and writers open")
    done: "queue.Queue" = queue.Queue()
    workers = [TextWorker(tp, dev, done, chunk_reads, compress, n_bins, fasta_out) for dev in devices]
    if n_bins:
        totals["routes"] += [0] * n_bins
    budget = threading.Semaphore(2 * len(workers) * TextWorker.SLOTS + 2)  # batches between reader and disk
    failure: List[BaseException] = []

    progress = report.Progress()

    def emit(item: _Done):
        res = item.res
        totals["in_pairs"] += item.n
        progress.update(item.n)
        if item.counts is not None:  # (demultiplexing: [0] stays 0, the barcodes' pairs are routes 3 ..)
            for q in range(len(item.counts)):
                totals["routes"][q] += int(item.counts[q])
        else:
            for q in range(3):
                totals["routes"][q] += int(res.route_count[q])
        for m in range(2 if paired else 1):
            totals["written_bp"][m] += int(res.written_bp[m])
        sizes = item.sizes if item.sizes is not None else res.route_bytes
        jobs = []
        for m in range(2 if paired else 1):
            at = 0
            for route in range(len(outs)):
                nbytes = int(sizes[route][m])
                fh = (outs[route] + [None])[m]
                if nbytes and fh is not None:
                    jobs.append((fh, memoryview(item.out[m])[at:at + nbytes]))
                at += nbytes
        shared = _Shared(item.out, len(jobs), budget.release)
        for fh, view in jobs:
            fh.put(view, shared)

    def collect():
        waiting, next_k, alive = {}, 0, len(workers)
        try:
            while alive or waiting:
                item = done.get()
                if isinstance(item, TextWorker):
                    alive -= 1
                    if item.error is not None:
                        raise item.error
                    if not alive and waiting and next_k not in waiting:
                        raise RuntimeError("a batch went missing between the GPU workers and the writers")
                    continue
                waiting[item.k] = item
                while next_k in waiting:
                    emit(waiting.pop(next_k))
                    next_k += 1
        except BaseException as exc:
            failure.append(exc)
            while alive:  # keep the workers from blocking on a dead consumer
                item = done.get()
                if isinstance(item, TextWorker):
                    alive -= 1
                else:
                    for b in item.out:
                        fastq.PINNED.give(b)
                    budget.release()

    collector = threading.Thread(target=collect, daemon=True, name="cutseq-collect")
    for w in workers:
        w.start()
    collector.start()
    first_error: Optional[BaseException] = None
    try:
        k = 0
        while not failure:
            t0 = time.perf_counter()
            b1 = r1.get()
            b2 = r2.get() if r2 is not None else None
            _tick("main_wait_readers", t0)
            if b1 is None and b2 is None:
                break
            if r2 is not None and (b1 is None or b2 is None or b1.n != b2.n):
                for b in (b1, b2):
                    if b is not None:
                        b.release()
                raise fastq.FastqFormatError(
                    "Reads are improperly paired! There are more reads in one file than in the other, "
                    "or a record is truncated.")
            while not budget.acquire(timeout=0.2):
                if failure:
                    break
            w = workers[k % len(workers)]
            while not failure:
                try:
                    w.inbox.put((k, b1, b2), timeout=0.2)
                    break
                except queue.Full:
                    continue
            k += 1
    except BaseException as exc:
        first_error = exc
    finally:
        r1.close()
        if r2 is not None:
            r2.close()
        for w in workers:
            while True:
                try:
                    w.inbox.put(None, timeout=0.2)
                    break
                except queue.Full:
                    if not w.is_alive():
                        break
        for w in workers:
            w.join()
        collector.join()
        for group in outs:
            for fh in group:
                if fh is None:
                    continue
                try:
                    fh.close()
                except BaseException as exc:
                    first_error = first_error or exc
    if first_error is None and failure:
        first_error = failure[0]
    if first_error is None:
        first_error = next((w.error for w in workers if w.error is not None), None)
    if first_error is not None:
        raise first_error
    # (the pinned arena stays: page-locking gigabytes costs more than a short run; it is freed when the process ends)
    stats = [w.stats for w in workers if w.stats is not None]
    for m in range(2 if paired else 1):
        totals["in_bp"][m] = sum(int(pair[m]["in_bp"]) for pair in stats)
        totals["out_bp"][m] = sum(int(pair[m]["out_bp"]) for pair in stats)
    totals["seconds"] = time.perf_counter() - t_start
    progress.close()
    totals["bin_names"] = list(args.demux[0]) if n_bins else None
    totals["stats"] = stats
    totals["devices"] = devices
    totals["path"] = "text"
    if PROFILE:
        import json
        import sys
        with _prof_lock:
            print(json.dumps({"cutseq_profile": {k: round(v, 3) for k, v in sorted(_prof.items())},
                              "seconds": round(totals["seconds"], 3)}), file=sys.stderr)
            _prof.clear()
    return totals
