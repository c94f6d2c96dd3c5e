"""``cutseq --ranks N``: one PROCESS per GPU for the command line as well.

The reference scales on the host with ``make_runner(inpaths, cores=N)`` (cutseq/run.py:436, 753): N worker
processes behind one reader.  Here the per-read work is on the GPUs and what has to scale with them is the byte
pushing on the host (read / inflate, upload, download, deflate / write), so every rank gets its own reader, its own
thread pool, its own GPU and its own output part:

  parent   no HIP call.  Splits the input so that rank r takes the same records of both mates:
             * multi-member gzip files whose members hold the same records in both mates (what this CLI, and anything
               else that writes R1 and R2 in lockstep, produces): by MEMBER INDEX, from a scan for member starts --
               nothing is inflated twice.  The assumption is checked, not trusted: the first record ids at every split
               point in the parent, every pair's ids and the record counts of the two mates in the ranks (the checks
               every run makes); a run that fails them is started again with the exact split below.
             * everything else: at RECORD indices from one counting pass -- plain text: parallel ``pread`` + newline
               count, exact byte offsets; gzip: the members inflate in the pool (a second time in the ranks: the price
               of members that do not line up), a rank starts at the member that holds its first line and skips the
               lines in front of it.
           Spawns the children with their share of the host threads, puts their output parts behind rank 0's files
           (pre-sized, parallel in-kernel copies: gzip members / plain text -- the concatenation IS the file), merges
           the statistics, prints the report.
  child    ``python -m cutseq_amd.run ... --rank-spec FILE``: the ordinary text path on its share, on its GPU.

Output is byte-identical (decompressed) to the one-process run; no data crosses between ranks.
A gzip input that is one single member cannot be entered in the middle: such a run falls back to one rank.
"""
from __future__ import annotations

import json
import logging
import os
import subprocess
import sys
import tempfile
import time
from typing import List, Optional

import numpy as np

from . import codec, fastq, report

_PIECE = 8 << 20


def _lines_plain(path: str):
    """-> (piece offsets, newlines per piece, file size, ends with newline) by parallel pread + count."""
    from .textio import _host
    L = _host()
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    pool = fastq._pool()
    tls = {}

    def job(off):
        import threading
        buf = tls.get(threading.get_ident())
        if buf is None:
            buf = tls[threading.get_ident()] = np.empty(_PIECE, dtype=np.uint8)
        got = os.preadv(fd, [memoryview(buf)], off)
        return int(L.csh_count_newlines(buf.ctypes.data, got))

    try:
        offs = list(range(0, size, _PIECE))
        counts = list(pool.map(job, offs))
        last = os.pread(fd, 1, size - 1) if size else b"\n"
    finally:
        os.close(fd)
    return offs, counts, size, last == b"\n"


def _line_start_plain(path: str, offs, counts, line: int) -> int:
    """Byte offset at which line number ``line`` (0-based) starts."""
    if line == 0:
        return 0
    from .textio import _host
    L = _host()
    cum = 0
    for off, c in zip(offs, counts):
        if cum + c >= line:
            with open(path, "rb") as fh:
                fh.seek(off)
                piece = np.frombuffer(fh.read(_PIECE), dtype=np.uint8)
            return off + int(L.csh_after_kth_newline(piece.ctypes.data, piece.size, line - cum))
        cum += c
    raise ValueError("line beyond the end of the file")


def _lines_gzip(path: str):
    """-> (member start offsets, newlines per block, ends with newline); an offset of -1 marks a stream that cannot be
    entered there."""
    from .textio import _host
    L = _host()
    src = codec.GzipSource(path, fastq._pool(), fastq.ARENA.take, fastq.ARENA.give)
    offs, counts, last = [], [], True
    try:
        for off, arr, nbytes in src.indexed_blocks():
            offs.append(off)
            counts.append(int(L.csh_count_newlines(arr.ctypes.data, nbytes)))
            if nbytes:
                last = bool(arr[nbytes - 1] == 10)
            if isinstance(arr.base, fastq.mmap.mmap):
                fastq.ARENA.give(arr)
    finally:
        src.close()
    return offs, counts, last


def gzip_members(path: str) -> Optional[List[int]]:
    """Start offsets of the members of a gzip file WITHOUT inflating it: every ``1f 8b 08`` with clean flag bits is a
    candidate (csh_find_gzip_magic, a memchr walk), a candidate counts when the first kilobytes behind it inflate
    without an error -- the magic turns up in compressed data by chance a few times per gigabyte, and such a place is
    not the start of a deflate stream.  None: the file is not a gzip file / has one member only."""
    import mmap
    import zlib
    size = os.path.getsize(path)
    if size < 18:
        return None
    with open(path, "rb") as fh:
        buf = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    try:
        if buf[:2] != b"\x1f\x8b":
            return None
        if codec._bgzf_block_size(buf, 0):
            return None  # BGZF: 64 KB blocks cut anywhere in the text, and the reader walks them in spans of its own
        H = codec._pinflate()
        base = codec._view_address(memoryview(buf)) if H is not None else 0
        out, pos = [0], 1
        while pos < size - 18:
            p = H.csh_find_gzip_magic(base, pos, size - 2) if H is not None else buf.find(b"\x1f\x8b\x08", pos)
            if p < 0:
                break
            pos = p + 1
            if (buf[p + 3] & 0xE0) != 0:
                continue
            try:
                head = zlib.decompressobj(31).decompress(buf[p:min(size, p + 65536)], 4096)
            except zlib.error:
                continue
            if head:
                out.append(p)
        return out if len(out) > 1 else None
    finally:
        try:
            buf.close()
        except BufferError:
            pass


def _first_id(path: str, offset: int) -> Optional[bytes]:
    """The id of the record a member starts with (up to the first blank, one trailing 1 / 2 / 3 dropped: dnaio's
    record_names_match); None when the member does not start with a record."""
    import zlib
    with open(path, "rb") as fh:
        fh.seek(offset)
        raw = fh.read(1 << 16)
    try:
        text = zlib.decompressobj(31).decompress(raw, 1 << 14)
    except zlib.error:
        return None
    lines = text.split(b"\n", 4)
    line = lines[0]
    if not line.startswith(b"@"):
        return None
    # a quality line may start with '@' as well: where the member's first four lines are all there they must look like a
    # record -- '+' line third, sequence and quality lines of one length (with one file there is no mate's id to compare)
    if len(lines) == 5 and not (lines[2].startswith(b"+") and len(lines[1].rstrip(b"\r")) == len(lines[3].rstrip(b"\r"))):
        return None
    name = line[1:].split(b" ", 1)[0].split(b"\t", 1)[0]
    return name[:-1] if name[-1:] in (b"1", b"2", b"3") else name


def split_by_members(paths: List[str], world: int):
    """-> (shares[rank][file] = dict(start, stop), None) or (None, reason): ranks take member index ranges -- valid when
    member k holds the same records in every file, which is checked at every split point (and record for record in
    the ranks)."""
    index = []
    for path in paths:
        if not codec.is_gzip(path):
            return None, f"{path} is not a gzip file"
        members = gzip_members(path)
        if members is None:
            return None, f"{path} is one gzip member"
        index.append(members)
    n = len(index[0])
    if any(len(m) != n for m in index):
        return None, "the input files have different numbers of gzip members"
    if n < world:
        return None, "fewer gzip members than ranks"
    cuts = [n * r // world for r in range(world + 1)]
    for k in cuts[1:-1]:
        ids = [_first_id(path, members[k]) for path, members in zip(paths, index)]
        if any(i is None for i in ids) or len(set(ids)) != 1:
            return None, "the members of the input files do not hold the same records"
    shares = []
    for r in range(world):
        row = []
        for path, members in zip(paths, index):
            stop = members[cuts[r + 1]] if cuts[r + 1] < n else None
            row.append({"start": int(members[cuts[r]]), "stop": stop})
        shares.append(row)
    return shares, None


def split_inputs(paths: List[str], world: int):
    """-> (shares[rank][file] = dict(start, skip_lines, max_records), records) or (None, reason)."""
    tables, totals = [], []
    for path in paths:
        if codec.is_gzip(path):
            offs, counts, ends_nl = _lines_gzip(path)
            if any(o < 0 for o in offs) or len(offs) < 2:
                return None, f"{path} is one gzip member: it cannot be entered in the middle"
            tables.append(("gz", offs, counts))
        else:
            offs, counts, _size, ends_nl = _lines_plain(path)
            tables.append(("plain", offs, counts))
        lines = sum(counts) + (0 if ends_nl else 1)
        totals.append(lines // 4)
    records = min(totals)
    if len(set(totals)) != 1:
        return None, "the input files hold different numbers of records"
    if records < world:
        return None, "fewer records than ranks"
    shares = []
    for r in range(world):
        lo, hi = records * r // world, records * (r + 1) // world
        row = []
        for path, (kind, offs, counts) in zip(paths, tables):
            line = 4 * lo
            if kind == "plain":
                start, skip = _line_start_plain(path, offs, counts, line), 0
            elif line == 0:
                start, skip = offs[0], 0
            else:
                cum, b = 0, 0
                # the block in which line number `line` STARTS: the one that holds newline number `line`
                for i, c in enumerate(counts):
                    if cum + c >= line:
                        b = i
                        break
                    cum += c
                start, skip = offs[b], line - cum
            # the last rank reads to the end of the file (and applies the reader's own end-of-file checks)
            row.append({"start": int(start), "skip_lines": int(skip), "max_records": None if r == world - 1 else int(hi - lo)})
        shares.append(row)
    return shares, records


def _append_parts(final: str, parts: List[str]) -> None:
    """``final`` (rank 0's file) + the other ranks' parts, in rank order: the file is extended to its final size once and
    every part is copied to its place with in-kernel copies, all parts at once (gzip members / plain text: the
    concatenation IS the file)."""
    sizes = [os.path.getsize(p) for p in parts]
    if not parts:
        return
    from concurrent.futures import ThreadPoolExecutor

    fd = os.open(final, os.O_RDWR)
    try:
        base = os.fstat(fd).st_size
        os.ftruncate(fd, base + sum(sizes))
        offsets = [base + sum(sizes[:i]) for i in range(len(parts))]

        def copy(job):
            part, at, size = job
            src = os.open(part, os.O_RDONLY)
            try:
                done = 0
                while done < size:
                    try:
                        n = os.copy_file_range(src, fd, size - done, done, at + done)
                    except (OSError, AttributeError):
                        n = 0
                    if n <= 0:  # no in-kernel copy between these files: an ordinary one for the rest
                        while done < size:
                            block = os.pread(src, min(8 << 20, size - done), done)
                            if not block:
                                raise OSError(f"{part}: shorter than its size")
                            os.pwrite(fd, block, at + done)
                            done += len(block)
                        break
                    done += n
            finally:
                os.close(src)
            os.unlink(part)

        with ThreadPoolExecutor(max(1, min(8, len(parts)))) as pool:
            list(pool.map(copy, zip(parts, offsets, sizes)))
    finally:
        os.close(fd)


def _host_threads(args) -> int:
    return int(args.threads) if getattr(args, "threads", None) else fastq.pool_size()


def run_parent(argv: List[str], args, tp) -> Optional[dict]:
    """Split, spawn, merge.  -> the merged run totals (what ``report`` takes), or None when the run has to fall back
    to one process."""
    world = int(args.ranks)
    paths = list(args.input_file)
    names_all = [n for names in (args.output_file, args.short_file, args.untrimmed_file) for n in names if n]
    if "-" in paths or "-" in names_all:
        logging.warning(f"--ranks {world} ignored: standard input / output cannot be shared between ranks.")
        return None
    for path in paths:
        container, first, _ = codec.sniff_input(path)
        if container not in ("plain", "gzip") or first in (b">", b"#"):
            logging.warning(f"--ranks {world} ignored: {path} is not a plain or gzip FASTQ file.")
            return None
    t0 = time.perf_counter()
    if getattr(args, "threads", None):
        try:
            fastq.set_threads(args.threads)  # (the counting pass below honours -t too)
        except RuntimeError:
            pass
    shares, why = split_by_members(paths, world)
    if shares is not None:
        try:
            return _run_ranks(argv, args, world, shares, t0, "gzip members (nothing inflated twice)")
        except RankFailure as exc:
            # only a rank that found its share's records out of step (exit code RANK_EXIT_RECORDS: FASTQ format / pairing
            # errors) says anything about the split; a full disk, a GPU error or a missing file would fail the same way
            # again, twice as late
            if exc.code != RANK_EXIT_RECORDS:
                raise
            logging.warning(f"--ranks {world}: the split by gzip members did not hold ({exc}); once more with the exact split.")
    shares, info = split_inputs(paths, world)
    if shares is None:
        logging.warning(f"--ranks {world} ignored: {info}.")
        return None
    return _run_ranks(argv, args, world, shares, t0, f"record indices from a counting pass ({why})")


RANK_EXIT_RECORDS = 3  # a rank's exit code for "my share's records do not line up" (run.run_cutseq)


class RankFailure(RuntimeError):
    def __init__(self, message: str, code: int = 1):
        super().__init__(message)
        self.code = code


def _run_ranks(argv: List[str], args, world: int, shares, t0: float, how: str) -> dict:
    want = os.environ.get("CUTSEQ_DEVICES")
    devices = [int(x) for x in want.split(",")] if want else list(range(world))
    groups = {"output_file": args.output_file, "short_file": args.short_file, "untrimmed_file": args.untrimmed_file}
    for b, names in enumerate(getattr(args, "demux_files", None) or []):  # demultiplexing: a pair of files per barcode
        groups[f"demux_files:{b}"] = names
    if getattr(args, "demux_files", None):
        groups["output_file"] = [None] * len(args.output_file)  # (nothing is written under the common trimmed names)
    work = tempfile.mkdtemp(prefix="cutseq_ranks_")
    children, specs = [], []
    # every rank gets its share of the host threads (N pools of all cores each would oversubscribe the host N times)
    per_rank = max(1, _host_threads(args) // world)
    child_argv = _strip_option(_strip_option(_strip_option(argv, "--ranks"), "--threads"), "-t") + ["-t", str(per_rank)]

    def part_name(name, r):  # rank 0 writes the final files themselves; a part keeps the container's ending
        if not name or r == 0:
            return name
        ending = next((e for e in (".gz", ".bz2", ".xz", ".zst") if name.endswith(e)), "")
        return f"{name}.rank{r}.part{ending}"

    try:
        for r in range(world):
            spec = {
                "rank": r, "world": world, "inputs": shares[r],
                "outputs": {k: [part_name(name, r) for name in names] for k, names in groups.items()},
                "totals_file": os.path.join(work, f"totals{r}.json"),
            }
            path = os.path.join(work, f"spec{r}.json")
            with open(path, "w") as fh:
                json.dump(spec, fh)
            specs.append(spec)
            env = dict(os.environ, CUTSEQ_DEVICES=str(devices[r % len(devices)]), CUTSEQ_PROGRESS="0")  # (one Done line: the parent's)
            children.append(subprocess.Popen([sys.executable, "-m", "cutseq_amd.run"] + child_argv + ["--rank-spec", path], env=env))
        left = list(range(world))
        while left:  # a rank that dies is noticed at once, whatever its number, and takes the others with it
            for r in list(left):
                code = children[r].poll()
                if code is None:
                    continue
                left.remove(r)
                if code != 0:
                    raise RankFailure(f"rank {r} failed (exit code {code})", code)
            if left:
                time.sleep(0.01)
        for key, names in groups.items():
            for i, name in enumerate(names):
                if name:
                    _append_parts(name, [s["outputs"][key][i] for s in specs[1:]])
        totals = report.new_totals()
        stats, devs = [], []
        for s in specs:
            with open(s["totals_file"]) as fh:
                part = json.load(fh)
            report.merge_totals(totals, part)
            stats += part["stats"]
            devs += part["devices"]
        bin_names = list(args.demux[0]) if getattr(args, "demux", None) else None
        totals.update(stats=stats, devices=devs, bin_names=bin_names, seconds=time.perf_counter() - t0, ranks=world,
                      ranks_split=how, threads_per_rank=per_rank)
        progress = report.Progress()
        progress.t0 -= totals["seconds"]
        progress.update(totals["in_pairs"])
        progress.close()
        return totals
    except BaseException:
        for c in children:
            if c.poll() is None:
                c.kill()
        for c in children:
            try:
                c.wait(timeout=30)
            except Exception:  # noqa: BLE001
                pass
        for s in specs:
            for names in s["outputs"].values():
                for name in names:
                    if name and os.path.exists(name):
                        os.unlink(name)
        raise
    finally:
        for name in os.listdir(work):
            os.unlink(os.path.join(work, name))
        os.rmdir(work)


def _strip_option(argv: List[str], name: str) -> List[str]:
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == name:
            skip = True
            continue
        if a.startswith(name + "="):
            continue
        out.append(a)
    return out


def load_spec(path: str) -> dict:
    with open(path) as fh:
        return json.load(fh)


def dump_totals(spec: dict, totals: dict) -> None:
    keep = {k: totals[k] for k in ("in_pairs", "routes", "in_bp", "out_bp", "written_bp", "stats", "devices")}
    with open(spec["totals_file"], "w") as fh:
        json.dump(keep, fh)
