"""``cutseq --ranks N``: one PROCESS per GPU for the command line as well.

The reference scales on the host with ``make_runner(inpaths, cores=N)`` (cutseq/run.py:436, 753): N worker
processes behind one reader.  Here the per-read work is on the GPUs and what has to scale with them is the byte
pushing on the host (read / inflate, upload, download, deflate / write), so every rank gets its own reader, its own
thread pool, its own GPU and its own output part:

  parent   no HIP call.  Splits the input at RECORD indices (rank r takes records [R r / N, R (r + 1) / N) of both
           mates): one pass that counts newlines -- plain text: parallel ``pread`` + count, exact byte offsets;
           gzip (multi-member / BGZF): members inflate in the pool, a rank starts at the member that holds its first
           line and skips the lines in front of it.  Spawns the children, concatenates their output parts in rank
           order (gzip members / plain text: the concatenation IS the file), merges the statistics, prints the report.
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


def _concatenate(parts: List[str], final: str) -> None:
    with open(final, "wb") as dst:
        for part in parts:
            with open(part, "rb") as src:
                size = os.fstat(src.fileno()).st_size
                at = 0
                while at < size:
                    try:
                        sent = os.sendfile(dst.fileno(), src.fileno(), at, size - at)
                    except OSError:
                        sent = 0
                    if sent <= 0:  # no in-kernel copy between these files: ordinary copy of the rest
                        src.seek(at)
                        while True:
                            block = src.read(8 << 20)
                            if not block:
                                break
                            dst.write(block)
                        break
                    at += sent
            os.unlink(part)


def run_parent(argv: List[str], args, tp) -> Optional[dict]:
    """Split, spawn, merge.  -> the merged run totals (what ``report`` takes), or None when the run has to fall back
    to one process."""
    world = int(args.ranks)
    paths = list(args.input_file)
    t0 = time.perf_counter()
    shares, info = split_inputs(paths, world)
    if shares is None:
        logging.warning(f"--ranks {world} ignored: {info}.")
        return None
    want = os.environ.get("CUTSEQ_DEVICES")
    devices = [int(x) for x in want.split(",")] if want else list(range(world))
    groups = {"output_file": args.output_file, "short_file": args.short_file, "untrimmed_file": args.untrimmed_file}
    for b, names in enumerate(getattr(args, "demux_files", None) or []):  # demultiplexing: a pair of files per barcode
        groups[f"demux_files:{b}"] = names
    if getattr(args, "demux_files", None):
        groups["output_file"] = [None] * len(args.output_file)  # (nothing is written under the common trimmed names)
    work = tempfile.mkdtemp(prefix="cutseq_ranks_")
    children, specs = [], []
    child_argv = _strip_option(argv, "--ranks")
    try:
        for r in range(world):
            spec = {
                "rank": r, "world": world, "inputs": shares[r],
                # (a part keeps the ".gz" ending: the writers pick their codec by the name)
                "outputs": {k: [(f"{name}.rank{r}.part{'.gz' if name.endswith('.gz') else ''}" if name else None) for name in names]
                            for k, names in groups.items()},
                "totals_file": os.path.join(work, f"totals{r}.json"),
            }
            path = os.path.join(work, f"spec{r}.json")
            with open(path, "w") as fh:
                json.dump(spec, fh)
            specs.append(spec)
            env = dict(os.environ, CUTSEQ_DEVICES=str(devices[r % len(devices)]), CUTSEQ_PROGRESS="0")  # (one Done line: the parent's)
            children.append(subprocess.Popen([sys.executable, "-m", "cutseq_amd.run"] + child_argv + ["--rank-spec", path], env=env))
        codes = [c.wait() for c in children]
        if any(codes):
            raise RuntimeError(f"rank {next(i for i, c in enumerate(codes) if c)} failed (exit code {max(codes)})")
        for key, names in groups.items():
            for i, name in enumerate(names):
                if name:
                    _concatenate([s["outputs"][key][i] for s in specs], name)
        totals = report.new_totals()
        stats, devs = [], []
        for s in specs:
            with open(s["totals_file"]) as fh:
                part = json.load(fh)
            report.merge_totals(totals, part)
            stats += part["stats"]
            devs += part["devices"]
        bin_names = list(args.demux[0]) if getattr(args, "demux", None) else None
        totals.update(stats=stats, devices=devs, bin_names=bin_names, seconds=time.perf_counter() - t0, ranks=world)
        progress = report.Progress()
        progress.t0 -= totals["seconds"]
        progress.update(totals["in_pairs"])
        progress.close()
        return totals
    except BaseException:
        for c in children:
            if c.poll() is None:
                c.kill()
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
