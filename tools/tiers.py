#!/usr/bin/env python3
"""Tier T (pinned host -> H2D -> kernels -> D2H) and tier E (FASTQ files in -> FASTQ files out through the product
CLI) rates, SURVEY.md 8d.  NOT the headline: bench.py reports the HBM-resident kernel rate and appends what
``run_all`` returns as ``tiers`` -- so the driver's own bench run witnesses them.

Tier E: a seeded synthetic FASTQ pair (the bench workload's reads, fixed-width names) is written to /dev/shm as plain
text, as multi-member gzip (what this CLI and bgzip write) and as ONE gzip member per file (what a sequencer
writes); ``python -m cutseq_amd.run`` runs on them in fresh child processes (interpreter, HIP and engine start-up
included: what a user of the command line sees).  Every run's decompressed output is checked: all runs must give the
same bytes (length + CRC-32 per file), and the head of the trimmed streams must equal the CPU oracle's results for the
first pairs, formatted by the record-logic specification (tests/hostfmt.py) -- output order is input order.

    python tools/tiers.py [pairs]
"""
from __future__ import annotations

import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from cutseq_amd import capi, plan as planmod, workloads  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig  # noqa: E402
from cutseq_amd.engine import TrimEngine  # noqa: E402

ID_DIGITS = 9
ORACLE_PAIRS = 100_000  # head of the run checked record by record against the oracle


def takarav3_plan():
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    return planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)


def host_threads() -> int:
    return max(1, min(16, len(os.sched_getaffinity(0))))


# ---------------------------------------------------------------- tier T

def pinned_like(L, arr):
    p = L.cs_alloc_pinned(arr.nbytes)
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(arr.nbytes,)).view(arr.dtype).reshape(arr.shape)
    out[...] = arr
    return out


def tier_t(batch, n=1 << 20, rounds=8):
    """cs_trim_batch with two slots in flight: the copies of one overlap the kernels of the other."""
    tp = takarav3_plan()
    L = capi.load()
    arrs = [pinned_like(L, a[:n]) for a in (batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2)]
    with TrimEngine(tp, device=0, slots=2, max_reads=n, max_stride=batch.stride) as eng:
        for s in (0, 1):
            eng.submit(s, *arrs)
            eng.wait(s)
        t0 = time.perf_counter()
        for r in range(rounds):
            eng.submit(r & 1, *arrs)
            if r:
                eng.wait((r - 1) & 1)
        eng.wait((rounds - 1) & 1)
        dt = time.perf_counter() - t0
    for a in arrs:
        L.cs_free_pinned(a.ctypes.data)
    return {"pairs": n * rounds, "seconds": round(dt, 4), "M_pairs_per_s": round(n * rounds / dt / 1e6, 2),
            "host_GBps": round(n * rounds * (4 * batch.stride + 4 + 16) / dt / 1e9, 2),
            "what": "cs_trim_batch: pinned host arrays -> H2D -> scan + resolve kernels -> D2H of the 8-byte results, two slots in flight"}


# ---------------------------------------------------------------- synthetic FASTQ text

def fastq_text(seq: np.ndarray, qual: np.ndarray, lens: np.ndarray, mate: int, first: int = 0) -> np.ndarray:
    """[n, record_bytes] uint8: ``@SIM:<9 digits> <mate>:N:0:IDX\\n<seq>\\n+\\n<qual>\\n`` -- every read of the bench
    workload is 150 bases, so records have one size and the text is assembled by block copies."""
    n, L = seq.shape[0], int(lens[0])
    assert (lens == L).all(), "fixed-length reads expected"
    head = f"@SIM:{'0' * ID_DIGITS} {mate}:N:0:IDX\n".encode()
    rec = np.empty((n, len(head) + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :len(head)] = np.frombuffer(head, dtype=np.uint8)
    idx = np.arange(first, first + n, dtype=np.int64)
    for d in range(ID_DIGITS):
        rec[:, 5 + ID_DIGITS - 1 - d] = (idx % 10 + 48).astype(np.uint8)
        idx //= 10
    o = len(head)
    rec[:, o:o + L] = seq[:, :L]
    rec[:, o + L:o + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, o + L + 3:o + 2 * L + 3] = qual[:, :L]
    rec[:, -1] = 10
    return rec


def names_of(n: int, mate: int, first: int = 0):
    return [f"SIM:{first + i:0{ID_DIGITS}d} {mate}:N:0:IDX".encode() for i in range(n)]


def write_inputs(work: Path, batch, n: int, pool: ThreadPoolExecutor, piece_records: int = 262_144):
    """plain_R{1,2}.fastq, multi_R{1,2}.fastq.gz (one member per 65 536 records), single_R{1,2}.fastq.gz (ONE member:
    raw-deflate pieces of 262 144 records joined by sync flushes, the way pigz builds one; level 1, zlib's deflate_fast
    like gzip -1).  A job of the pool is 262 144 records of one mate: their text, their four members, their raw piece
    (numpy copies and zlib both run outside the interpreter lock) -- assembling 2.6 GB of text on one thread and
    compressing one 1.3 GB piece per mate used to be half of bench.py's wall time."""
    MEMBER, PIECE = 65_536, piece_records

    def job(args):
        mate, lo, hi, last = args
        seq, qual, lens = (batch.seq1, batch.qual1, batch.len1) if mate == 1 else (batch.seq2, batch.qual2, batch.len2)
        text = fastq_text(seq[lo:hi], qual[lo:hi], lens[lo:hi], mate, first=lo)
        rec_bytes = text.shape[1]
        flat = memoryview(text.reshape(-1))
        members = []
        for at in range(0, hi - lo, MEMBER):
            c = zlib.compressobj(1, zlib.DEFLATED, 31)
            members.append(c.compress(flat[at * rec_bytes:(at + MEMBER) * rec_bytes]) + c.flush())
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        raw = c.compress(flat) + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)
        return flat, members, raw, zlib.crc32(flat)

    sizes = {}
    spans = [(lo, min(n, lo + PIECE)) for lo in range(0, n, PIECE)]
    futs = {mate: [pool.submit(job, (mate, lo, hi, hi == n)) for lo, hi in spans] for mate in (1, 2)}
    for mate in (1, 2):
        crc = total = 0
        with open(work / f"plain_R{mate}.fastq", "wb") as f_plain, open(work / f"multi_R{mate}.fastq.gz", "wb") as f_multi, \
                open(work / f"single_R{mate}.fastq.gz", "wb") as f_single:
            f_single.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\xff")
            for k, fut in enumerate(futs[mate]):
                flat, members, raw, piece_crc = fut.result()
                futs[mate][k] = None
                f_plain.write(flat)
                for blob in members:
                    f_multi.write(blob)
                f_single.write(raw)
                crc = _crc32_combine(crc, piece_crc, len(flat))
                total += len(flat)
            f_single.write(int(crc).to_bytes(4, "little") + int(total & 0xffffffff).to_bytes(4, "little"))
        sizes[f"R{mate}"] = {"plain": total, "multi_gz": os.path.getsize(work / f"multi_R{mate}.fastq.gz"),
                             "single_gz": os.path.getsize(work / f"single_R{mate}.fastq.gz")}
    return sizes


def _crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """CRC-32 of A + B from crc(A), crc(B), len(B): crc(A) advanced over len2 zero bytes (multiplication by
    x^(8 len2) mod the CRC polynomial, bit-reflected, square-and-multiply), XOR crc(B)."""
    def times(a, b):  # a * b mod P over GF(2), reflected representation
        p = 0
        while b:
            if b & 0x80000000:
                p ^= a
            a = (a >> 1) ^ (0xEDB88320 if a & 1 else 0)
            b = (b << 1) & 0xFFFFFFFF
        return p
    power, x = 0x80000000, 0x00800000  # 1 and x^8
    k = len2
    while k:
        if k & 1:
            power = times(power, x)
        x = times(x, x)
        k >>= 1
    return times(crc1, power) ^ crc2


# ---------------------------------------------------------------- tier E

def _stream_sig(path: Path):
    """((length, CRC-32) of a file's decompressed bytes, its first 64 MiB) -- zlib, member by member (the checker must
    not be the product's own decoder)."""
    head_cap = 64 << 20
    if not str(path).endswith(".gz"):
        text = path.read_bytes()
        return (len(text), zlib.crc32(text)), text[:head_cap]
    total, crc, head = 0, 0, []
    head_len = 0
    with open(path, "rb") as fh:
        d = zlib.decompressobj(31)
        buf = b""
        while True:
            if not buf:
                buf = fh.read(8 << 20)
                if not buf:
                    break
            piece = d.decompress(buf)
            buf = d.unused_data if d.eof else b""
            if piece:
                total += len(piece)
                crc = zlib.crc32(piece, crc)
                if head_len < head_cap:
                    head.append(piece[: head_cap - head_len])
                    head_len += len(head[-1])
            if d.eof:
                d = zlib.decompressobj(31)
    return (total, crc), b"".join(head)


def oracle_head(batch, tp, n_head: int):
    """trimmed R1 / R2 streams of the first ``n_head`` pairs: C oracle results + tests/hostfmt.py record logic."""
    import oracle
    sys.path.insert(0, str(ROOT / "tests"))
    import hostfmt

    a1, n1, a2, n2 = tp.pack()
    params = tp.params()
    th = oracle.host_threads()
    r1, _, _ = oracle.trim_mate(a1, n1, params, batch.seq1[:n_head], batch.qual1[:n_head], batch.len1[:n_head], threads=th)
    r2, _, _ = oracle.trim_mate(a2, n2, params, batch.seq2[:n_head], batch.qual2[:n_head], batch.len2[:n_head], threads=th)
    nm1, nm2 = names_of(n_head, 1), names_of(n_head, 2)
    out1, out2 = [], []
    for i in range(n_head):
        l1, l2 = int(batch.len1[i]), int(batch.len2[i])
        route, rec1, rec2 = hostfmt.format_pair(nm1[i], batch.seq1[i, :l1].tobytes(), batch.qual1[i, :l1].tobytes(), r1[i],
                                                nm2[i], batch.seq2[i, :l2].tobytes(), batch.qual2[i, :l2].tobytes(), r2[i], tp)
        if route == 0:
            out1.append(rec1)
            out2.append(rec2)
    return b"".join(out1), b"".join(out2)


def tier_e(batch, n: int, work: Path):
    work.mkdir(parents=True, exist_ok=True)
    free = shutil.disk_usage(work).free
    need = n * 2 * 330 * 4
    if free < need:
        return {"skipped": f"{work}: {free >> 20} MiB free, {need >> 20} MiB needed for {n} pairs"}
    out = {"pairs": n, "workdir": str(work), "host_threads": host_threads(),
           "what": "cutseq_amd.run -A TAKARAV3 --trim-polyA on files on tmpfs, one fresh child process per form running the "
                   "command line twice: M_pairs_per_s / seconds = the second run (warm: what a long run sustains), "
                   "first_run_* = the first (interpreter, HIP and engine start-up, page-locking, cold output pages)"}
    tp = takarav3_plan()
    pool = ThreadPoolExecutor(host_threads())
    try:
        t0 = time.perf_counter()
        out["input_bytes"] = write_inputs(work, batch, n, pool)
        out["make_inputs_s"] = round(time.perf_counter() - t0, 2)
        t0 = time.perf_counter()
        want1, want2 = oracle_head(batch, tp, min(ORACLE_PAIRS, n))
        out["oracle_head_s"] = round(time.perf_counter() - t0, 2)
        forms = {
            "E_plain": (["plain_R1.fastq", "plain_R2.fastq"], ".fastq"),
            "E_plain_to_gz": (["plain_R1.fastq", "plain_R2.fastq"], ".fastq.gz"),
            "E_gz": (["multi_R1.fastq.gz", "multi_R2.fastq.gz"], ".fastq.gz"),
            "E_gz_single": (["single_R1.fastq.gz", "single_R2.fastq.gz"], ".fastq.gz"),
        }
        sigs = {}
        env = dict(os.environ, PYTHONPATH=str(ROOT))
        for tag, (inputs, ext) in forms.items():
            outs = [str(work / f"{tag}_{k}{ext}") for k in ("o1", "o2", "s1", "s2")]
            args = ["-A", "TAKARAV3", "--trim-polyA", str(work / inputs[0]), str(work / inputs[1]),
                    "-o", outs[0], outs[1], "-s", outs[2], outs[3]]
            # a fresh process runs the command line twice: the first run is what a user sees (interpreter, HIP, engines,
            # page-locked buffers, cold page cache of the outputs), the second the steady rate of a long run
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--child", json.dumps(args)], cwd=str(ROOT),
                               env=env, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            line = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if r.returncode != 0 or not line:
                out[tag] = {"error": (r.stderr or r.stdout)[-400:]}
                continue
            child = json.loads(line[-1])
            got = list(pool.map(_stream_sig, [Path(p) for p in outs]))
            sigs[tag] = [g[0] for g in got]
            head_ok = got[0][1][:len(want1)] == want1[:len(got[0][1])] and got[1][1][:len(want2)] == want2[:len(got[1][1])]
            head_ok = head_ok and len(got[0][1]) >= min(len(want1), 64 << 20)
            out[tag] = {"M_pairs_per_s": round(n / child["second_s"] / 1e6, 2), "seconds": round(child["second_s"], 3),
                        "first_run_seconds": round(child["first_s"], 3),
                        "first_run_M_pairs_per_s": round(n / child["first_s"] / 1e6, 2),
                        "process_seconds": round(dt, 3),
                        "output_bytes": sum(os.path.getsize(p) for p in outs),
                        "decompressed_crc32": [f"{g[0][1]:08x}" for g in got],
                        "head_equals_oracle": bool(head_ok)}
            for p in outs:
                os.unlink(p)
        first = next(iter(sigs.values()), None)
        out["all_runs_same_output"] = bool(sigs) and all(v == first for v in sigs.values()) and len(sigs) == len(forms)
        out["oracle_check"] = (f"trimmed R1/R2 streams start with the oracle's records for the first {min(ORACLE_PAIRS, n)} pairs "
                               "(tests/hostfmt.py formatting); all forms give the same decompressed bytes")
    finally:
        pool.shutdown()
        shutil.rmtree(work, ignore_errors=True)
    return out


def run_all(batch, n_pairs: int = 4_000_000, work: str = "/dev/shm/cutseq_bench_tiers"):
    """-> {"T": ..., "E_plain": ..., "E_plain_to_gz": ..., "E_gz": ..., "E_gz_single": ..., ...}; never raises."""
    res = {}
    try:
        res["T"] = tier_t(batch, n=min(1 << 20, batch.seq1.shape[0]))
    except Exception as exc:  # noqa: BLE001 -- an extra leg must not take the bench line down
        res["T"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:
        e = tier_e(batch, min(n_pairs, batch.seq1.shape[0]), Path(work))
        for key in ("E_plain", "E_plain_to_gz", "E_gz", "E_gz_single"):
            if key in e:
                res[key] = e.pop(key)
        res["E_setup"] = e
    except Exception as exc:  # noqa: BLE001
        res["E_setup"] = {"error": f"{type(exc).__name__}: {exc}"}
    return res


def child_main(argv_json: str) -> None:
    """--child: the command line twice in this process, wall time of each on stdout."""
    from cutseq_amd import run as cli
    args = json.loads(argv_json)
    times = []
    outs = [a for a in args if "/E_" in a]  # (this module's own output names)
    for k in range(2):
        if k:  # a run writes NEW files: freeing the first run's gigabytes of tmpfs pages is not its job
            for path in outs:
                if os.path.exists(path):
                    os.unlink(path)
        t0 = time.perf_counter()
        try:
            cli.main(list(args))
        except SystemExit as exc:
            if exc.code:
                raise
        times.append(time.perf_counter() - t0)
    print(json.dumps({"first_s": times[0], "second_s": times[1]}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child_main(sys.argv[2])
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    print(json.dumps(run_all(workloads.make_batch("config3", n), n)))
