"""Shared helpers for the parity tests (CPU oracle side + record plumbing)."""
from __future__ import annotations

import gzip
import random
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

import oracle
from oracle import pyref
from cutseq_amd import abi, plan as planmod
import hostfmt
from cutseq_amd.common import BarcodeConfig
from cutseq_amd.synth import SynthBatch

GOLDEN = Path(__file__).resolve().parent / "golden"


def pack_reads(reads: Sequence[Tuple[str, str]], stride: Optional[int] = None):
    """[(seq, qual)] -> (seq[n, stride] u8, qual[n, stride] u8, len[n] u16)."""
    n = len(reads)
    longest = max([len(s) for s, _ in reads] + [1])
    if stride is None:
        stride = (longest + 3) // 4 * 4
    seq = np.zeros((n, stride), dtype=np.uint8)
    qual = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, (s, q) in enumerate(reads):
        assert len(s) == len(q) <= stride
        seq[i, : len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
        qual[i, : len(q)] = np.frombuffer(q.encode(), dtype=np.uint8)
        lens[i] = len(s)
    return seq, qual, lens


def batch_from_reads(r1: Sequence[Tuple[str, str]], r2: Optional[Sequence[Tuple[str, str]]] = None) -> SynthBatch:
    longest = max([len(s) for s, _ in list(r1) + list(r2 or [])] + [1])
    stride = (longest + 3) // 4 * 4
    a = pack_reads(r1, stride)
    b = pack_reads(r2, stride) if r2 is not None else (None, None, None)
    return SynthBatch(a[0], a[1], a[2], b[0], b[1], b[2])


def row_bytes(arr: np.ndarray, lens: np.ndarray, i: int) -> bytes:
    return arr[i, : int(lens[i])].tobytes()


def oracle_run(tp: planmod.TrimPlan, batch: SynthBatch, threads: int = 1):
    """-> ((res1, cap2_1, stats1), (res2, None, stats2) | None) through the C oracle."""
    a1, n1, a2, n2 = tp.pack()
    params = tp.params()
    m1 = oracle.trim_mate(a1, n1, params, batch.seq1, batch.qual1, batch.len1, want_cap2=tp.needs_cap2,
                          threads=threads)
    m2 = None
    if tp.paired:
        m2 = oracle.trim_mate(a2, n2, params, batch.seq2, batch.qual2, batch.len2, threads=threads)
    return m1, m2


def format_batch(tp: planmod.TrimPlan, batch: SynthBatch, names1: Sequence[bytes], names2, res1, cap2, res2):
    """results -> [(route, rec1, rec2|None)] using the product's host formatter."""
    out = []
    for i in range(batch.n):
        s1, q1 = row_bytes(batch.seq1, batch.len1, i), row_bytes(batch.qual1, batch.len1, i)
        if tp.paired:
            s2, q2 = row_bytes(batch.seq2, batch.len2, i), row_bytes(batch.qual2, batch.len2, i)
            out.append(hostfmt.format_pair(names1[i], s1, q1, res1[i], names2[i], s2, q2, res2[i], tp))
        else:
            rt, rec = hostfmt.format_single(names1[i], s1, q1, res1[i], cap2[i] if cap2 is not None else None, tp)
            out.append((rt, rec, None))
    return out


def to_pyref_settings(st: planmod.CutadaptConfig) -> pyref.Settings:
    return pyref.Settings(
        ensure_inline_barcode=st.ensure_inline_barcode, trim_polyA=st.trim_polyA,
        trim_polyA_wo_direction=st.trim_polyA_wo_direction, conditional_cutter=st.conditional_cutter,
        min_length=st.min_length, min_quality=st.min_quality, auto_rc=st.auto_rc,
        force_trim_min_length=st.force_trim_min_length, force_anywhere=st.force_anywhere,
        select_rule=st.select_rule, indel_tie=st.indel_tie, case_rule=st.case_rule, shortcut=st.shortcut,
    )


def pyref_run(scheme: str, st: planmod.CutadaptConfig, batch: SynthBatch, names1, names2=None,
              untrimmed_requested=False):
    """The string-slicing restatement, record by record -> [(route, rec1, rec2|None)]."""
    bc = BarcodeConfig(scheme)
    ps = to_pyref_settings(st)
    routes = {"trimmed": 0, "short": 1, "untrimmed": 2}
    out = []
    if batch.seq2 is not None:
        pipe = pyref.PairedPipeline(bc, ps, untrimmed_requested)
        swap = pipe.swap
        for i in range(batch.n):
            r1 = pyref.Read(names1[i].decode(), row_bytes(batch.seq1, batch.len1, i).decode(),
                            row_bytes(batch.qual1, batch.len1, i).decode())
            r2 = pyref.Read(names2[i].decode(), row_bytes(batch.seq2, batch.len2, i).decode(),
                            row_bytes(batch.qual2, batch.len2, i).decode())
            rt, o1, o2 = pipe.process(r1, r2)
            out.append((routes[rt], o1.fastq().encode(), o2.fastq().encode()))
    else:
        pipe = pyref.SinglePipeline(bc, ps, untrimmed_requested)
        for i in range(batch.n):
            r1 = pyref.Read(names1[i].decode(), row_bytes(batch.seq1, batch.len1, i).decode(),
                            row_bytes(batch.qual1, batch.len1, i).decode())
            rt, o1 = pipe.process(r1)
            out.append((routes[rt], o1.fastq().encode(), None))
    return out


def compile_plan(scheme: str, st: planmod.CutadaptConfig, paired: bool, untrimmed_requested=False):
    bc = BarcodeConfig(scheme)
    fn = planmod.compile_paired if paired else planmod.compile_single
    return fn(bc, st, untrimmed_requested)


def read_fastq_gz(path, limit=None):
    """Tiny FASTQ reader for fixtures -> [(name, seq, qual)] as bytes."""
    out = []
    with gzip.open(path, "rb") as fh:
        while True:
            h = fh.readline()
            if not h:
                break
            s, _, q = fh.readline(), fh.readline(), fh.readline()
            out.append((h.rstrip(b"\r\n")[1:], s.rstrip(b"\r\n"), q.rstrip(b"\r\n")))
            if limit and len(out) >= limit:
                break
    return out


def batch_from_records(rec1, rec2=None) -> SynthBatch:
    r1 = [(s.decode(), q.decode()) for _, s, q in rec1]
    r2 = [(s.decode(), q.decode()) for _, s, q in rec2] if rec2 is not None else None
    return batch_from_reads(r1, r2)


def random_dna(rng: random.Random, n: int, alphabet="ACGT") -> str:
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng: random.Random, s: str, n_edits: int, alphabet="ACGT") -> str:
    s = list(s)
    for _ in range(n_edits):
        kind = rng.choice("sid")
        if kind == "s" and s:
            s[rng.randrange(len(s))] = rng.choice(alphabet)
        elif kind == "i":
            s.insert(rng.randrange(len(s) + 1), rng.choice(alphabet))
        elif s:
            del s[rng.randrange(len(s))]
    return "".join(s)


def soft_mask(batch: SynthBatch, fraction: float = 0.2, seed: int = 9) -> None:
    """Lower-case a random stretch of `fraction` of the reads in place (soft-masked bases): the
    aligner sees them through sequence.upper(), the output keeps them."""
    rng = np.random.default_rng(seed)
    for seq in (batch.seq1, batch.seq2):
        if seq is None:
            continue
        n, stride = seq.shape
        rows = np.nonzero(rng.random(n) < fraction)[0]
        for r in rows:
            lo = int(rng.integers(0, stride))
            hi = int(rng.integers(lo, stride + 1))
            seg = seq[r, lo:hi]
            letters = (seg >= ord("A")) & (seg <= ord("Z"))
            seg[letters] |= 0x20


def write_fastq(path: str, names: Sequence[bytes], batch_seq: np.ndarray, batch_qual: np.ndarray, lens: np.ndarray,
                gz_members: int = 0) -> None:
    """Plain-Python FASTQ writer for test inputs.  ``gz_members`` > 0: gzip with that many records per member
    (a multi-member file, like the CLI's own output); 0: one member (``.gz`` names) or plain text."""
    recs = []
    for i, name in enumerate(names):
        n = int(lens[i])
        recs.append(b"@" + name + b"\n" + batch_seq[i, :n].tobytes() + b"\n+\n" + batch_qual[i, :n].tobytes() + b"\n")
    if not path.endswith(".gz"):
        Path(path).write_bytes(b"".join(recs))
        return
    step = gz_members or len(recs) or 1
    with open(path, "wb") as fh:
        for lo in range(0, max(len(recs), 1), step):
            fh.write(gzip.compress(b"".join(recs[lo:lo + step]), 1))
