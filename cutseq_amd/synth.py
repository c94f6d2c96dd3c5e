"""Seeded synthetic read generator: TAKARAV3-shaped libraries.

Molecule (top strand):  P5 | mask5 | insert | mask3 | UMI | P7
    R1 = mask5 + insert + mask3 + UMI + p7.fw ... (read-through when the insert is short)
    R2 = rc(UMI) + rc(mask3) + rc(insert) + rc(mask5) + p5.rc ...
Mixture (mirrors the reference fixture test/input_R{1,2}.fq.gz): ~35 % of inserts shorter than
the read so the 3' adapter shows up (positions skewed towards the read end), ~9 % of those end
with only a 3..19 nt adapter prefix, 0.1 % carry a 5' adapter artefact, 2 % carry a poly-T/A
stretch, 0.7 % N, 1 % substitutions, a few single-base indels inside the adapter, qualities
from the four bins the fixture uses (# - 9 I) with a degrading tail on 20 % of the reads.

Everything is numpy-vectorised and chunked; ``seed`` + chunk index fully determines the bytes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Tuple

import numpy as np

from .common import BarcodeConfig, reverse_complement

DEFAULT_SEED = 0xC0FFEE
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


@dataclass
class SynthBatch:
    """SoA batch as the C ABI wants it (uint8 [n, stride] x2, uint16 [n])."""

    seq1: np.ndarray
    qual1: np.ndarray
    len1: np.ndarray
    seq2: np.ndarray | None
    qual2: np.ndarray | None
    len2: np.ndarray | None

    @property
    def n(self) -> int:
        return self.seq1.shape[0]

    @property
    def stride(self) -> int:
        return self.seq1.shape[1]


def _stride_for(read_len: int) -> int:
    return (read_len + 3) // 4 * 4


def _rand_bases(rng, shape):
    return _BASES[rng.integers(0, 4, size=shape, dtype=np.uint8)]


def _qualities(rng, n, L):
    q = np.full((n, L), ord("I"), dtype=np.uint8)
    r = rng.random((n, L), dtype=np.float32)
    q[r < 0.16] = ord("9")
    q[r < 0.08] = ord("-")
    # degrading tail on 20 % of the reads
    tail = rng.random(n) < 0.20
    t0 = rng.integers(int(L * 0.6), L, size=n)
    pos = np.arange(L)[None, :]
    in_tail = tail[:, None] & (pos >= t0[:, None])
    r2 = rng.random((n, L), dtype=np.float32)
    q[in_tail & (r2 < 0.55)] = ord("#")
    q[in_tail & (r2 >= 0.55) & (r2 < 0.80)] = ord("-")
    return q


def _apply_errors(rng, seq, qual, region_lo, region_hi, sub_rate, indel_frac, n_rate):
    """substitutions everywhere, one single-base indel inside [region_lo, region_hi) for a few reads, Ns."""
    n, L = seq.shape
    sub = rng.random((n, L), dtype=np.float32) < sub_rate
    seq[sub] = _rand_bases(rng, int(sub.sum()))
    # single-base deletion or insertion inside the adapter region
    has_region = region_hi > region_lo
    pick = has_region & (rng.random(n) < indel_frac)
    if pick.any():
        d = np.where(pick, rng.integers(0, 1 << 30, size=n) % np.maximum(region_hi - region_lo, 1) + region_lo, L + 1)
        is_del = rng.random(n) < 0.5
        pos = np.arange(L)[None, :]
        idx_del = np.minimum(pos + (pos >= d[:, None]), L - 1)
        idx_ins = np.maximum(pos - (pos > d[:, None]), 0)
        idx = np.where(is_del[:, None], idx_del, idx_ins)
        seq[:] = np.take_along_axis(seq, idx, axis=1)
        ins_rows = np.nonzero(pick & ~is_del)[0]
        seq[ins_rows, np.minimum(d[ins_rows], L - 1)] = _rand_bases(rng, ins_rows.size)
    nmask = rng.random((n, L), dtype=np.float32) < n_rate
    seq[nmask] = ord("N")
    qual[nmask] = ord("#")


def _mate(rng, L, insert, ins_len, head, tail_fixed, adapter, art5, art_seq):
    """Assemble one mate: [5' artefact] + head + insert + tail_fixed + adapter + random filler.

    Returns (seq [n, L], adapter_region_lo, adapter_region_hi)."""
    n = insert.shape[0]
    h, t, a = head.shape[1], tail_fixed.shape[1], len(adapter)
    pos = np.arange(L)[None, :]
    shift = np.where(art5, len(art_seq), 0)[:, None]
    p = pos - shift
    adapter_arr = np.frombuffer(adapter.encode(), dtype=np.uint8)
    out = _rand_bases(rng, (n, L))  # filler past the adapter
    ins_hi = h + ins_len[:, None]
    tail_hi = ins_hi + t
    ad_hi = tail_hi + a
    if h:
        out = np.where((p >= 0) & (p < h), np.take_along_axis(head, np.clip(p, 0, h - 1), axis=1), out)
    out = np.where((p >= h) & (p < ins_hi),
                   np.take_along_axis(insert, np.clip(p - h, 0, insert.shape[1] - 1), axis=1), out)
    if t:
        out = np.where((p >= ins_hi) & (p < tail_hi),
                       np.take_along_axis(tail_fixed, np.clip(p - ins_hi, 0, t - 1), axis=1), out)
    out = np.where((p >= tail_hi) & (p < ad_hi), adapter_arr[np.clip(p - tail_hi, 0, a - 1)], out)
    if art5.any():
        art_arr = np.frombuffer(art_seq.encode(), dtype=np.uint8)
        out = np.where(pos < shift, art_arr[np.clip(pos, 0, len(art_seq) - 1)], out)
    lo = np.clip(tail_hi[:, 0] + shift[:, 0], 0, L)
    hi = np.clip(ad_hi[:, 0] + shift[:, 0], 0, L)
    return np.ascontiguousarray(out), lo, hi


def generate_pairs(n: int, read_len: int = 150, scheme: str | BarcodeConfig | None = None,
                   seed: int = DEFAULT_SEED, chunk_index: int = 0, single_end: bool = False,
                   adapter_fraction: float = 0.35, partial_fraction: float = 0.09,
                   poly_fraction: float = 0.02, art5_fraction: float = 0.001,
                   sub_rate: float = 0.01, indel_frac: float = 0.03, n_rate: float = 0.007) -> SynthBatch:
    """Generate ``n`` pairs (or single reads) of ``read_len`` bases."""
    if scheme is None:
        from .common import BUILDIN_ADAPTERS
        scheme = BUILDIN_ADAPTERS["TAKARAV3"]
    bc = scheme if isinstance(scheme, BarcodeConfig) else BarcodeConfig(scheme)
    rng = np.random.default_rng([seed, chunk_index])
    L = read_len
    stride = _stride_for(L)
    m5, m3, u5, u3 = bc.mask5.len, bc.mask3.len, bc.umi5.len, bc.umi3.len
    i5, i3 = bc.inline5.fw, bc.inline3.fw
    max_ins = L + 40
    # insert lengths: adapter-bearing reads skew towards the read end
    r = rng.random(n)
    full = L - (len(i5) + u5 + m5)  # insert length at which the 3' structure just leaves R1
    short_len = (full - 1 - (rng.random(n) ** 2.0) * (full - 25)).astype(np.int64)
    partial_len = full - (m3 + u3 + len(i3)) - rng.integers(3, 20, size=n)
    long_len = rng.integers(full, max_ins, size=n)
    ins_len = np.where(r < adapter_fraction, short_len, long_len)
    ins_len = np.where((r >= adapter_fraction) & (r < adapter_fraction + partial_fraction), partial_len, ins_len)
    ins_len = np.clip(ins_len, 1, max_ins).astype(np.int64)
    insert = _rand_bases(rng, (n, max_ins))
    # poly-T at the 5' end of the insert as R1 sees it ('-' strand libraries), poly-A for '+'
    poly = rng.random(n) < poly_fraction
    plen = rng.integers(10, 41, size=n)
    pos_i = np.arange(max_ins)[None, :]
    if bc.strand == "+":
        at = poly[:, None] & (pos_i >= (ins_len - plen)[:, None]) & (pos_i < ins_len[:, None])
        insert[at] = ord("A")
    else:
        at = poly[:, None] & (pos_i < np.minimum(plen, ins_len)[:, None])
        insert[at] = ord("T")
    umi5 = _rand_bases(rng, (n, max(u5, 1)))[:, :u5]
    umi3 = _rand_bases(rng, (n, max(u3, 1)))[:, :u3]
    mask5 = _rand_bases(rng, (n, max(m5, 1)))[:, :m5]
    mask3 = _rand_bases(rng, (n, max(m3, 1)))[:, :m3]

    def fixed(s):
        return np.broadcast_to(np.frombuffer(s.encode(), dtype=np.uint8), (n, len(s)))

    head1 = np.concatenate([fixed(i5), umi5, mask5], axis=1)
    tail1 = np.concatenate([mask3, umi3, fixed(i3)], axis=1)
    art5 = rng.random(n) < art5_fraction
    seq1, lo1, hi1 = _mate(rng, L, insert, ins_len, head1, tail1, bc.p7.fw, art5, bc.p5.fw)
    qual1 = _qualities(rng, n, L)
    _apply_errors(rng, seq1, qual1, lo1, hi1, sub_rate, indel_frac, n_rate)
    out1 = _pack(seq1, qual1, L, stride)
    if single_end:
        return SynthBatch(out1[0], out1[1], out1[2], None, None, None)
    # mate 2 reads the bottom strand
    rc_idx = np.clip(ins_len[:, None] - 1 - np.arange(max_ins)[None, :], 0, max_ins - 1)
    insert_rc = _COMP[np.take_along_axis(insert, rc_idx, axis=1)]
    head2 = _COMP[tail1[:, ::-1]]
    tail2 = _COMP[head1[:, ::-1]]
    art5b = rng.random(n) < art5_fraction
    seq2, lo2, hi2 = _mate(rng, L, insert_rc, ins_len, head2, tail2, bc.p5.rc, art5b, bc.p7.rc)
    qual2 = _qualities(rng, n, L)
    _apply_errors(rng, seq2, qual2, lo2, hi2, sub_rate, indel_frac, n_rate)
    out2 = _pack(seq2, qual2, L, stride)
    return SynthBatch(out1[0], out1[1], out1[2], out2[0], out2[1], out2[2])


def _pack(seq, qual, L, stride):
    n = seq.shape[0]
    if stride != L:
        s = np.zeros((n, stride), dtype=np.uint8)
        q = np.zeros((n, stride), dtype=np.uint8)
        s[:, :L] = seq
        q[:, :L] = qual
    else:
        s, q = np.ascontiguousarray(seq), np.ascontiguousarray(qual)
    return s, q, np.full(n, L, dtype=np.uint16)


def generate_single_adapter(n: int, read_len: int = 150, adapter: str = "AGATCGGAAGAGC",
                            seed: int = DEFAULT_SEED, chunk_index: int = 0) -> SynthBatch:
    """BASELINE.json config 2: single-end reads, one 3' adapter."""
    scheme = f"ACACGACGCTCTTCCGATCT>{adapter}"
    return generate_pairs(n, read_len, scheme, seed, chunk_index, single_end=True, poly_fraction=0.0,
                          art5_fraction=0.0)


def iter_chunks(total: int, chunk: int) -> Iterator[Tuple[int, int]]:
    for i, lo in enumerate(range(0, total, chunk)):
        yield i, min(chunk, total - lo)


def headers(n: int, mate: int, first: int = 0):
    """``SIM:<idx> <mate>:N:0:IDX`` (id + comment, like the fixture's Illumina headers)."""
    return [f"SIM:{first + i} {mate}:N:0:IDX" for i in range(n)]
