"""Look-up tables for the demultiplexing op (``CS_OP_DEMUX``, include/cutseq_hip.h) -- an extension, the
reference has no demultiplexer (BASELINE.json config 5).

The op stands for one ``AdapterCutter([PrefixAdapter(barcode, rate)])`` per barcode.  Whether such an
adapter matches, and how many bases it removes, depends on nothing but the first ``m + k`` bases of the
read (``Aligner.locate`` stops at column ``m + k`` without a free query start).  So the table is built by
RUNNING those adapter ops -- the device's own, through the C ABI -- on every possible prefix of every
length 0 .. m + k over the alphabet {A, C, T, G, other} and merging the outcomes (an exact copy of a barcode
first, else the lowest index; more than one claimant is flagged): "equal to B independent
``--ensure-inline-barcode`` runs" holds by construction.
"""
from __future__ import annotations

import threading

import numpy as np

from . import abi
from .plan import DemuxOp, MateChain, TrimPlan

DIGITS = b"ACTGN"  # digit d of the base-5 index = bits 2:1 of the ASCII code for A, C, T, G; 4 = anything else


def table_entries(span: int) -> int:
    return (5 ** (span + 1) - 1) // 4


def all_prefixes(span: int):
    """Every string over DIGITS of length 0 .. span as rows of a read batch, in table order
    -> (seq[n, stride] u8, len[n] u16)."""
    n = table_entries(span)
    stride = max(4, (span + 3) // 4 * 4)
    seq = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    letters = np.frombuffer(DIGITS, dtype=np.uint8)
    start = 0
    for length in range(span + 1):
        count = 5 ** length
        idx = np.arange(count, dtype=np.int64)
        for t in range(length):
            seq[start:start + count, t] = letters[(idx // 5 ** t) % 5]
        lens[start:start + count] = length
        start += count
    return seq, lens


def build_table(op: DemuxOp, device: int = 0, select_rule: int = abi.CS_SELECT_LEFTMOST,
                indel_tie: int = abi.CS_TIE_INSERTION) -> np.ndarray:
    """The table of ``op`` (uint16, ``table_entries(m + k)`` entries), computed on ``device``: ONE pass of the op's
    other form -- the barcodes' own PrefixAdapter ops on every candidate, merged on the device (CS_DEMUX_BY_OPS,
    ``cs_plan_set_demux_ops``) -- over every possible prefix.  (Until round 3 this ran one single-barcode engine per
    barcode and merged in numpy: five seconds for a 96-plex, most of a short run.)"""
    from .engine import TrimEngine

    span = op.m + op.k
    seq, lens = all_prefixes(span)
    qual = np.full_like(seq, ord("I"))
    n = len(lens)
    probe = DemuxOp(list(op.barcodes), op.max_error_rate, abi.CS_F_INLINE, required=False, by_ops=True)
    one = TrimPlan(r1=MateChain([probe]), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                   select_rule=select_rule, indel_tie=indel_tie)
    bc = np.empty(n, dtype=np.uint8)
    with TrimEngine(one, device=device, slots=1, max_reads=n, max_stride=seq.shape[1]) as eng:
        res, _, _ = eng.submit(0, seq, qual, lens, bc=bc)
        eng.wait(0)
    ambiguous = ((res["flags"] & abi.CS_F_AMBIGUOUS) != 0).astype(np.uint16)
    table = bc.astype(np.uint16) | (res["start"].astype(np.uint16) << 8) | (ambiguous << 14)
    return np.ascontiguousarray(table, dtype=np.uint16)


_table_lock = threading.RLock()  # (re-entrant: building a table runs an engine of its own)


def ensure_tables(plan: TrimPlan, device: int = 0) -> None:
    """Build the tables the plan's demultiplexing ops still lack (once: engines of several devices share the plan)."""
    with _table_lock:
        for _mate, _i, op in plan.demux_ops():
            if op.table is None and op.tabulated:  # (longer barcodes: no table, the device runs their ops)
                op.table = build_table(op, device, plan.select_rule, plan.indel_tie)


def ambiguous_prefixes(op: DemuxOp) -> int:
    """How many full-length prefixes (m + k bases of A/C/G/T only) more than one barcode claims."""
    if not op.tabulated:
        raise ValueError("no table of every prefix for barcodes this long")
    span = op.m + op.k
    start = table_entries(span - 1)
    block = op.table[start:]
    idx = np.arange(len(block), dtype=np.int64)
    acgt = np.ones(len(block), dtype=bool)
    for t in range(span):
        acgt &= (idx // 5 ** t) % 5 != 4
    return int(((block & 0x4000) != 0)[acgt].sum())


def read_barcode_file(path: str):
    """``name<TAB>sequence`` (or ``sequence`` alone) per line, ``#`` comments -> ([names], [sequences])."""
    names, codes = [], []
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            fields = line.replace(",", "\t").split()
            if len(fields) == 1:
                names.append(fields[0].upper())
                codes.append(fields[0].upper())
            else:
                names.append(fields[0])
                codes.append(fields[1].upper())
    if len(set(names)) != len(names):
        raise ValueError(f"{path}: duplicate barcode name")
    for name in names:  # names become parts of output file names
        if not name or name in (".", "..") or any(c in name for c in "/\\\0") or name != name.strip():
            raise ValueError(f"{path}: barcode name {name!r} cannot be used in a file name")
    return names, codes
