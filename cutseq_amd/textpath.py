"""Text path of the engine: record-aligned FASTQ text in, finished FASTQ text out (``cs_text_*`` in
``include/cutseq_hip.h``).

Counterpart of what dnaio's reader / writer and the name modifiers do around cutadapt's modifier loop for the
reference (cutseq/run.py:434-441, 751-758 reader; 330, 378, 537-542, 642-645 SuffixRemover / Renamer; 446-471,
760-793 filters and sinks): the device finds the records in the uploaded text, trims them and writes the output
records of the three routes; the host reads / inflates and deflates / writes, nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import abi, capi
from .engine import TrimEngine
from .plan import CutOp, TrimPlan

ROUTES = ("trimmed", "short", "untrimmed")


class TextFormatError(ValueError):
    """A record the reference's reader would refuse (malformed FASTQ, mate ids that differ)."""

    def __init__(self, code: int, record: int, message: str):
        super().__init__(message)
        self.code, self.record = code, record


class ReadTooLong(ValueError):
    """A read beyond CS_MAX_READ (the text path's positions are 32-bit, its long-read kernel serial per read)."""

    def __init__(self, longest: int):
        super().__init__(f"reads longer than {abi.CS_MAX_READ} nt are not supported (a read of {longest} nt)")
        self.longest = longest


def max_tag(plan: TrimPlan) -> int:
    """Most bytes a record name can gain: '_' + every captured cut (Renamer template, cutseq/run.py:378, 643)."""
    if not plan.has_umi:
        return 0
    total = 1
    for chain in (plan.r1, plan.r2):
        if chain is None:
            continue
        total += sum(abs(op.length) for op in chain.ops if isinstance(op, CutOp) and op.capture)
    return total


_EMPTY = np.zeros(16, dtype=np.uint8)  # stands in for an empty text (a null pointer means "no such mate")


def _address(buf) -> int:
    if buf is None:
        return 0
    if isinstance(buf, int):
        return buf
    if len(buf) == 0:
        return _EMPTY.ctypes.data
    return np.frombuffer(buf, dtype=np.uint8).ctypes.data if not isinstance(buf, np.ndarray) else buf.ctypes.data


class TextEngine:
    """``slots`` batches in flight on one :class:`TrimEngine` (its plan, streams and statistics block)."""

    def __init__(self, engine: TrimEngine, slots: int = 3, max_text_bytes: int = 64 << 20, max_records: int = 1 << 18,
                 stride: int = 152, compress: bool = False, bins: int = 0, fasta: bool = False):
        """``compress``: every route's output leaves the device as one gzip member (``res.route_bytes`` then counts
        compressed bytes): what ``.gz`` output files take as they are.  ``bins``: the plan demultiplexes (table form)
        into that many barcodes; the trimmed records of barcode b are route 3 + b (:meth:`routes`)."""
        self.L = capi.load()
        self.engine = engine
        plan = engine.plan
        self.slots, self.max_text_bytes, self.max_records, self.stride = slots, max_text_bytes, max_records, stride
        p = abi.cs_text_params()
        p.has_umi = 1 if plan.has_umi else 0
        p.untrimmed_filter = 1 if plan.untrimmed_filter else 0
        p.reverse_complement = 1 if plan.reverse_complement else 0
        p.compress = 1 if compress else 0
        p.fasta_out = 1 if fasta else 0  # records leave as ">id\nsequence\n" (no qualities to write)
        self.compress = bool(compress)
        p.n_bins = int(bins)
        self.n_routes = 3 + int(bins)
        p.max_tag = max_tag(plan)
        self._keep = []  # the literals must outlive the call
        for field, chain in (("suffix1", plan.r1), ("suffix2", plan.r2)):
            lits = [s.encode() for s in (chain.name_suffixes if chain is not None else ())][:2]
            self._keep += lits
            arr = getattr(p, field)
            for i, lit in enumerate(lits):
                arr[i] = lit
        self._h = C.c_void_p()
        capi.check(self.L.cs_text_create(engine._eng_h, C.byref(p), slots, max_text_bytes, max_records, stride,
                                         C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.L.cs_text_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def submit(self, slot: int, text1, bytes1: int, text2=None, bytes2: int = 0, n_records: int = 0) -> None:
        """Asynchronous.  ``text1`` / ``text2``: buffers (or addresses) holding ``n_records`` complete records each;
        they must stay untouched until :meth:`wait` returns."""
        capi.check(self.L.cs_text_submit(self._h, slot, _address(text1), bytes1, _address(text2) or None, bytes2, n_records))

    def wait(self, slot: int, first_record: int = 0, names=("mate 1", "mate 2")) -> abi.cs_text_result:
        """Blocks until the batch is formatted on the device.  Raises what the reference's reader would raise for a
        bad record.  Reads longer than the rows are not an error: they took the long-read kernel (``res.n_long``)."""
        res = abi.cs_text_result()
        capi.check(self.L.cs_text_wait(self._h, slot, C.byref(res)))
        if res.error == abi.CS_TEXT_ERR_TOO_LONG:
            raise ReadTooLong(int(res.max_len))
        if res.error == abi.CS_TEXT_ERR_MALFORMED:
            raise TextFormatError(res.error, first_record + res.error_record,
                                  f"malformed FASTQ record {first_record + res.error_record + 1} "
                                  "(expected '@' header, sequence, '+' line and a quality line of equal length)")
        if res.error == abi.CS_TEXT_ERR_IDS_DIFFER:
            raise TextFormatError(res.error, first_record + res.error_record,
                                  f"Input read IDs not identical in record {first_record + res.error_record + 1}")
        if res.error == abi.CS_TEXT_ERR_LINE_COUNT:
            raise TextFormatError(res.error, first_record,
                                  f"the text block does not hold the announced number of records ({res.n_records} records, "
                                  f"{res.n_lines[0]} / {res.n_lines[1]} line ends)")
        return res

    def routes(self, slot: int):
        """Behind :meth:`wait`: (bytes[route, mate] as fetched, text_bytes[route, mate], count[route]) of every route."""
        got = np.zeros((self.n_routes, 2), dtype=np.uint64)
        raw = np.zeros((self.n_routes, 2), dtype=np.uint64)
        count = np.zeros(self.n_routes, dtype=np.uint32)
        capi.check(self.L.cs_text_routes(self._h, slot, got.ctypes.data, raw.ctypes.data, count.ctypes.data))
        return got, raw, count

    def fetch(self, slot: int, dst1, dst2=None) -> None:
        """Output text of the batch into the caller's buffers (``res.out_bytes[m]`` bytes each); frees the slot."""
        capi.check(self.L.cs_text_fetch(self._h, slot, _address(dst1) or None, _address(dst2) or None))

    def run(self, text1: bytes, n_records: int, text2: Optional[bytes] = None, slot: int = 0):
        """Synchronous convenience wrapper (tests): -> (streams[route][mate] bytes, counts[route])."""
        self.submit(slot, text1, len(text1), text2, len(text2) if text2 is not None else 0, n_records)
        res = self.wait(slot)
        got, _, count = self.routes(slot)
        out = [np.empty(max(int(res.out_bytes[m]), 1), dtype=np.uint8) for m in range(2)]
        self.fetch(slot, out[0], out[1] if text2 is not None else None)
        if self.n_routes > 3:
            return split_routes(got, out, text2 is not None), [int(c) for c in count]
        return split_routes(res, out, text2 is not None), [int(c) for c in res.route_count]


def split_routes(res, out, paired: bool):
    """-> streams[route][mate] (bytes) from the per-mate buffers :meth:`TextEngine.fetch` filled.  ``res``: the
    ``cs_text_result`` of the batch, or the bytes[route, mate] array of :meth:`TextEngine.routes`."""
    sizes = res.route_bytes if isinstance(res, abi.cs_text_result) else res
    streams = [[b"", b""] for _ in range(len(sizes))]
    for m in range(2 if paired else 1):
        at = 0
        for route in range(len(sizes)):
            n = int(sizes[route][m])
            streams[route][m] = out[m][at:at + n].tobytes()
            at += n
    return streams
