"""Run statistics and the two reports the reference emits: the stderr TSV line
(``minimal_report``, cutseq/run.py:489, 810) and ``--json-file`` (``json_report``,
cutseq/run.py:222-302).

The reference fills both from cutadapt's ``Statistics`` object.  cutadapt is not available to
this build, so the field layout below follows cutadapt's ``minimal_report`` / ``Statistics.as_json``
as recalled (same caveat as the trimming semantics: unpinned).  One reference quirk is kept on
purpose: cutseq patches ``Statistics._collect_modifier`` to swallow the assertion that fires on a
second ``AdapterCutter`` (run.py:59-73), so ``w/adapters`` and ``adapters_read*`` only ever
describe the FIRST AdapterCutter of each mate's chain.
"""
from __future__ import annotations

import json
from typing import Optional

import numpy as np

from . import __version__, abi
from .plan import AdapterOp, MateChain, TrimPlan

# cutadapt's filter identifiers, in report order
FILTER_KEYS = ("too_short", "too_long", "too_many_n", "too_many_expected_errors", "casava_filtered",
               "discard_trimmed", "discard_untrimmed")

# Adapter.descriptive_identifier() of the classes the reference instantiates
ADAPTER_TYPES = {
    "BackAdapter": ("three_prime_end", "regular_three_prime"),
    "RightmostFrontAdapter": ("five_prime_end", "rightmost_five_prime"),
    "PrefixAdapter": ("five_prime_end", "anchored_five_prime"),
    "SuffixAdapter": ("three_prime_end", "anchored_three_prime"),
    "NonInternalBackAdapter": ("three_prime_end", "noninternal_three_prime"),
    "NonInternalFrontAdapter": ("five_prime_end", "noninternal_five_prime"),
}


def new_totals() -> dict:
    return {"in_pairs": 0, "routes": [0, 0, 0], "in_bp": [0, 0], "out_bp": [0, 0], "written_bp": [0, 0]}


def account_chunk(totals: dict, tp: TrimPlan, len1: np.ndarray, res1: np.ndarray,
                  len2: Optional[np.ndarray] = None, res2: Optional[np.ndarray] = None) -> None:
    """Fold one chunk's results into the run totals.  ``written_bp`` counts what reaches the final
    sink only (cutadapt's ``written_bp``): pairs routed to the short / untrimmed files are excluded."""
    flags = res1["flags"].astype(np.uint8)
    if res2 is not None:
        flags = flags | res2["flags"].astype(np.uint8)
    dropped = (flags & abi.CS_F_TOO_SHORT) != 0
    if tp.untrimmed_filter:
        dropped |= (flags & abi.CS_F_UNTRIMMED) != 0
    keep = ~dropped
    totals["in_pairs"] += int(len(res1))
    for m, (lens, res) in enumerate(((len1, res1), (len2, res2))):
        if res is None:
            continue
        span = res["stop"].astype(np.int64) - res["start"].astype(np.int64)
        totals["in_bp"][m] += int(lens.sum(dtype=np.int64))
        totals["out_bp"][m] += int(span.sum())
        totals["written_bp"][m] += int(span[keep].sum())


def merge_totals(into: dict, part: dict) -> None:
    into["in_pairs"] += part["in_pairs"]
    for key in ("routes", "in_bp", "out_bp", "written_bp"):
        if len(into[key]) < len(part[key]):  # demultiplexing runs: one more stream per barcode
            into[key].extend([0] * (len(part[key]) - len(into[key])))
        for i, v in enumerate(part[key]):
            into[key][i] += v


def first_adapter(chain: Optional[MateChain]):
    """(stat slot, op) of the first AdapterCutter in a mate's chain, or (None, None)."""
    if chain is None:
        return None, None
    for slot, op in enumerate(chain.ops):
        if isinstance(op, AdapterOp):
            return slot, op
    return None, None


def _mate_sum(totals: dict, mate: int, field: str) -> int:
    """``totals["stats"]``: one [mate 1, mate 2] pair of ``cs_stats.as_dict()`` blocks per device worker."""
    return sum(int(pair[mate][field]) for pair in totals["stats"])


def _matched(totals: dict, mate: int, slot: Optional[int]) -> int:
    if slot is None:
        return 0
    return sum(int(pair[mate]["op_matched"][slot]) for pair in totals["stats"])


def written_pairs(totals: dict) -> int:
    """Pairs that reached a final sink: the trimmed stream, or one stream per barcode when demultiplexing."""
    return totals["routes"][0] + sum(totals["routes"][3:])


def minimal_report(tp: TrimPlan, totals: dict) -> str:
    """Header line + value line, tab separated (cutadapt ``minimal_report`` field order)."""
    fields = ["status", "in_reads", "in_bp", "too_short", "too_long", "too_many_n", "out_reads",
              "w/adapters", "qualtrim_bp", "out_bp"]
    s1, _ = first_adapter(tp.r1)
    vals = ["OK", totals["in_pairs"], sum(totals["in_bp"]), totals["routes"][1], 0, 0, written_pairs(totals),
            _matched(totals, 0, s1), _mate_sum(totals, 0, "qualtrim_bp"), totals["written_bp"][0]]
    if tp.paired:
        s2, _ = first_adapter(tp.r2)
        fields += ["w/adapters2", "qualtrim2_bp", "out2_bp"]
        vals += [_matched(totals, 1, s2), _mate_sum(totals, 1, "qualtrim_bp"), totals["written_bp"][1]]
    return "\t".join(fields) + "\n" + "\t".join(str(v) for v in vals)


def error_lengths(op: AdapterOp) -> list:
    """cutadapt ``ErrorRanges._compute_lengths``: for 1, 2, ... allowed errors the longest match that still allows
    one error fewer, and "the last number is always the adapter length" (appended when the list is empty or ends
    below it) -- rate 0.2, 20 nt: [4, 9, 14, 19, 20]; an adapter that allows no error at all: [m]."""
    lengths = [int(errors / op.max_error_rate) - 1 for errors in range(1, int(op.max_error_rate * op.m) + 1)]
    if not lengths or lengths[-1] < op.m:
        lengths.append(op.m)
    return lengths


def _adapter_json(op: AdapterOp, name: str, matches: int) -> dict:
    end, type_name = ADAPTER_TYPES[op.kind_name]
    side = {
        "type": type_name,
        "sequence": op.sequence,
        "error_rate": op.max_error_rate,
        "indels": True,
        "error_lengths": error_lengths(op),
        "matches": matches,
        # cutadapt reports the bases in front of the adapter for 3' ends only; the one adapter the reference's
        # report ever describes per mate is the 5' one (first AdapterCutter, run.py:59-73)
        "adjacent_bases": None,
        "dominant_adjacent_base": None,
        "trimmed_lengths": [],  # the reference empties these lists itself (run.py:286-300)
    }
    d = {"name": name, "total_matches": matches, "on_reverse_complement": None, "linked": False,
         "five_prime_end": None, "three_prime_end": None}
    d[end] = side
    return d


def _host_threads() -> int:
    from . import fastq
    return fastq.pool_size()


def json_report(tp: TrimPlan, totals: dict, barcode, input1, input2, output1, output2, short1, short2,
                untrimmed1, untrimmed2) -> dict:
    """Same header block as the reference's ``json_report`` (run.py:262-283) followed by the
    ``Statistics.as_json()`` sections; engine-specific counters live under ``"engine"``."""
    paired = tp.paired
    s1, a1 = first_adapter(tp.r1)
    s2, a2 = first_adapter(tp.r2) if paired else (None, None)
    filtered = {k: None for k in FILTER_KEYS}
    filtered["too_short"] = totals["routes"][1]
    q1, q2 = _mate_sum(totals, 0, "qualtrim_bp"), (_mate_sum(totals, 1, "qualtrim_bp") if paired else None)
    engine = {"name": "cutseq_amd", "version": __version__, "devices": totals.get("devices"),
              "seconds": totals.get("seconds"), "is_untrimmed_any": totals["routes"][2] if tp.untrimmed_filter else None,
              "per_device": totals["stats"],
              **({"ranks": totals["ranks"], "ranks_split": totals.get("ranks_split"),
                  "threads_per_rank": totals.get("threads_per_rank")} if totals.get("ranks") else {}),
              **({"demultiplexed": dict(zip(totals.get("bin_names") or [], totals["routes"][3:]))}
                 if tp.demux is not None else {})}
    # Key order as the reference's file has it: its own header dict (tag, cutadapt_version, input, output, barcode),
    # then ``d.update(stats.as_json())`` -- keys it already holds keep their place and take as_json's value, the others
    # follow in as_json's order (schema_version, python_version, command_line_arguments, cores, read_counts, ...).
    # Engine-specific counters come last, under a key cutadapt does not have.
    import platform
    import sys
    d = {
        "tag": "Cutadapt report",
        # the report layout is cutadapt 5's; the numbers come from this engine (local version label says so)
        "cutadapt_version": f"5.0+cutseq.amd.{__version__}",
        "input": {"path1": input1, "path2": input2, "paired": True if input2 else False},
        "output": {"output1": output1, "output2": output2, "short1": short1, "short2": short2,
                   "untrimmed1": untrimmed1, "untrimmed2": untrimmed2},
        "barcode": barcode.to_dict(),
        "schema_version": [0, 3],
        "python_version": platform.python_version(),
        "command_line_arguments": list(sys.argv[1:]),
        "cores": totals.get("threads") or _host_threads(),  # (cutadapt: the runner's cores; here: the host pool)
        "read_counts": {
            "input": totals["in_pairs"],
            "filtered": filtered,
            "output": written_pairs(totals),
            "reverse_complemented": None,
            "read1_with_adapter": _matched(totals, 0, s1) if a1 is not None else None,
            "read2_with_adapter": (_matched(totals, 1, s2) if a2 is not None else None) if paired else None,
        },
        "basepair_counts": {
            "input": sum(totals["in_bp"]),
            "input_read1": totals["in_bp"][0],
            "input_read2": totals["in_bp"][1] if paired else None,
            "quality_trimmed": q1 + (q2 or 0),
            "quality_trimmed_read1": q1,
            "quality_trimmed_read2": q2,
            "poly_a_trimmed": None,  # cutadapt's PolyATrimmer is not in the reference's chain
            "poly_a_trimmed_read1": None,
            "poly_a_trimmed_read2": None,
            "output": sum(totals["written_bp"]),
            "output_read1": totals["written_bp"][0],
            "output_read2": totals["written_bp"][1] if paired else None,
        },
        "adapters_read1": [_adapter_json(a1, "1", _matched(totals, 0, s1))] if a1 is not None else [],
        "adapters_read2": ([_adapter_json(a2, "2", _matched(totals, 1, s2))] if a2 is not None else []) if paired else None,
        "poly_a_trimmed_read1": None,
        "poly_a_trimmed_read2": None,
        "engine": engine,
    }
    return d


def write_json(path: str, report: dict) -> None:
    with open(path, "w") as fh:
        fh.write(json.dumps(report, indent=2))


class Progress:
    """Counterpart of cutadapt's ``Progress()``, which the reference hands to ``runner.run`` unconditionally
    (cutseq/run.py:473, 794): a status line on stderr while the run goes on and a ``Done`` line with reads, µs per read
    and M reads per minute at its end (layout recalled from cutadapt's ``utils.Progress``).  The running line -- carriage
    returns, one update per second -- is only drawn on a terminal; the final line is always written, as the reference's
    is.  ``CUTSEQ_PROGRESS=0`` silences both."""

    def __init__(self, stream=None):
        import os
        import sys
        import threading
        import time
        self._time = time
        self.stream = stream if stream is not None else sys.stderr
        self.enabled = os.environ.get("CUTSEQ_PROGRESS", "1") != "0"
        self.live = self.enabled and hasattr(self.stream, "isatty") and self.stream.isatty()
        self.t0 = time.time()
        self.last = self.t0
        self.total = 0
        self._lock = threading.Lock()
        self._frames = 0

    def _line(self, head: str, now: float) -> str:
        elapsed = max(now - self.t0, 1e-9)
        per_item = elapsed / self.total if self.total else 0.0
        per_minute = self.total / elapsed * 60.0 / 1e6
        h, rem = divmod(int(elapsed), 3600)
        m, sec = divmod(rem, 60)
        return f"{head:<10s} {h:02d}:{m:02d}:{sec:02d} {self.total:13,d} reads @ {per_item * 1e6:5.1F} µs/read; {per_minute:6.2F} M reads/minute"

    def update(self, n: int) -> None:
        with self._lock:
            self.total += int(n)
            now = self._time.time()
            if self.live and now - self.last >= 1.0:
                self.last = now
                self._frames += 1
                k = self._frames % 8
                print("\r" + self._line("[" + "-" * k + "8<" + "-" * (7 - k) + "]", now), end="", file=self.stream, flush=True)

    def close(self) -> None:
        if not self.enabled:
            return
        with self._lock:
            print(("\r" if self.live else "") + self._line("Done", self._time.time()), file=self.stream, flush=True)
