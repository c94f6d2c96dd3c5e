"""Record logic as a SPECIFICATION (test infrastructure, not on the product path: the product formats on the device,
text_kernels.hip.inc, or through csh_format_chunk): header rewriting, pair routing and FASTQ formatting.

This is the string work the reference leaves to cutadapt's ``SuffixRemover``,
``Renamer`` / ``PairedEndRenamer`` (cutseq/run.py:330, 377-380, 537-542, 642-645), the
filter steps (run.py:446-471, 763-792) and dnaio's FASTQ writer.  It consumes the 8-byte
``cs_result`` records the device produced; no trimming arithmetic happens here.
"""
from __future__ import annotations

from typing import Sequence, Tuple

from cutseq_amd import abi

ROUTE_TRIMMED, ROUTE_SHORT, ROUTE_UNTRIMMED = 0, 1, 2
ROUTE_NAMES = ("trimmed", "short", "untrimmed")

_COMPLEMENT = bytes.maketrans(b"ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", b"TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn")


def strip_suffixes(name: bytes, suffixes: Sequence[bytes]) -> bytes:
    """SuffixRemover chain: each literal is tested against the END OF THE WHOLE HEADER."""
    for suf in suffixes:
        if name.endswith(suf):
            name = name[: len(name) - len(suf)]
    return name


def read_id(name: bytes) -> bytes:
    """Renamer.parse_name: ``name.split(maxsplit=1)[0]``; a header without a second field
    (including an empty or all-blank header) is its own id."""
    fields = name.split(None, 1)
    return fields[0] if len(fields) == 2 else name


def pair_id(name: bytes) -> bytes:
    """dnaio.record_names_match's notion of an id: up to the first space or tab."""
    cut = len(name)
    for sep in (b" ", b"\t"):
        p = name.find(sep)
        if p >= 0:
            cut = min(cut, p)
    return name[:cut]


def ids_match(name1: bytes, name2: bytes) -> bool:
    a, b = pair_id(name1), pair_id(name2)
    if a and b and a[-1:] in (b"1", b"2", b"3") and b[-1:] in (b"1", b"2", b"3"):
        a, b = a[:-1], b[:-1]
    return a == b


def route(flags1: int, flags2: int = 0, untrimmed_filter: bool = False) -> int:
    """PairedEndFilter(pair_filter_mode='any') chain: TooShort first, then IsUntrimmedAny."""
    f = flags1 | flags2
    if f & abi.CS_F_TOO_SHORT:
        return ROUTE_SHORT
    if untrimmed_filter and (f & abi.CS_F_UNTRIMMED):
        return ROUTE_UNTRIMMED
    return ROUTE_TRIMMED


def capture(seq: bytes, off: int, length: int) -> bytes:
    return seq[off : off + length]


def fastq_record(name: bytes, seq: bytes, qual: bytes, start: int, stop: int, rc: bool = False) -> bytes:
    s, q = seq[start:stop], qual[start:stop]
    if rc:
        s, q = s.translate(_COMPLEMENT)[::-1], q[::-1]
    return b"@" + name + b"\n" + s + b"\n+\n" + q + b"\n"


def format_single(name: bytes, seq: bytes, qual: bytes, res, cap2, plan) -> Tuple[int, bytes]:
    """One single-end record -> (route, FASTQ bytes)."""
    name = strip_suffixes(name, [s.encode() for s in plan.r1.name_suffixes])
    rid = read_id(name)
    if plan.has_umi:
        tag = capture(seq, int(res["cap_off"]), int(res["cap_len"]))
        if cap2 is not None:
            tag += capture(seq, int(cap2["off"]), int(cap2["len"]))
        rid = rid + b"_" + tag
    rt = route(int(res["flags"]), 0, plan.untrimmed_filter)
    return rt, fastq_record(rid, seq, qual, int(res["start"]), int(res["stop"]), plan.reverse_complement)


def format_pair(name1: bytes, seq1: bytes, qual1: bytes, res1, name2: bytes, seq2: bytes, qual2: bytes, res2,
                plan) -> Tuple[int, bytes, bytes]:
    """One pair -> (route, R1 FASTQ bytes, R2 FASTQ bytes)."""
    name1 = strip_suffixes(name1, [s.encode() for s in plan.r1.name_suffixes])
    name2 = strip_suffixes(name2, [s.encode() for s in plan.r2.name_suffixes])
    if not ids_match(name1, name2):
        raise ValueError(
            f"Input read IDs not identical: '{read_id(name1).decode(errors='replace')}' != "
            f"'{read_id(name2).decode(errors='replace')}'"
        )
    id1, id2 = read_id(name1), read_id(name2)
    if plan.has_umi:
        tag = b"_" + capture(seq1, int(res1["cap_off"]), int(res1["cap_len"])) + capture(
            seq2, int(res2["cap_off"]), int(res2["cap_len"])
        )
        id1, id2 = id1 + tag, id2 + tag
    rt = route(int(res1["flags"]), int(res2["flags"]), plan.untrimmed_filter)
    rec1 = fastq_record(id1, seq1, qual1, int(res1["start"]), int(res1["stop"]))
    rec2 = fastq_record(id2, seq2, qual2, int(res2["start"]), int(res2["stop"]))
    return rt, rec1, rec2
