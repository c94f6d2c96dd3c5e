"""Op-chain compiler: ``BarcodeConfig`` + flags -> per-mate op tables for the device.

Counterpart of the modifier-list assembly in the reference's ``pipeline_single``
(cutseq/run.py:326-426) and ``pipeline_paired`` (cutseq/run.py:533-731).  Where the
reference instantiates cutadapt ``Modifier`` objects, this module emits plain
descriptors (:class:`AdapterOp`, :class:`CutOp`, :class:`QTrimOp`) in the same order;
``TrimPlan.pack()`` turns them into the ``cs_op`` POD table of ``include/cutseq_hip.h``.
Header work (SuffixRemover, Renamer) stays on the host and is described by
``MateChain.name_suffixes`` / ``TrimPlan.has_umi``.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

from . import abi
from .common import BarcodeConfig

MAX_ERRORS = 0.2  # hard-coded in the reference (run.py:326, 533)
POLY_MAX_ERRORS = 0.15  # run.py:389, 674
POLY_LENGTH = 100  # run.py:390, 675


class CutadaptConfig:
    """Settings bag; same attribute names and defaults as cutseq/run.py:198-219."""

    def __init__(self):
        self.rname_suffix = False
        self.ensure_inline_barcode = False
        self.trim_polyA = False
        self.trim_polyA_wo_direction = False
        self.conditional_cutter = True
        self.min_length = 20
        self.min_quality = 20
        self.auto_rc = False
        self.dry_run = False
        self.threads = 1
        self.json_file = None
        self.force_trim_min_length = 50
        self.force_anywhere = False
        # not in the reference: the cutadapt rules this build restates from recollection, one switch
        # each (tools/parity_exposure.py counts what every switch changes)
        self.select_rule = abi.CS_SELECT_LEFTMOST  # cutadapt >= 4.0 candidate selection
        self.shortcut = abi.CS_SHORTCUT_NONE       # cutadapt >= 3: no str.find before the aligner
        self.case_rule = abi.CS_CASE_FOLD          # match_to aligns sequence.upper()
        self.indel_tie = abi.CS_TIE_INSERTION      # SURVEY appendix B.2 order
        # extension (BASELINE.json config 5): demultiplex on these inline barcodes (replaces the scheme's inline5)
        self.demux_barcodes = None


@dataclass
class AdapterOp:
    """One ``AdapterCutter([<adapter>], times=1)``."""

    kind_name: str  # cutadapt class the reference instantiates
    sequence: str
    max_error_rate: float
    min_overlap: int
    where: int
    remove: int
    rightmost: bool = False
    shortcut: int = abi.CS_SHORTCUT_NONE
    match_flag: int = 0
    required: bool = False

    def __post_init__(self):
        # cutadapt SingleAdapter: upper-case, U->T, min_overlap capped at the length
        self.sequence = self.sequence.upper().replace("U", "T")
        if not self.sequence:
            raise ValueError("Adapter sequence is empty")
        if len(self.sequence) > abi.CS_MAX_ADAPTER:
            raise ValueError(
                f"adapter of {len(self.sequence)} nt exceeds CS_MAX_ADAPTER={abi.CS_MAX_ADAPTER}"
            )
        if self.max_error_rate >= 1:
            self.max_error_rate = self.max_error_rate / len(self.sequence)
        self.min_overlap = min(self.min_overlap, len(self.sequence))

    @property
    def m(self) -> int:
        return len(self.sequence)

    @property
    def k(self) -> int:
        return int(self.max_error_rate * self.m)  # C double -> int truncation

    def thresholds(self) -> List[int]:
        """thr[L] = largest cost with ``cost <= L * rate`` (evaluated in IEEE double)."""
        out = []
        for length in range(self.m + 1):
            bound = length * self.max_error_rate
            c = int(bound)
            while c + 1 <= bound:
                c += 1
            out.append(c)
        return out

    def __repr__(self):
        seq = self.sequence if len(set(self.sequence)) > 1 or self.m < 12 else f"{self.sequence[0]}x{self.m}"
        return (
            f"AdapterCutter({self.kind_name}(sequence={seq!r}, max_error_rate={self.max_error_rate}, "
            f"min_overlap={self.min_overlap}))"
        )


@dataclass
class CutOp:
    """``UnconditionalCutter(length)`` or ``ConditionalCutter(length, force_trim_min_length)``."""

    length: int
    conditional: bool = False
    force_min_len: int = 50
    capture: int = 0  # 0 none, 1 first capture, 2 second capture

    def __repr__(self):
        if self.conditional:
            return f"ConditionalCutter(length={self.length}, force_trim_min_length={self.force_min_len})"
        return f"UnconditionalCutter(length={self.length})"


@dataclass
class QTrimOp:
    cutoff_back: int
    base: int = 33

    def __repr__(self):
        return f"QualityTrimmer(cutoff_front=0, cutoff_back={self.cutoff_back}, base={self.base})"


@dataclass
class DemuxOp:
    """Extension (BASELINE.json config 5): one pass over ``barcodes`` where the reference would run
    ``AdapterCutter([PrefixAdapter(barcode, max_error_rate)])`` once per barcode with
    ``--ensure-inline-barcode`` (cutseq/run.py:357-362, 592-597).  ``table`` is the look-up table of
    ``include/cutseq_hip.h`` (CS_OP_DEMUX); :mod:`cutseq_amd.demux` builds it on the device."""

    barcodes: Sequence[str]
    max_error_rate: float
    match_flag: int = 0
    required: bool = True
    table: Optional[object] = None  # numpy uint16 array once built
    by_ops: bool = False  # the barcodes' own ops even where a table would fit (how the tables themselves are built)
    at_end: bool = False  # the barcodes sit at the 3' end of the read (SuffixAdapter ops; second form only)

    def __post_init__(self):
        self.barcodes = [b.upper().replace("U", "T") for b in self.barcodes]
        if not self.barcodes or len(self.barcodes) > 255:
            raise ValueError("between 1 and 255 barcodes, please")
        if len({len(b) for b in self.barcodes}) != 1 or not self.barcodes[0]:
            raise ValueError("demultiplexing needs barcodes of one common, non-zero length")
        if len(set(self.barcodes)) != len(self.barcodes):
            raise ValueError("duplicate barcode")
        if any(set(b) - set("ACGT") for b in self.barcodes):
            raise ValueError("barcodes must consist of A, C, G, T")
        if self.m + self.k > abi.CS_DEMUX_MAX_LONG:
            raise ValueError(f"barcode length + allowed errors must not exceed {abi.CS_DEMUX_MAX_LONG}")

    @property
    def m(self) -> int:
        return len(self.barcodes[0])

    @property
    def k(self) -> int:
        return int(self.max_error_rate * self.m)

    @property
    def tabulated(self) -> bool:
        """True: the op is a look-up table over every prefix of m + k bases (cs_plan_set_demux); False: longer
        barcodes, the op carries one PrefixAdapter op per barcode (cs_plan_set_demux_ops)."""
        return self.m + self.k <= abi.CS_DEMUX_MAX_PREFIX and not self.by_ops and not self.at_end

    def barcode_ops(self):
        """The barcodes' own adapter ops, as single-barcode plans hold them (cutseq/run.py:357-362, 592-597)."""
        make = suffix if self.at_end else prefix
        return [make(code, self.max_error_rate, self.match_flag, required=self.required) for code in self.barcodes]

    def __repr__(self):
        return (f"Demultiplexer({len(self.barcodes)} x PrefixAdapter(length={self.m}, "
                f"max_error_rate={self.max_error_rate}))")


Op = Union[AdapterOp, CutOp, QTrimOp, DemuxOp]


@dataclass
class MateChain:
    ops: List[Op] = field(default_factory=list)
    name_suffixes: Sequence[str] = ()  # SuffixRemover literals, applied in order
    rename_at: Optional[int] = None  # ops in front of the (PairedEnd)Renamer in the reference's modifier list

    def describe(self) -> List[str]:
        lines = [f"SuffixRemover({s!r})" for s in self.name_suffixes]
        return lines + [repr(o) for o in self.ops]


@dataclass
class TrimPlan:
    """Everything the engine needs for one run."""

    r1: MateChain
    r2: Optional[MateChain]
    has_umi: bool  # Renamer template carries the captured bases
    min_length: int
    untrimmed_filter: bool  # IsUntrimmedAny filter step installed
    swap_outputs: bool = False  # paired --auto-rc on a '-' library: R1/R2 output files swapped
    reverse_complement: bool = False  # single-end --auto-rc on a '-' library
    select_rule: int = abi.CS_SELECT_LEFTMOST
    use_filter: bool = True
    case_rule: int = abi.CS_CASE_FOLD
    indel_tie: int = abi.CS_TIE_INSERTION

    @property
    def paired(self) -> bool:
        return self.r2 is not None

    @property
    def needs_cap2(self) -> bool:
        return any(isinstance(o, CutOp) and o.capture == 2 for o in self.r1.ops)

    @property
    def demux(self) -> Optional[DemuxOp]:
        """The plan's demultiplexing op, if it has one (mate 1: 5' inline barcode; mate 2: 3' inline barcode)."""
        return next((o for _m, _i, o in self.demux_ops()), None)

    @property
    def demux_mate(self) -> int:
        """1 or 2: the mate whose chain holds the demultiplexing op (0 without one)."""
        return next((m for m, _i, _o in self.demux_ops()), 0)

    def demux_ops(self):
        """(mate, op index, op) of every demultiplexing op."""
        for mate, chain in ((1, self.r1), (2, self.r2)):
            if chain is None:
                continue
            for i, o in enumerate(chain.ops):
                if isinstance(o, DemuxOp):
                    yield mate, i, o

    def params(self) -> abi.cs_params:
        p = abi.cs_params()
        p.abi_version = abi.CS_ABI_VERSION
        p.min_length = max(0, min(int(self.min_length), 0xFFFF))
        p.select_rule = self.select_rule
        p.use_filter = 1 if self.use_filter else 0
        p.case_rule = self.case_rule
        p.indel_tie = self.indel_tie
        return p

    def pack(self):
        """-> (cs_op array r1, n1, cs_op array r2 | None, n2)."""
        a1 = pack_ops(self.r1.ops)
        a2 = pack_ops(self.r2.ops) if self.r2 is not None else None
        return a1, len(self.r1.ops), a2, (len(self.r2.ops) if self.r2 is not None else 0)


def pack_ops(ops: Sequence[Op], limit: int = abi.CS_MAX_OPS):
    if len(ops) > limit:
        raise ValueError(f"op chain of {len(ops)} exceeds CS_MAX_OPS={limit}")
    arr = (abi.cs_op * max(1, len(ops)))()
    for i, op in enumerate(ops):
        c = arr[i]
        c.stat_slot = i % abi.CS_MAX_OPS
        if isinstance(op, AdapterOp):
            c.kind = abi.CS_OP_ADAPTER
            c.align_flags = op.where
            c.reversed = 1 if op.rightmost else 0
            c.remove = op.remove
            c.shortcut = op.shortcut
            c.match_flag = op.match_flag
            c.required = 1 if op.required else 0
            c.m = op.m
            c.k = op.k
            c.min_overlap = op.min_overlap
            seq = op.sequence[::-1] if op.rightmost else op.sequence
            for j, ch in enumerate(seq.encode("ascii")):
                c.seq[j] = ch
            for j, t in enumerate(op.thresholds()):
                c.thr[j] = t
        elif isinstance(op, CutOp):
            c.kind = abi.CS_OP_CUT
            if not -0x7FFF <= op.length <= 0x7FFF:
                raise ValueError("cut length out of range")
            c.cut_len = op.length
            c.conditional = 1 if op.conditional else 0
            c.force_min_len = max(0, min(int(op.force_min_len), 0xFFFF))
            c.capture = op.capture
        elif isinstance(op, DemuxOp):
            c.kind = abi.CS_OP_DEMUX
            c.shortcut = abi.CS_DEMUX_BY_OPS if op.by_ops else 0
            c.reversed = 1 if op.at_end else 0
            c.m, c.k = op.m, op.k
            c.match_flag = op.match_flag
            c.required = 1 if op.required else 0
        elif isinstance(op, QTrimOp):
            c.kind = abi.CS_OP_QTRIM
            c.q_cutoff = max(-0x8000, min(int(op.cutoff_back), 0x7FFF))
            c.q_base = op.base
        else:  # pragma: no cover
            raise TypeError(op)
    return arr


# ---- adapter factories, one per cutadapt class the reference uses (run.py:17-24) ----


def rightmost_front(seq, rate, min_overlap, flag=0, shortcut=abi.CS_SHORTCUT_NONE):
    return AdapterOp("RightmostFrontAdapter", seq, rate, min_overlap, abi.CS_WHERE_BACK,
                     abi.CS_REMOVE_BEFORE, rightmost=True, shortcut=shortcut, match_flag=flag)


def back(seq, rate, min_overlap, force_anywhere=False, flag=0, shortcut=abi.CS_SHORTCUT_NONE):
    where = abi.CS_WHERE_ANYWHERE if force_anywhere else abi.CS_WHERE_BACK
    return AdapterOp("BackAdapter", seq, rate, min_overlap, where, abi.CS_REMOVE_AFTER,
                     shortcut=shortcut, match_flag=flag)


def prefix(seq, rate, flag=0, required=False):
    return AdapterOp("PrefixAdapter", seq, rate, len(seq), abi.CS_WHERE_PREFIX, abi.CS_REMOVE_BEFORE,
                     match_flag=flag, required=required)


def suffix(seq, rate, flag=0, required=False):
    return AdapterOp("SuffixAdapter", seq, rate, len(seq), abi.CS_WHERE_SUFFIX, abi.CS_REMOVE_AFTER,
                     match_flag=flag, required=required)


def non_internal_back(seq, rate, flag=0):
    return AdapterOp("NonInternalBackAdapter", seq, rate, 3, abi.CS_WHERE_BACK_NOT_INTERNAL,
                     abi.CS_REMOVE_AFTER, match_flag=flag)


def non_internal_front(seq, rate, flag=0):
    return AdapterOp("NonInternalFrontAdapter", seq, rate, 3, abi.CS_WHERE_FRONT_NOT_INTERNAL,
                     abi.CS_REMOVE_BEFORE, match_flag=flag)


def _poly_a():
    return non_internal_back("A" * POLY_LENGTH, POLY_MAX_ERRORS, abi.CS_F_POLY)


def _poly_t():
    return non_internal_front("T" * POLY_LENGTH, POLY_MAX_ERRORS, abi.CS_F_POLY)


def _demux_op(barcode: BarcodeConfig, settings, paired: bool):
    """-> (op for the 5' inline barcode | None, op for the 3' inline barcode | None): the scheme's 5' inline barcode is
    the one the barcode list stands for when there is one, else the 3' one -- which starts R2, reverse-complemented
    (cutseq/run.py:604-608), so that is where its op goes."""
    codes = getattr(settings, "demux_barcodes", None)
    if not codes:
        return None, None
    if barcode.inline5.len == 0 and barcode.inline3.len == 0:
        raise ValueError("demultiplexing needs a scheme with an inline barcode, e.g. P5(ATCACG)NNNN>P7")
    at5 = barcode.inline5.len > 0
    inline = barcode.inline5 if at5 else barcode.inline3
    if not at5 and paired:
        from .common import reverse_complement
        codes = [reverse_complement(c.upper().replace("U", "T")) for c in codes]
    # (single-end reads and a 3' barcode: it ends the read once the 3' adapter is gone -- SuffixAdapter ops, run.py:364-370)
    op = DemuxOp(list(codes), MAX_ERRORS, abi.CS_F_INLINE, required=True, at_end=not at5 and not paired)
    if op.m != inline.len:
        raise ValueError(f"the barcodes are {op.m} nt long, the scheme's inline barcode {inline.len}")
    return (op, None) if at5 else (None, op)


def compile_single(barcode: BarcodeConfig, settings: CutadaptConfig, untrimmed_requested: bool = False) -> TrimPlan:
    """Single-end chain, step for step as cutseq/run.py:326-426."""
    untrimmed_filter = (
        barcode.inline5.len + barcode.inline3.len > 0 and settings.ensure_inline_barcode
    ) or untrimmed_requested  # run.py:453-456
    ops: List[Op] = []
    # step 2 / 3: 5' template-switch artefact, 3' read-through
    sc = getattr(settings, "shortcut", abi.CS_SHORTCUT_NONE)
    ops.append(rightmost_front(barcode.p5.fw, MAX_ERRORS, 10, abi.CS_F_ADAPTER5, sc))
    ops.append(back(barcode.p7.fw, MAX_ERRORS, 3, settings.force_anywhere, abi.CS_F_ADAPTER3, sc))
    # step 4: inline barcodes
    demux, demux3 = _demux_op(barcode, settings, paired=False)
    if demux is not None or demux3 is not None:
        untrimmed_filter = True  # a read without any of the barcodes goes where --ensure-inline-barcode sends it
    if demux is not None:
        ops.append(demux)
    elif barcode.inline5.len > 0:
        ops.append(prefix(barcode.inline5.fw, MAX_ERRORS, abi.CS_F_INLINE, required=untrimmed_filter))
    if barcode.inline3.len > 0:
        ops.append(demux3 if demux3 is not None else
                   suffix(barcode.inline3.fw, MAX_ERRORS, abi.CS_F_INLINE, required=untrimmed_filter))
    # step 5: UMI (always unconditional in single-end mode), then rename
    cap = 0
    if barcode.umi5.len > 0:
        cap += 1
        ops.append(CutOp(barcode.umi5.len, capture=cap))
    if barcode.umi3.len > 0:
        cap += 1
        ops.append(CutOp(-barcode.umi3.len, capture=cap))
    rename_at = len(ops)  # the Renamer sits here in the reference's list (run.py:377-380)
    # step 6: masks
    if barcode.mask5.len > 0:
        ops.append(CutOp(barcode.mask5.len))
    if barcode.mask3.len > 0:
        ops.append(CutOp(-barcode.mask3.len))
    # step 7: poly-A
    if settings.trim_polyA:
        if settings.trim_polyA_wo_direction:
            ops += [_poly_a(), _poly_t()]
        elif barcode.strand == "+":
            ops.append(_poly_a())
        elif barcode.strand == "-":
            ops.append(_poly_t())
        else:
            logging.info("No strand information provided, skip polyA trimming.")
    # step 8
    ops.append(QTrimOp(settings.min_quality))
    # step 9
    rc = False
    if settings.auto_rc:
        if barcode.strand == "-":
            rc = True
        else:
            logging.warning("Library is not (-) strand, but --auto-rc is enabled. Ignored.")
    return TrimPlan(
        r1=MateChain(ops, (".1", "/1"), rename_at),
        r2=None,
        has_umi=barcode.umi5.len + barcode.umi3.len > 0,
        min_length=settings.min_length,
        untrimmed_filter=untrimmed_filter,
        reverse_complement=rc,
        select_rule=getattr(settings, "select_rule", abi.CS_SELECT_LEFTMOST),
        case_rule=getattr(settings, "case_rule", abi.CS_CASE_FOLD),
        indel_tie=getattr(settings, "indel_tie", abi.CS_TIE_INSERTION),
    )


def compile_paired(barcode: BarcodeConfig, settings: CutadaptConfig, untrimmed_requested: bool = False) -> TrimPlan:
    """Paired-end chain, step for step as cutseq/run.py:533-731."""
    untrimmed_filter = (
        barcode.inline5.len + barcode.inline3.len > 0 and settings.ensure_inline_barcode
    ) or untrimmed_requested  # run.py:771-774
    fmin = settings.force_trim_min_length
    cond = bool(settings.conditional_cutter)
    o1: List[Op] = []
    o2: List[Op] = []
    # step 2
    sc = getattr(settings, "shortcut", abi.CS_SHORTCUT_NONE)
    o1.append(rightmost_front(barcode.p5.fw, MAX_ERRORS, 10, abi.CS_F_ADAPTER5, sc))
    o2.append(rightmost_front(barcode.p7.rc, MAX_ERRORS, 10, abi.CS_F_ADAPTER5, sc))
    # step 3
    o1.append(back(barcode.p7.fw, MAX_ERRORS, 3, settings.force_anywhere, abi.CS_F_ADAPTER3, sc))
    o2.append(back(barcode.p5.rc, MAX_ERRORS, 3, settings.force_anywhere, abi.CS_F_ADAPTER3, sc))
    # step 4
    demux, demux3 = _demux_op(barcode, settings, paired=True)
    if demux is not None or demux3 is not None:
        untrimmed_filter = True
    if demux is not None:
        o1.append(demux)
        o2.append(CutOp(-barcode.inline5.len))
    elif barcode.inline5.len > 0:
        o1.append(prefix(barcode.inline5.fw, MAX_ERRORS, abi.CS_F_INLINE, required=untrimmed_filter))
        o2.append(CutOp(-barcode.inline5.len))
    if barcode.inline3.len > 0:
        o1.append(CutOp(-barcode.inline3.len))
        o2.append(demux3 if demux3 is not None else
                  prefix(barcode.inline3.rc, MAX_ERRORS, abi.CS_F_INLINE, required=untrimmed_filter))
    # step 5: the mate that starts with the UMI always loses it; the other mate only on read-through
    if barcode.umi5.len > 0:
        o1.append(CutOp(barcode.umi5.len, capture=1))
        o2.append(CutOp(-barcode.umi5.len, conditional=cond, force_min_len=fmin))
    if barcode.umi3.len > 0:
        o1.append(CutOp(-barcode.umi3.len, conditional=cond, force_min_len=fmin))
        o2.append(CutOp(barcode.umi3.len, capture=1))
    rename_at = len(o1)  # the PairedEndRenamer sits here in the reference's list (run.py:642-645)
    # step 6
    if barcode.mask5.len > 0:
        o1.append(CutOp(barcode.mask5.len))
        o2.append(CutOp(-barcode.mask5.len, conditional=cond, force_min_len=fmin))
    if barcode.mask3.len > 0:
        o1.append(CutOp(-barcode.mask3.len, conditional=cond, force_min_len=fmin))
        o2.append(CutOp(barcode.mask3.len))
    # step 7
    if settings.trim_polyA:
        if settings.trim_polyA_wo_direction:
            o1 += [_poly_a(), _poly_t()]
            o2 += [_poly_t(), _poly_a()]
        elif barcode.strand == "+":
            o1.append(_poly_a())
            o2.append(_poly_t())
        elif barcode.strand == "-":
            o1.append(_poly_t())
            o2.append(_poly_a())
        else:
            logging.info("No strand information provided, skip polyA trimming.")
    # step 8
    o1.append(QTrimOp(settings.min_quality))
    o2.append(QTrimOp(settings.min_quality))
    # step 9: no modifier in paired mode, only the output files swap
    swap = False
    if settings.auto_rc:
        if barcode.strand == "-":
            swap = True
        else:
            logging.warning("Library is not (-) strand, but --auto-rc is enabled. Ignored.")
    return TrimPlan(
        r1=MateChain(o1, (".1", "/1"), rename_at),
        r2=MateChain(o2, (".2", "/2"), rename_at),
        has_umi=barcode.umi5.len + barcode.umi3.len > 0,
        min_length=settings.min_length,
        untrimmed_filter=untrimmed_filter,
        swap_outputs=swap,
        select_rule=getattr(settings, "select_rule", abi.CS_SELECT_LEFTMOST),
        case_rule=getattr(settings, "case_rule", abi.CS_CASE_FOLD),
        indel_tie=getattr(settings, "indel_tie", abi.CS_TIE_INSERTION),
    )


def single_adapter_plan(sequence: str, max_error_rate: float = 0.1, min_overlap: int = 3,
                        min_length: int = 0, **kw) -> TrimPlan:
    """BASELINE.json config 2: one regular 3' adapter, nothing else (not expressible in the
    reference CLI, exercised through the op table directly)."""
    op = back(sequence, max_error_rate, min_overlap, flag=abi.CS_F_ADAPTER3)
    return TrimPlan(r1=MateChain([op]), r2=None, has_umi=False, min_length=min_length,
                    untrimmed_filter=False, **kw)
