"""Pure-Python restatement of the reference path, record in -> record out.

TEST INFRASTRUCTURE ONLY (second, independent checker next to ``cutseq_oracle.c``; small
inputs only).  PARITY UNPINNED: cutadapt (``cutadapt~=5.0``, reference pyproject.toml:17)
is absent here, so the classes below restate cutadapt 5.x from knowledge of its source:

* :class:`Aligner`            <- cutadapt/_align.pyx  ``Aligner.locate``
* ``*Adapter.match_to``       <- cutadapt/adapters.py
* :func:`quality_trim_index`  <- cutadapt/qualtrim.pyx
* cutters / renamers          <- cutadapt/modifiers.py
* filters                     <- cutadapt/steps.py, predicates.py

and :func:`build_single` / :func:`build_paired` restate how the reference wires them
(cutseq/run.py:326-426 and 533-731, ``ConditionalCutter`` run.py:145-161,
``IsUntrimmedAny`` run.py:97-110).  Unlike the C oracle and the GPU kernel, which work
on intervals of the original record, this one really slices strings the way cutadapt
does, and evaluates ``cost <= length * max_error_rate`` in floating point.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

REF_START, QUERY_START, REF_END, QUERY_STOP = 1, 2, 4, 8
BACK = QUERY_START | QUERY_STOP | REF_END
FRONT = QUERY_START | QUERY_STOP | REF_START
PREFIX = QUERY_STOP
SUFFIX = QUERY_START
FRONT_NOT_INTERNAL = REF_START | QUERY_STOP
BACK_NOT_INTERNAL = QUERY_START | REF_END
ANYWHERE = 15

SELECT_LEFTMOST, SELECT_SCORE = 0, 1
TIE_INSERTION, TIE_DELETION = 0, 1
CASE_FOLD, CASE_SENSITIVE = 0, 1
SHORTCUT_NONE, SHORTCUT_FIND = 0, 1


class Aligner:
    """Semi-global unit-cost DP with (cost, score, origin) cells and an error-rate budget."""

    def __init__(self, reference: str, max_error_rate: float, flags: int, min_overlap: int = 1,
                 select_rule: int = SELECT_LEFTMOST, indel_tie: int = TIE_INSERTION):
        self.indel_tie = indel_tie
        self.reference = reference
        self.m = len(reference)
        self.max_error_rate = max_error_rate
        self.flags = flags
        self.min_overlap = min_overlap
        self.select_rule = select_rule

    def _better(self, best, cand, best_length, length, n) -> bool:
        m = self.m
        b_origin, b_cost, b_score = best[0], best[1], best[2]
        origin, cost, score = cand
        if b_cost == m + n + 1:
            return True
        if self.select_rule == SELECT_SCORE:
            return score > b_score or (score == b_score and cost < b_cost)
        return (origin <= b_origin + m // 2 and score > b_score) or (length > best_length and score > b_score)

    def locate(self, query: str) -> Optional[Tuple[int, int, int, int, int, int]]:
        ref, m, n = self.reference, self.m, len(query)
        rate = self.max_error_rate
        start_in_ref = bool(self.flags & REF_START)
        start_in_query = bool(self.flags & QUERY_START)
        stop_in_ref = bool(self.flags & REF_END)
        stop_in_query = bool(self.flags & QUERY_STOP)
        k = int(rate * m)
        max_n, min_n = n, 0
        if not start_in_query:
            max_n = min(n, m + k)
        if not stop_in_query:
            min_n = max(0, n - m - k)
        cost = [0] * (m + 1)
        score = [0] * (m + 1)
        origin = [0] * (m + 1)
        for i in range(m + 1):
            if not start_in_ref and not start_in_query:
                cost[i], origin[i] = max(i, min_n), 0
            elif start_in_ref and not start_in_query:
                cost[i], origin[i] = min_n, min(0, min_n - i)
            elif not start_in_ref and start_in_query:
                cost[i], origin[i] = i, max(0, min_n - i)
            else:
                cost[i], origin[i] = min(i, min_n), min_n - i
        # best = [origin, cost, score, ref_stop, query_stop]
        best = [0, m + n + 1, 0, m, n]
        last = m if start_in_ref else min(m, k + 1)
        for j in range(min_n + 1, max_n + 1):
            d_cost, d_score, d_origin = cost[0], score[0], origin[0]
            if start_in_query:
                origin[0] = j
            else:
                cost[0] = j
            qc = query[j - 1]
            for i in range(1, last + 1):
                if ref[i - 1] == qc:
                    c, o, s = d_cost, d_origin, d_score + 1
                else:
                    c_diag, c_del, c_ins = d_cost + 1, cost[i] + 1, cost[i - 1] + 1
                    if c_diag <= c_del and c_diag <= c_ins:
                        c, o, s = c_diag, d_origin, d_score - 1
                    elif (c_ins <= c_del) if self.indel_tie == TIE_INSERTION else (c_ins < c_del):
                        c, o, s = c_ins, origin[i - 1], score[i - 1] - 2
                    else:
                        c, o, s = c_del, origin[i], score[i] - 2
                d_cost, d_score, d_origin = cost[i], score[i], origin[i]
                cost[i], origin[i], score[i] = c, o, s
            while last >= 0 and cost[last] > k:
                last -= 1
            if last < m:
                last += 1
            elif stop_in_query:
                c, s, o = cost[m], score[m], origin[m]
                length = m + min(o, 0)
                ok = length >= self.min_overlap and c <= length * rate
                best_length = m + min(best[0], 0)
                if ok and self._better(best, (o, c, s), best_length, length, n):
                    best = [o, c, s, m, j]
                    if c == 0 and o >= 0:
                        break
        if max_n == n:
            first_i = 0 if stop_in_ref else m
            for i in range(m, first_i - 1, -1):
                c, s, o = cost[i], score[i], origin[i]
                length = i + min(o, 0)
                ok = length >= self.min_overlap and c <= length * rate
                best_length = best[3] + min(best[0], 0)
                if ok and self._better(best, (o, c, s), best_length, length, n):
                    best = [o, c, s, i, n]
        if best[1] == m + n + 1:
            return None
        if best[0] >= 0:
            ref_start, query_start = 0, best[0]
        else:
            ref_start, query_start = -best[0], 0
        return (ref_start, best[3], query_start, best[4], best[2], best[1])


@dataclass
class Match:
    astart: int
    astop: int
    rstart: int
    rstop: int
    score: int
    errors: int
    adapter: object
    remove_before: bool

    def trimmed(self, read: "Read") -> "Read":
        return read[self.rstop:] if self.remove_before else read[: self.rstart]


class SingleAdapter:
    where = BACK
    remove_before = False

    def __init__(self, sequence: str, max_errors: float = 0.1, min_overlap: int = 3,
                 select_rule: int = SELECT_LEFTMOST, indel_tie: int = TIE_INSERTION,
                 case_rule: int = CASE_FOLD, shortcut: int = SHORTCUT_NONE):
        self.indel_tie, self.case_rule, self.shortcut = indel_tie, case_rule, shortcut
        self.sequence = sequence.upper().replace("U", "T")
        if not self.sequence:
            raise ValueError("Adapter sequence is empty")
        if max_errors >= 1:
            max_errors /= len(self.sequence)
        self.max_error_rate = max_errors
        self.min_overlap = min(min_overlap, len(self.sequence))
        self.select_rule = select_rule
        self.aligner = self._aligner()

    def _aligner(self):
        return Aligner(self.sequence, self.max_error_rate, self.where, self.min_overlap, self.select_rule,
                       self.indel_tie)

    def _seen(self, sequence: str) -> str:
        """The read as the aligner sees it: ``sequence.upper()`` (cutadapt ``match_to``)."""
        return sequence.upper() if self.case_rule == CASE_FOLD else sequence

    def _wrap(self, alignment):
        if alignment is None:
            return None
        return Match(*alignment, adapter=self, remove_before=self.remove_before)

    def match_to(self, sequence: str):
        return self._wrap(self.aligner.locate(self._seen(sequence)))


class BackAdapter(SingleAdapter):
    def __init__(self, sequence, max_errors=0.1, min_overlap=3, force_anywhere=False, **kw):
        self.where = ANYWHERE if force_anywhere else BACK
        super().__init__(sequence, max_errors, min_overlap, **kw)

    def match_to(self, sequence: str):
        sequence = self._seen(sequence)
        if self.shortcut == SHORTCUT_FIND:  # cutadapt <= 2.x: exact occurrence first (opt-in)
            pos = sequence.find(self.sequence)
            if pos >= 0:
                m = len(self.sequence)
                return self._wrap((0, m, pos, pos + m, m, 0))
        return self._wrap(self.aligner.locate(sequence))


class FrontAdapter(SingleAdapter):
    """Regular 5' adapter (``-g``); cutseq only uses the rightmost variant, the guide tables use this one."""
    where = FRONT
    remove_before = True


class RightmostFrontAdapter(SingleAdapter):
    remove_before = True

    def _aligner(self):
        return Aligner(self.sequence[::-1], self.max_error_rate, BACK, self.min_overlap, self.select_rule,
                       self.indel_tie)

    def match_to(self, sequence: str):
        sequence = self._seen(sequence)
        m, n = len(self.sequence), len(sequence)
        if self.shortcut == SHORTCUT_FIND:
            pos = sequence.rfind(self.sequence)
            if pos >= 0:
                return self._wrap((0, m, pos, pos + m, m, 0))
        aln = self.aligner.locate(sequence[::-1])
        if aln is None:
            return None
        ref_start, ref_end, q_start, q_end, score, errors = aln
        return self._wrap((m - ref_end, m - ref_start, n - q_end, n - q_start, score, errors))


class NonInternalFrontAdapter(SingleAdapter):
    where = FRONT_NOT_INTERNAL
    remove_before = True


class NonInternalBackAdapter(SingleAdapter):
    where = BACK_NOT_INTERNAL


class PrefixAdapter(SingleAdapter):
    where = PREFIX
    remove_before = True

    def __init__(self, sequence, max_errors=0.1, **kw):
        kw.pop("shortcut", None)
        super().__init__(sequence, max_errors, len(sequence), **kw)


class SuffixAdapter(SingleAdapter):
    where = SUFFIX

    def __init__(self, sequence, max_errors=0.1, **kw):
        kw.pop("shortcut", None)
        super().__init__(sequence, max_errors, len(sequence), **kw)


@dataclass
class Read:
    name: str
    sequence: str
    qualities: str

    def __len__(self):
        return len(self.sequence)

    def __getitem__(self, key):
        return Read(self.name, self.sequence[key], self.qualities[key])

    def fastq(self) -> str:
        return f"@{self.name}\n{self.sequence}\n+\n{self.qualities}\n"


@dataclass
class ModificationInfo:
    matches: List[Match] = field(default_factory=list)
    cut_prefix: Optional[str] = None
    cut_suffix: Optional[str] = None


class SuffixRemover:
    def __init__(self, suffix):
        self.suffix = suffix

    def __call__(self, read, info):
        if read.name.endswith(self.suffix):
            read = Read(read.name[: -len(self.suffix)], read.sequence, read.qualities)
        return read


class AdapterCutter:
    def __init__(self, adapter):
        self.adapter = adapter
        self.with_adapters = 0

    def __call__(self, read, info):
        match = self.adapter.match_to(read.sequence)
        if match is None:
            return read
        self.with_adapters += 1
        info.matches.append(match)
        return match.trimmed(read)


class UnconditionalCutter:
    def __init__(self, length):
        self.length = length

    def __call__(self, read, info):
        if self.length > 0:
            info.cut_prefix = read.sequence[: self.length]
            return read[self.length:]
        if self.length < 0:
            info.cut_suffix = read.sequence[self.length:]
            return read[: self.length]
        return read


class ConditionalCutter(UnconditionalCutter):
    """cutseq/run.py:113-161."""

    def __init__(self, length, force_trim_min_length=50):
        super().__init__(length)
        self.force_trim_min_length = force_trim_min_length

    def __call__(self, read, info):
        if not info.matches and len(read.sequence) < self.force_trim_min_length:
            return read
        return super().__call__(read, info)


def quality_trim_index(qualities: str, cutoff_front: int, cutoff_back: int, base: int = 33):
    start, stop = 0, len(qualities)
    s = max_qual = 0
    for i, ch in enumerate(qualities):
        s += cutoff_front - (ord(ch) - base)
        if s < 0:
            break
        if s > max_qual:
            max_qual, start = s, i + 1
    s = max_qual = 0
    for i in range(len(qualities) - 1, -1, -1):
        s += cutoff_back - (ord(qualities[i]) - base)
        if s < 0:
            break
        if s > max_qual:
            max_qual, stop = s, i
    if start >= stop:
        start = stop = 0
    return start, stop


class QualityTrimmer:
    def __init__(self, cutoff_front, cutoff_back, base=33):
        self.cutoff_front, self.cutoff_back, self.base = cutoff_front, cutoff_back, base
        self.trimmed_bases = 0

    def __call__(self, read, info):
        start, stop = quality_trim_index(read.qualities, self.cutoff_front, self.cutoff_back, self.base)
        self.trimmed_bases += len(read) - (stop - start)
        return read[start:stop]


def parse_name(name: str) -> Tuple[str, str]:
    fields = name.split(maxsplit=1)
    if len(fields) == 2:
        return fields[0], fields[1]
    return name, ""


def _ids_match(name1: str, name2: str) -> bool:
    """dnaio.record_names_match: ids up to the first space/tab, a trailing 1/2/3 ignored."""
    def ident(name):
        cut = len(name)
        for sep in " \t":
            p = name.find(sep)
            if p >= 0:
                cut = min(cut, p)
        return name[:cut]

    a, b = ident(name1), ident(name2)
    if a and b and a[-1] in "123" and b[-1] in "123":
        a, b = a[:-1], b[:-1]
    return a == b


COMPLEMENT = str.maketrans("ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn")


def reverse_complement_read(read: Read) -> Read:
    return Read(read.name, read.sequence.translate(COMPLEMENT)[::-1], read.qualities[::-1])


@dataclass
class Settings:
    ensure_inline_barcode: bool = False
    trim_polyA: bool = False
    trim_polyA_wo_direction: bool = False
    conditional_cutter: bool = True
    min_length: int = 20
    min_quality: int = 20
    auto_rc: bool = False
    force_trim_min_length: int = 50
    force_anywhere: bool = False
    select_rule: int = SELECT_LEFTMOST
    indel_tie: int = TIE_INSERTION
    case_rule: int = CASE_FOLD
    shortcut: int = SHORTCUT_NONE

    def rules(self) -> dict:
        return dict(select_rule=self.select_rule, indel_tie=self.indel_tie, case_rule=self.case_rule,
                    shortcut=self.shortcut)


class SinglePipeline:
    """pipeline_single restated: ``process(read)`` -> (route, Read)."""

    def __init__(self, bc, st: Settings, untrimmed_requested=False):
        e, rules = 0.2, st.rules()
        self.mods = [SuffixRemover(".1"), SuffixRemover("/1")]
        self.mods.append(AdapterCutter(RightmostFrontAdapter(bc.p5.fw, e, 10, **rules)))
        self.mods.append(AdapterCutter(BackAdapter(bc.p7.fw, e, 3, st.force_anywhere, **rules)))
        self.required = []
        if bc.inline5.len:
            a = PrefixAdapter(bc.inline5.fw, e, **rules)
            self.required.append(a)
            self.mods.append(AdapterCutter(a))
        if bc.inline3.len:
            a = SuffixAdapter(bc.inline3.fw, e, **rules)
            self.required.append(a)
            self.mods.append(AdapterCutter(a))
        if bc.umi5.len:
            self.mods.append(UnconditionalCutter(bc.umi5.len))
        if bc.umi3.len:
            self.mods.append(UnconditionalCutter(-bc.umi3.len))
        self.rename_at = len(self.mods)
        self.has_umi = bc.umi5.len + bc.umi3.len > 0
        if bc.mask5.len:
            self.mods.append(UnconditionalCutter(bc.mask5.len))
        if bc.mask3.len:
            self.mods.append(UnconditionalCutter(-bc.mask3.len))
        if st.trim_polyA:
            fwd = lambda: AdapterCutter(NonInternalBackAdapter("A" * 100, 0.15, **rules))
            rev = lambda: AdapterCutter(NonInternalFrontAdapter("T" * 100, 0.15, **rules))
            if st.trim_polyA_wo_direction:
                self.mods += [fwd(), rev()]
            elif bc.strand == "+":
                self.mods.append(fwd())
            elif bc.strand == "-":
                self.mods.append(rev())
        self.mods.append(QualityTrimmer(0, st.min_quality))
        self.rc = st.auto_rc and bc.strand == "-"
        self.min_length = st.min_length
        self.untrimmed_filter = (bc.inline5.len + bc.inline3.len > 0 and st.ensure_inline_barcode) or untrimmed_requested

    def process(self, read: Read):
        info = ModificationInfo()
        for idx, mod in enumerate(self.mods):
            if idx == self.rename_at:
                read = self._rename(read, info)
            read = mod(read, info)
        if self.rename_at >= len(self.mods):
            read = self._rename(read, info)
        if self.rc:
            read = reverse_complement_read(read)
        if len(read) < self.min_length:
            return "short", read
        if self.untrimmed_filter:
            matched = [mt.adapter for mt in info.matches]
            if any(a not in matched for a in self.required):
                return "untrimmed", read
        return "trimmed", read

    def _rename(self, read, info):
        rid, _ = parse_name(read.name)
        if self.has_umi:
            rid = f"{rid}_{info.cut_prefix or ''}{info.cut_suffix or ''}"
        return Read(rid, read.sequence, read.qualities)


class PairedPipeline:
    """pipeline_paired restated: ``process(r1, r2)`` -> (route, Read, Read)."""

    def __init__(self, bc, st: Settings, untrimmed_requested=False):
        e, rules, f = 0.2, st.rules(), st.force_trim_min_length

        def cond(n):
            return ConditionalCutter(n, f) if st.conditional_cutter else UnconditionalCutter(n)

        mods = [(SuffixRemover(".1"), SuffixRemover(".2")), (SuffixRemover("/1"), SuffixRemover("/2"))]
        mods.append((AdapterCutter(RightmostFrontAdapter(bc.p5.fw, e, 10, **rules)),
                     AdapterCutter(RightmostFrontAdapter(bc.p7.rc, e, 10, **rules))))
        mods.append((AdapterCutter(BackAdapter(bc.p7.fw, e, 3, st.force_anywhere, **rules)),
                     AdapterCutter(BackAdapter(bc.p5.rc, e, 3, st.force_anywhere, **rules))))
        self.req1, self.req2 = [], []
        if bc.inline5.len:
            a = PrefixAdapter(bc.inline5.fw, e, **rules)
            self.req1.append(a)
            mods.append((AdapterCutter(a), UnconditionalCutter(-bc.inline5.len)))
        if bc.inline3.len:
            a = PrefixAdapter(bc.inline3.rc, e, **rules)
            self.req2.append(a)
            mods.append((UnconditionalCutter(-bc.inline3.len), AdapterCutter(a)))
        if bc.umi5.len:
            mods.append((UnconditionalCutter(bc.umi5.len), cond(-bc.umi5.len)))
        if bc.umi3.len:
            mods.append((cond(-bc.umi3.len), UnconditionalCutter(bc.umi3.len)))
        self.rename_at = len(mods)
        self.has_umi = bc.umi5.len + bc.umi3.len > 0
        if bc.mask5.len:
            mods.append((UnconditionalCutter(bc.mask5.len), cond(-bc.mask5.len)))
        if bc.mask3.len:
            mods.append((cond(-bc.mask3.len), UnconditionalCutter(bc.mask3.len)))
        if st.trim_polyA:
            pa = lambda: AdapterCutter(NonInternalBackAdapter("A" * 100, 0.15, **rules))
            pt = lambda: AdapterCutter(NonInternalFrontAdapter("T" * 100, 0.15, **rules))
            if st.trim_polyA_wo_direction:
                mods += [(pa(), pt()), (pt(), pa())]
            elif bc.strand == "+":
                mods.append((pa(), pt()))
            elif bc.strand == "-":
                mods.append((pt(), pa()))
        mods.append((QualityTrimmer(0, st.min_quality), QualityTrimmer(0, st.min_quality)))
        self.mods = mods
        self.swap = st.auto_rc and bc.strand == "-"
        self.min_length = st.min_length
        self.untrimmed_filter = (bc.inline5.len + bc.inline3.len > 0 and st.ensure_inline_barcode) or untrimmed_requested

    def process(self, r1: Read, r2: Read):
        i1, i2 = ModificationInfo(), ModificationInfo()
        for idx, (m1, m2) in enumerate(self.mods):
            if idx == self.rename_at:
                r1, r2 = self._rename(r1, r2, i1, i2)
            r1, r2 = m1(r1, i1), m2(r2, i2)
        if len(r1) < self.min_length or len(r2) < self.min_length:
            return "short", r1, r2
        if self.untrimmed_filter:
            def missing(req, info):
                got = [mt.adapter for mt in info.matches]
                return any(a not in got for a in req)
            if missing(self.req1, i1) or missing(self.req2, i2):
                return "untrimmed", r1, r2
        return "trimmed", r1, r2

    def _rename(self, r1, r2, i1, i2):
        id1, _ = parse_name(r1.name)
        id2, _ = parse_name(r2.name)
        if not _ids_match(r1.name, r2.name):
            raise ValueError(f"Input read IDs not identical: '{id1}' != '{id2}'")
        if self.has_umi:
            tag = f"_{i1.cut_prefix or ''}{i2.cut_prefix or ''}"
            id1, id2 = id1 + tag, id2 + tag
        return Read(id1, r1.sequence, r1.qualities), Read(id2, r2.sequence, r2.qualities)
