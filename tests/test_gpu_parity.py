"""Parity of the HIP path against the CPU oracle, bit-exact, through the C ABI.

Every test here needs a real MI355X (``-m gpu``).  Results (8-byte records), second
captures and the device statistics block must equal the oracle's on the same inputs.
"""
import random

import numpy as np
import pytest

import oracle
from cutseq_amd import abi, plan as planmod, synth
import hostfmt
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine
from cutseq_amd.synth import SynthBatch

import util
from test_oracle import CHAIN_CASES, WHERE

pytestmark = pytest.mark.gpu


def stats_dict(st):
    d = st.as_dict()
    d.pop("n_exact_dp", None)  # diagnostic: the oracle has no notion of the pre-filter
    d.pop("n_refiltered", None)  # ... nor of the existence-only scan
    return d


def run_both(tp: planmod.TrimPlan, batch: SynthBatch, threads: int = 8):
    """-> None; asserts device == oracle for results, cap2 and stats."""
    (o1, ocap2, ost1), m2 = util.oracle_run(tp, batch, threads=threads)
    with TrimEngine(tp, device=0, slots=1, max_reads=max(batch.n, 1), max_stride=batch.stride) as eng:
        g1, gcap2, g2 = eng.trim(batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2)
        gst1, gst2 = eng.stats()
    bad = np.nonzero(g1 != o1)[0]
    assert bad.size == 0, (bad[:5], g1[bad[:5]], o1[bad[:5]],
                           [util.row_bytes(batch.seq1, batch.len1, int(i)) for i in bad[:2]])
    if ocap2 is not None:
        assert (gcap2 == ocap2).all()
    assert stats_dict(gst1) == stats_dict(ost1)
    if m2 is not None:
        o2, _, ost2 = m2
        bad = np.nonzero(g2 != o2)[0]
        assert bad.size == 0, (bad[:5], g2[bad[:5]], o2[bad[:5]],
                               [util.row_bytes(batch.seq2, batch.len2, int(i)) for i in bad[:2]])
        assert stats_dict(gst2) == stats_dict(ost2)
    return g1, g2


def adversarial_reads(rng: random.Random, ref: str, count: int, alpha: str, max_len: int = 70):
    reads = []
    for _ in range(count):
        style = rng.random()
        if style < 0.3:
            s = util.random_dna(rng, rng.randint(0, max_len), alpha)
        else:
            core = util.mutate(rng, ref, rng.randint(0, 3), alpha)
            cut = rng.random()
            if cut < 0.3:
                core = core[: rng.randint(0, len(core))]
            elif cut < 0.6:
                core = core[rng.randint(0, len(core)):]
            s = util.random_dna(rng, rng.randint(0, 30), alpha) + core + util.random_dna(rng, rng.randint(0, 30), alpha)
            if rng.random() < 0.3:
                s += util.mutate(rng, ref, rng.randint(0, 2), alpha) + util.random_dna(rng, rng.randint(0, 8), alpha)
        s = s[:max_len + 60]
        reads.append((s, "I" * len(s)))
    return reads


def one_adapter_plan(seq, rate, min_overlap, where, remove, rightmost=False, shortcut=0, rule=0, use_filter=True,
                     tie=abi.CS_TIE_INSERTION, case=abi.CS_CASE_FOLD):
    op = planmod.AdapterOp("test", seq, rate, min_overlap, where, remove, rightmost=rightmost, shortcut=shortcut,
                           match_flag=abi.CS_F_ADAPTER3)
    return planmod.TrimPlan(r1=planmod.MateChain([op]), r2=None, has_umi=False, min_length=0,
                            untrimmed_filter=False, select_rule=rule, use_filter=use_filter, indel_tie=tie,
                            case_rule=case)


@pytest.mark.parametrize("use_filter", [True, False])
@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("where", sorted(WHERE))
def test_single_adapter_all_flag_sets(where, rule, use_filter):
    """Every cutadapt 'Where' the reference can instantiate (and ANYWHERE/FRONT), short
    adversarial reads over small alphabets, with and without the bit-parallel pre-filter."""
    rng = random.Random(hash((where, rule)) & 0xFFFF)
    for trial in range(12):
        alpha = rng.choice(["AC", "ACG", "ACGT", "ACGTN"])
        m = rng.choice([1, 2, 3, 5, 8, 13, 20, 31, 32])
        ref = util.random_dna(rng, m, alpha.replace("N", ""))
        rate = rng.choice([0.0, 0.1, 0.2, 0.34])
        mo = rng.randint(1, m)
        remove = rng.choice([abi.CS_REMOVE_BEFORE, abi.CS_REMOVE_AFTER])
        rightmost = rng.random() < 0.3
        shortcut = abi.CS_SHORTCUT_FIND if rng.random() < 0.3 else abi.CS_SHORTCUT_NONE
        tie = rng.choice([abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION])
        case = abi.CS_CASE_SENSITIVE if rng.random() < 0.25 else abi.CS_CASE_FOLD
        tp = one_adapter_plan(ref, rate, mo, WHERE[where], remove, rightmost, shortcut, rule, use_filter, tie, case)
        batch = util.batch_from_reads(adversarial_reads(rng, ref, 700, alpha))
        util.soft_mask(batch, 0.15, seed=trial)
        run_both(tp, batch, threads=4)


@pytest.mark.parametrize("m", [33, 48, 64, 65, 100, 128])
def test_long_adapters(m):
    """m <= 64: 64-bit bit-vectors; above: no pre-filter, exact DP on every read."""
    rng = random.Random(m)
    ref = util.random_dna(rng, m)
    for where, shortcut in (("BACK", 1), ("FRONT_NI", 0), ("BACK_NI", 0), ("ANYWHERE", 1), ("PREFIX", 0)):
        tp = one_adapter_plan(ref, 0.15, 3, WHERE[where], abi.CS_REMOVE_AFTER, False, shortcut)
        batch = util.batch_from_reads(adversarial_reads(rng, ref, 400, "ACGT", max_len=200))
        run_both(tp, batch, threads=4)


def test_adapter_with_non_acgt_base():
    rng = random.Random(9)
    ref = "ACGTNNACGTAC"
    tp = one_adapter_plan(ref, 0.2, 3, WHERE["BACK"], abi.CS_REMOVE_AFTER, False, 1)
    batch = util.batch_from_reads(adversarial_reads(rng, ref, 600, "ACGTN"))
    run_both(tp, batch, threads=4)


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("base,where", [("A", "BACK_NI"), ("T", "FRONT_NI"), ("A", "BACK"), ("G", "SUFFIX")])
@pytest.mark.parametrize("m,rate,mo", [(100, 0.15, 3), (20, 0.2, 3), (100, 0.4, 3), (8, 0.5, 5)])
def test_homopolymer_adapters(base, where, m, rate, mo, rule):
    """poly-A/T ops (cutseq/run.py:388-413): mismatch-count pre-filter + windowed DP."""
    rng = random.Random(m * 7 + len(where))
    reads = []
    for _ in range(1500):
        body = util.random_dna(rng, rng.randint(0, 120))
        run = "".join(base if rng.random() > rng.choice([0.0, 0.05, 0.2]) else rng.choice("ACGT")
                      for _ in range(rng.choice([0, 1, 2, 3, 4, 6, 10, 25, 60, 110, 130])))
        s = (body + run) if where in ("BACK_NI", "BACK", "SUFFIX") else (run + body)
        if rng.random() < 0.2:
            s = util.mutate(rng, s, 2)
        reads.append((s, "I" * len(s)))
    tp = one_adapter_plan(base * m, rate, mo, WHERE[where], abi.CS_REMOVE_AFTER if "BACK" in where or where == "SUFFIX"
                          else abi.CS_REMOVE_BEFORE, rule=rule)
    run_both(tp, util.batch_from_reads(reads), threads=4)


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("where,base", [("BACK_NI", "A"), ("FRONT_NI", "T")])
def test_poly_runs_with_scattered_errors(where, base, rule):
    """Long noisy homopolymer runs: many candidate rows, spans beyond m/2 (the leftmost rule's overlap
    window), runs cut by foreign bases right at the threshold steps -- the cases the closed form's
    "visit only where it can matter" short cut has to get right."""
    rng = random.Random(99 + rule + len(where))
    reads = []
    for _ in range(3000):
        parts = []
        for _ in range(rng.randint(1, 6)):
            parts.append(base * rng.choice([1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 30, 51, 52]))
            parts.append(util.random_dna(rng, rng.choice([0, 1, 1, 1, 2, 3]), "ACGTN"))
        run = "".join(parts)[: rng.choice([20, 60, 100, 101, 140])]
        body = util.random_dna(rng, rng.choice([0, 0, 1, 5, 40]))
        s = (body + run) if where == "BACK_NI" else (run[::-1] + body)
        reads.append((s, "I" * len(s)))
    remove = abi.CS_REMOVE_AFTER if where == "BACK_NI" else abi.CS_REMOVE_BEFORE
    for m, rate in ((100, 0.15), (100, 0.3), (24, 0.25)):
        run_both(one_adapter_plan(base * m, rate, 3, WHERE[where], remove, rule=rule), util.batch_from_reads(reads), threads=8)


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("anywhere", [False, True])
def test_leading_adapter_pair_in_one_walk(rule, anywhere, monkeypatch):
    """The scan kernel walks the chain's first two ops -- RightmostFrontAdapter, then BackAdapter (cutseq/run.py:332-355,
    544-590) -- in ONE forward pass (myers_pair): the 5' op as a forward existence test (full-length occurrences anywhere,
    adapter suffixes of at least min_overlap bases at the read's start), the 3' op as its exact filter.  Random adapter
    pairs, overlaps and rates; reads built around partial and damaged copies of BOTH adapters at both ends, short
    reads (tail columns inside the test's first groups), soft-masked stretches, the 3' op with ANYWHERE flags
    (--force-anywhere).  Checked against the oracle, and the same batch with the merged walk switched off."""
    rng = random.Random(977 + rule + 2 * anywhere)
    for trial in range(10):
        alpha = rng.choice(["ACGT", "ACGT", "ACGTN", "AC"])
        m5, m3 = rng.choice([8, 12, 16, 20, 24, 26]), rng.choice([5, 13, 20, 28, 32])
        p5 = util.random_dna(rng, m5, alpha.replace("N", ""))
        p3 = util.random_dna(rng, m3, alpha.replace("N", ""))
        rate5, rate3 = rng.choice([0.0, 0.1, 0.2]), rng.choice([0.0, 0.1, 0.2, 0.3])
        mo5, mo3 = rng.randint(1, m5), rng.randint(1, min(m3, 10))
        reads = []
        for _ in range(3000):
            style = rng.random()
            body = util.random_dna(rng, rng.choice([0, 3, 10, 40, 90, 130]), alpha)
            if style < 0.35:  # a suffix of the 5' adapter at the very start, damaged, then the insert
                cut = rng.randint(0, m5)
                head = util.mutate(rng, p5[cut:], rng.choice([0, 0, 1, 2, 3]), alpha)
                s = head + body
            elif style < 0.5:  # the whole 5' adapter somewhere inside
                s = util.random_dna(rng, rng.randint(0, 30), alpha) + util.mutate(rng, p5, rng.choice([0, 1, 2, 5]), alpha) + body
            else:
                s = body
            tail = rng.random()
            if tail < 0.4:  # read-through into the 3' adapter
                s += util.mutate(rng, p3, rng.choice([0, 0, 1, 2, 4]), alpha) + util.random_dna(rng, rng.choice([0, 5, 30]), alpha)
            elif tail < 0.6:  # ... only its first bases fit
                s += util.mutate(rng, p3[: rng.randint(1, m3)], rng.choice([0, 0, 1]), alpha)
            s = s[: rng.choice([7, 15, 23, 31, 33, 150, 150, 150])]
            reads.append((s, "I" * len(s)))
        batch = util.batch_from_reads(reads)
        util.soft_mask(batch, 0.1, seed=trial)
        ops = [planmod.AdapterOp("p5", p5[::-1], rate5, mo5, WHERE["BACK"], abi.CS_REMOVE_BEFORE, rightmost=True,
                                 match_flag=abi.CS_F_ADAPTER5),
               planmod.AdapterOp("p3", p3, rate3, mo3, WHERE["ANYWHERE" if anywhere else "BACK"], abi.CS_REMOVE_AFTER,
                                 match_flag=abi.CS_F_ADAPTER3)]
        tp = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                              select_rule=rule, use_filter=True, indel_tie=rng.choice([abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION]))
        g_pair, _ = run_both(tp, batch)
        with monkeypatch.context() as mp:
            mp.setenv("CUTSEQ_PAIR", "0")
            g_apart, _ = run_both(tp, batch)
        assert np.array_equal(g_pair, g_apart)


@pytest.mark.parametrize("m", [8, 9, 16, 17, 24, 25, 31, 32])
def test_end_rows_four_per_step_and_the_tail_group(m):
    """Round 5's two rewrites in the leading walk, swept instead of sampled: (a) which rows of the last column are
    acceptable comes from four rows per step -- byte b of a dword walks rows 8 b + 1 .. 8 b + 8 -- so the adapter
    lengths sit on both sides of every byte boundary, min_overlap moves the first tested row, and the reads end in
    EVERY prefix length of the adapter with 0 / 1 / 2 errors (and one base more or less: the neighbouring rows);
    (b) the columns behind a read's last whole group of eight run as one masked group whose scores are logged, so the
    read lengths cover every remainder mod 8 (and reads shorter than one group).  A 3' adapter alone (the walk
    without the 5' test) and the reference's pair (RightmostFrontAdapter in front of it); both selection rules."""
    rng = random.Random(4200 + m)
    p3 = util.random_dna(rng, m, "ACGT")
    p5 = util.random_dna(rng, 12, "ACGT")
    reads = []
    for overlap in range(0, m + 1):
        for errs in (0, 1, 2):
            for body_len in (0, 1, 5, 17, 40, 41, 42, 43, 44, 45, 46, 47):
                tail = util.mutate(rng, p3[:overlap], errs, "ACGT") if overlap else ""
                s = util.random_dna(rng, body_len, "ACGT") + tail
                reads.append((s, "I" * len(s)))
                if overlap and rng.random() < 0.3:  # the whole adapter further in, and a partial one at the end
                    s2 = util.random_dna(rng, 9, "ACGT") + util.mutate(rng, p3, errs, "ACGT") + util.random_dna(rng, body_len % 7, "ACGT") + tail
                    reads.append((s2, "I" * len(s2)))
    batch = util.batch_from_reads(reads)
    for rate, mo in ((0.1, 3), (0.2, 1), (0.2, 3), (0.34, min(9, m)), (0.2, m)):
        for rule in (0, 1):
            run_both(one_adapter_plan(p3, rate, mo, WHERE["BACK"], abi.CS_REMOVE_AFTER, rule=rule), batch, threads=8)
        ops = [planmod.AdapterOp("p5", p5[::-1], 0.2, 6, WHERE["BACK"], abi.CS_REMOVE_BEFORE, rightmost=True,
                                 match_flag=abi.CS_F_ADAPTER5),
               planmod.AdapterOp("p3", p3, rate, mo, WHERE["BACK"], abi.CS_REMOVE_AFTER, match_flag=abi.CS_F_ADAPTER3)]
        tp = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                              select_rule=0, use_filter=True, indel_tie=abi.CS_TIE_INSERTION)
        run_both(tp, batch, threads=8)


@pytest.mark.parametrize("slots", ["", "1", "3"])
@pytest.mark.parametrize("solo", [False, True])
def test_item_log_of_the_leading_walk_overflows_into_the_queue(solo, slots, monkeypatch):
    """The leading walk only LOGS the groups of eight columns in which the 3' op has a candidate (trim_kernel.hip.inc,
    ItemLog: a few slots per lane in LDS) and does the candidate book-keeping behind the loops.  A read with more flagged
    groups than slots must not be decided there: it goes to the top region of the deferral queue and the resolve kernel
    filters the chain's first op exactly.  Reads with two to six copies of the 3' adapter (whole, damaged, with indels,
    overlapping, back to back), pair form (5' + 3' op) and lone form (the 3' op opens the chain); the default log and
    logs of one and three slots (CUTSEQ_ITEM_SLOTS)."""
    if slots:
        monkeypatch.setenv("CUTSEQ_ITEM_SLOTS", slots)
    rng = random.Random(4242 + solo)
    p5, p3 = "ACACGACGCTCTTCCGATCT", "AGATCGGAAGAGCACACGTC"
    total_refiltered = 0
    for trial in range(3):
        rate3 = [0.2, 0.1, 0.3][trial]
        reads = []
        for _ in range(4000):
            s = util.random_dna(rng, rng.choice([0, 5, 20, 40]), "ACGT")
            for _copy in range(rng.choice([1, 2, 2, 3, 4, 6])):
                s += util.mutate(rng, p3, rng.choice([0, 0, 1, 2, 3, 5]), "ACGT")
                s += util.random_dna(rng, rng.choice([0, 0, 1, 3, 8, 17, 30]), "ACGT")
            if rng.random() < 0.2:  # low complexity: the adapter's own first bases over and over
                s += p3[:7] * rng.randint(1, 6)
            s = s[: rng.choice([60, 100, 150, 150, 150, 230, 300])]
            reads.append((s, "I" * len(s)))
        batch = util.batch_from_reads(reads)
        ops = []
        if not solo:
            ops.append(planmod.AdapterOp("p5", p5[::-1], 0.2, 10, WHERE["BACK"], abi.CS_REMOVE_BEFORE, rightmost=True,
                                         match_flag=abi.CS_F_ADAPTER5))
        ops.append(planmod.AdapterOp("p3", p3, rate3, 3, WHERE["BACK"], abi.CS_REMOVE_AFTER, match_flag=abi.CS_F_ADAPTER3))
        ops.append(planmod.CutOp(-4))
        for rule in (0, 1):
            tp = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                                  select_rule=rule, use_filter=True)
            run_both(tp, batch)
            with TrimEngine(tp, device=0, slots=1, max_reads=batch.n, max_stride=batch.stride) as eng:
                eng.trim(batch.seq1, batch.qual1, batch.len1, None, None, None)
                total_refiltered += eng.stats()[0].n_refiltered
    if slots != "3":
        assert total_refiltered > 1000  # the reads above do overflow a log of one or two slots


@pytest.mark.parametrize("solo", [False, True])
@pytest.mark.parametrize("rule", [0, 1])
def test_hits_one_column_behind_the_first_cheapest_column(rule, solo):
    """A substitution-only occurrence of the 3' adapter whose LAST base is one of the substitutions costs the same in the
    column before it (the last base skipped), so the first cheapest column is not the hit's own; myers_verdict settles
    such a hit one column on when the read's base there differs from the adapter's last base.  The reads crowd that
    rule: substitutions at the last one to three adapter bases, adapters that end in a repeated base (the rule must
    stand back), the adapter's last base repeated behind the hit, indels near the end, a second copy close by, soft
    masks, Ns; adapters of 10 to 28 bases at several rates; both selection rules and tie orders; the walk with and
    without the 5' op in front."""
    import os
    rng = random.Random(5150 + rule + 2 * solo + int(os.environ.get("CS_PREFIX_SEED", "0")))
    p5 = "ACACGACGCTCTTCCGATCT"
    for trial in range(int(os.environ.get("CS_PREFIX_TRIALS", "16"))):
        m = rng.choice([10, 13, 16, 20, 20, 21, 24, 28])
        p3 = util.random_dna(rng, m, "ACGT")
        if trial % 3 == 1:
            p3 = p3[:-1] + p3[-2]  # ends in a repeated base
        if trial % 5 == 4:
            p3 = p3[: m // 2] + p3[m // 2 - 1] * (m - m // 2)  # a homopolymer tail
        rate = rng.choice([0.1, 0.2, 0.2, 0.25, 0.3])
        reads = []
        for _ in range(3000):
            core = list(p3)
            style = rng.random()
            if style < 0.55:
                for pos in rng.sample([m - 1, m - 1, m - 2, m - 3, rng.randrange(m), rng.randrange(m)], rng.choice([1, 1, 2, 3])):
                    core[pos] = rng.choice("ACGTN")
                hit = "".join(core)
            elif style < 0.75:
                hit = "".join(core)
                pos = rng.choice([m - 1, m - 2, m, rng.randrange(m)])
                hit = hit[:pos] + rng.choice("ACGT") + hit[pos:] if rng.random() < 0.5 else hit[:max(pos - 1, 0)] + hit[pos:]
                hit = util.mutate(rng, hit, rng.choice([0, 1]), "ACGT")
            else:
                hit = util.mutate(rng, p3, rng.choice([0, 1, 2, 4]), "ACGT")
            behind = rng.choice(["", p3[-1], p3[-1] * 2, p3[:3], util.random_dna(rng, 2, "ACGT")]) + \
                util.random_dna(rng, rng.choice([0, 1, 5, 30]), "ACGT")
            if rng.random() < 0.1:
                behind = behind[:2] + util.mutate(rng, p3, 1, "ACGT") + behind[2:]
            sq = util.random_dna(rng, rng.choice([0, m + 3, 40, 70, 100]), "ACGT") + hit + behind
            sq = sq[: rng.choice([60, 100, 150, 150])]
            reads.append((sq, "I" * len(sq)))
        batch = util.batch_from_reads(reads)
        util.soft_mask(batch, 0.1, seed=trial)
        ops = []
        if not solo:
            ops.append(planmod.AdapterOp("p5", p5[::-1], 0.2, 10, WHERE["BACK"], abi.CS_REMOVE_BEFORE, rightmost=True,
                                         match_flag=abi.CS_F_ADAPTER5))
        ops.append(planmod.AdapterOp("p3", p3, rate, 3, WHERE["BACK"], abi.CS_REMOVE_AFTER, match_flag=abi.CS_F_ADAPTER3))
        ops.append(planmod.CutOp(-3))
        tie = rng.choice([abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION])
        tp = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                              select_rule=rule, use_filter=True, indel_tie=tie)
        run_both(tp, batch)


@pytest.mark.parametrize("rule", [0, 1])
def test_anchored_prefix_hits_with_substitutions_settle_in_the_filter(rule):
    """PrefixAdapter (the inline barcode, cutseq/run.py:357-362, 592-597): with at most two errors a substitution-only
    hit in column m settles in the filter when column m is the first candidate column and the cheapest (myers_verdict's
    anchored-start rule).  The reads below crowd its edges: barcodes of 4 to 12 bases, rates that give k = 0..3, copies
    with substitutions at the first / last base, insertions and deletions next to them (candidates one column to either
    side, equal and lower scores), repeats of the barcode's own ends behind it, homopolymer and two-letter barcodes,
    reads barely longer than the barcode, soft-masked bases; both selection rules and both indel tie orders."""
    import os
    # (CS_PREFIX_TRIALS / CS_PREFIX_SEED: the same test as a soak with fresh seeds)
    rng = random.Random(8128 + rule + int(os.environ.get("CS_PREFIX_SEED", "0")))
    settled = 0
    for trial in range(int(os.environ.get("CS_PREFIX_TRIALS", "24"))):
        m = rng.choice([4, 5, 6, 6, 7, 8, 9, 10, 12])
        alpha = rng.choice(["ACGT", "ACGT", "AC", "A"]) if trial % 4 == 3 else "ACGT"
        bc = util.random_dna(rng, m, alpha)
        rate = rng.choice([0.0, 0.12, 0.2, 0.2, 0.25, 0.34])
        reads = []
        for _ in range(2500):
            style = rng.random()
            core = list(bc)
            if style < 0.25:  # substitutions only, edges favoured
                for _e in range(rng.choice([0, 1, 1, 2, 3])):
                    pos = rng.choice([0, m - 1, rng.randrange(m)])
                    core[pos] = rng.choice("ACGTN")
                head = "".join(core)
            elif style < 0.5:  # one indel at an edge, maybe a substitution elsewhere
                head = "".join(core)
                pos = rng.choice([0, 1, m - 1, m])
                head = head[:pos] + rng.choice("ACGT") + head[pos:] if rng.random() < 0.5 else head[:max(pos - 1, 0)] + head[pos:]
                if rng.random() < 0.4:
                    head = util.mutate(rng, head, 1, "ACGT")
            elif style < 0.7:
                head = util.mutate(rng, bc, rng.choice([1, 2, 3]), "ACGT")
            else:
                head = util.random_dna(rng, m, alpha)
            tail = rng.choice([bc[-2:], bc[:2], bc, "", bc[-1] * 3]) + util.random_dna(rng, rng.choice([0, 1, 2, 3, 5, 40]), "ACGT")
            sq = (head + tail)[: rng.choice([m - 1, m, m + 1, m + 2, m + 4, 60, 60])]
            reads.append((sq, "I" * len(sq)))
        batch = util.batch_from_reads(reads)
        util.soft_mask(batch, 0.1, seed=trial)
        ops = [planmod.AdapterOp("bc", bc, rate, m, WHERE["PREFIX"], abi.CS_REMOVE_BEFORE, match_flag=abi.CS_F_INLINE),
               planmod.CutOp(2)]
        tp = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                              select_rule=rule, use_filter=True, indel_tie=rng.choice([abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION]))
        g1, _ = run_both(tp, batch)
        tp_full = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                                   select_rule=rule, use_filter=False, indel_tie=tp.indel_tie)
        f1, _ = run_both(tp_full, batch)  # the exact DP on every read gives the same records
        assert np.array_equal(g1, f1)
        with TrimEngine(tp, device=0, slots=1, max_reads=batch.n, max_stride=batch.stride) as eng:
            eng.trim(batch.seq1, batch.qual1, batch.len1, None, None, None)
            st = eng.stats()[0]
            settled += int(st.op_matched[0]) - int(st.n_exact_dp)
    assert settled > 5000  # many matches never see the DP


@pytest.mark.parametrize("m3,rate,walks", [(10, 0.2, False), (8, 0.125, False), (13, 0.1, True), (20, 0.2, True)])
def test_loose_short_adapters_keep_the_op_loops_filter(m3, rate, walks):
    """A short adapter with a loose error bound finds candidates in random sequence all the time; logged in two LDS slots
    per lane, nearly every read would overflow into the resolve kernel.  The plan estimates the candidate rate from (m, k)
    and lets such an op keep the op loop's own filter (cutseq_hip.hip, log_friendly): parity either way, and the share
    of reads that take the overflow road stays small."""
    rng = random.Random(31 + m3)
    p3 = util.random_dna(rng, m3, "ACGT")
    reads = []
    for _ in range(20000):
        s = util.random_dna(rng, rng.choice([40, 90, 150]), "ACGT")
        if rng.random() < 0.35:
            s = s[: rng.randint(0, len(s))] + util.mutate(rng, p3, rng.choice([0, 0, 1]), "ACGT") + util.random_dna(rng, 20, "ACGT")
        s = s[:150]
        reads.append((s, "I" * len(s)))
    batch = util.batch_from_reads(reads)
    ops = [planmod.AdapterOp("p3", p3, rate, 3, WHERE["BACK"], abi.CS_REMOVE_AFTER, match_flag=abi.CS_F_ADAPTER3),
           planmod.CutOp(-2)]
    tp = planmod.TrimPlan(r1=planmod.MateChain(ops), r2=None, has_umi=False, min_length=0, untrimmed_filter=False,
                          select_rule=0, use_filter=True)
    run_both(tp, batch)
    with TrimEngine(tp, device=0, slots=1, max_reads=batch.n, max_stride=batch.stride) as eng:
        eng.trim(batch.seq1, batch.qual1, batch.len1, None, None, None)
        refiltered = eng.stats()[0].n_refiltered
    assert refiltered <= (0.02 * batch.n if walks else 0)


@pytest.mark.parametrize("switch", ["", "CUTSEQ_PAIR=0", "CUTSEQ_EXISTS=0", "CUTSEQ_CUT_RUNS=0", "CUTSEQ_ITEM_SLOTS=1"])
@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("name,flags,paired", CHAIN_CASES)
def test_chain_presets(name, flags, paired, rule, switch, monkeypatch):
    """Every preset x flag chain; also with the merged walk of the leading adapter pair switched off (CUTSEQ_PAIR=0:
    existence-only 5' scan + exact 3' filter apart), with the existence-only scan off as well (CUTSEQ_EXISTS=0: the exact
    filter for every op) and with one op-loop turn per CUT op instead of one per run of them (CUTSEQ_CUT_RUNS=0) -- the
    switches are read when the plan is created."""
    if switch:
        monkeypatch.setenv(*switch.split("="))
    scheme = BUILDIN_ADAPTERS.get(name, name)
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    st.select_rule = rule
    batch = synth.generate_pairs(3000, 150, scheme, seed=23, chunk_index=len(name) + rule, single_end=not paired,
                                 poly_fraction=0.15, art5_fraction=0.05, indel_frac=0.2)
    rng = np.random.default_rng(5)
    for lens in (batch.len1, batch.len2):
        if lens is None:
            continue
        cut = rng.random(batch.n) < 0.25
        lens[cut] = rng.integers(0, 150, size=int(cut.sum())).astype(np.uint16)
    util.soft_mask(batch, 0.2)
    tp = util.compile_plan(scheme, st, paired)
    run_both(tp, batch)


def test_quality_trimming_extremes():
    rng = random.Random(4)
    reads = []
    for _ in range(3000):
        n = rng.randint(0, 150)
        s = util.random_dna(rng, n)
        style = rng.random()
        if style < 0.2:
            q = "#" * n
        elif style < 0.4:
            q = "I" * n
        else:
            q = "".join(rng.choice("#-9I5!+") for _ in range(n))
        reads.append((s, q))
    for cutoff in (0, 20, 41):
        tp = planmod.TrimPlan(r1=planmod.MateChain([planmod.QTrimOp(cutoff)]), r2=None, has_umi=False,
                              min_length=20, untrimmed_filter=False)
        run_both(tp, util.batch_from_reads(reads))


def test_fixture_slice_byte_identical_fastq():
    """BASELINE.json config 1 (on the 1000-pair slice of the reference's input data): the
    FASTQ text formatted from device results equals the string-level restatement's."""
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R2.fq.gz")
    batch = util.batch_from_records(rec1, rec2)
    names1, names2 = [r[0] for r in rec1], [r[0] for r in rec2]
    for flags in ({}, {"trim_polyA": True}):
        st = planmod.CutadaptConfig()
        for k, v in flags.items():
            setattr(st, k, v)
        scheme = BUILDIN_ADAPTERS["TAKARAV3"]
        tp = util.compile_plan(scheme, st, True)
        g1, g2 = run_both(tp, batch)
        got = util.format_batch(tp, batch, names1, names2, g1, None, g2)
        want = util.pyref_run(scheme, st, batch, names1, names2)
        assert got == want


def test_edge_batches():
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], planmod.CutadaptConfig(), True)
    # one pair, 257 pairs (tile + 1), all-empty reads
    for n in (1, 255, 256, 257, 513):
        b = synth.generate_pairs(n, 150, seed=n)
        run_both(tp, b)
    b = synth.generate_pairs(300, 150, seed=2)
    b.len1[:] = 0
    b.len2[:] = 0
    run_both(tp, b)
    # n_reads == 0 is a no-op
    with TrimEngine(tp, device=0, slots=1, max_reads=16, max_stride=152) as eng:
        e = np.zeros((0, 152), dtype=np.uint8)
        l = np.zeros(0, dtype=np.uint16)
        r1, _, r2 = eng.trim(e, e, l, e, e, l)
        assert r1.size == 0 and r2.size == 0


@pytest.mark.parametrize("read_len", [36, 75, 100, 151, 250, 300, 600, 1500])
def test_read_lengths_and_strides(read_len):
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    b = synth.generate_pairs(700, read_len, seed=read_len)
    run_both(tp, b)


def test_config2_single_adapter_large():
    """BASELINE.json config 2 shape (single 3' adapter AGATCGGAAGAGC, 10 % errors), 400k reads."""
    tp = planmod.single_adapter_plan("AGATCGGAAGAGC", 0.1, 3)
    b = synth.generate_single_adapter(400_000, 150, seed=77)
    run_both(tp, b, threads=oracle.host_threads())


def test_config3_full_takarav3_large_and_properties():
    """BASELINE.json config 3 shape: TAKARAV3 + --trim-polyA, 300k pairs bit-exact against
    the oracle, then size-independent properties on a 2M-pair batch (no oracle involved):
    sub-interval results, statistics = sums over the records, chunk invariance."""
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    b = synth.generate_pairs(300_000, 150, seed=31)
    run_both(tp, b, threads=oracle.host_threads())

    parts = [synth.generate_pairs(250_000, 150, seed=99, chunk_index=i) for i in range(8)]
    big = SynthBatch(*[np.concatenate([getattr(p, f) for p in parts]) for f in
                       ("seq1", "qual1", "len1", "seq2", "qual2", "len2")])
    with TrimEngine(tp, device=0, slots=1, max_reads=big.n, max_stride=big.stride) as eng:
        r1, _, r2 = eng.trim(big.seq1, big.qual1, big.len1, big.seq2, big.qual2, big.len2)
        s1, s2 = eng.stats(reset=True)
        for r, lens, stt in ((r1, big.len1, s1), (r2, big.len2, s2)):
            assert (r["start"] <= r["stop"]).all() and (r["stop"] <= lens).all()
            assert stt.n_reads == big.n and stt.in_bp == int(lens.sum(dtype=np.int64))
            assert stt.out_bp == int((r["stop"].astype(np.int64) - r["start"]).sum())
            assert stt.n_too_short == int(((r["flags"] & abi.CS_F_TOO_SHORT) != 0).sum())
            assert stt.op_matched[1] == int(((r["flags"] & abi.CS_F_ADAPTER3) != 0).sum())
        # UMI capture on R2 is the first 8 bases unless a 5' adapter was cut first (or a 3' hit left < 8)
        no5 = (r2["flags"] & abi.CS_F_ADAPTER5) == 0
        plain = no5 & ((r2["flags"] & abi.CS_F_ADAPTER3) == 0)
        assert (r2["cap_off"][no5] == 0).all() and (r2["cap_len"] <= 8).all() and (r2["cap_len"][plain] == 8).all()
        # chunk invariance: the second half alone gives the same records
        h = big.n // 2
        q1, _, q2 = eng.trim(big.seq1[h:], big.qual1[h:], big.len1[h:], big.seq2[h:], big.qual2[h:], big.len2[h:])
        assert (q1 == r1[h:]).all() and (q2 == r2[h:]).all()
        # row permutation equivariance
        perm = np.random.default_rng(1).permutation(h)
        p1, _, p2 = eng.trim(np.ascontiguousarray(big.seq1[perm]), np.ascontiguousarray(big.qual1[perm]),
                             np.ascontiguousarray(big.len1[perm]), np.ascontiguousarray(big.seq2[perm]),
                             np.ascontiguousarray(big.qual2[perm]), np.ascontiguousarray(big.len2[perm]))
        assert (p1 == r1[perm]).all() and (p2 == r2[perm]).all()


def test_filter_is_result_neutral_on_large_batch():
    """use_filter=0 (exact DP on every read, full range) == use_filter=1, GPU vs GPU."""
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    b = synth.generate_pairs(100_000, 150, seed=8, poly_fraction=0.1)
    out = []
    for f in (True, False):
        tp.use_filter = f
        with TrimEngine(tp, device=0, slots=1, max_reads=b.n, max_stride=b.stride) as eng:
            out.append(eng.trim(b.seq1, b.qual1, b.len1, b.seq2, b.qual2, b.len2))
    assert (out[0][0] == out[1][0]).all() and (out[0][2] == out[1][2]).all()


def test_engine_reuse_and_concurrent_engines():
    """One engine across many launches (the tile hand-out counter is reset per launch), several
    engines alive at once (one __constant__ plan slot each), a ninth engine is refused."""
    from cutseq_amd import capi
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp_a = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    tp_b = util.compile_plan(BUILDIN_ADAPTERS["INLINE"], planmod.CutadaptConfig(), True)
    ba = synth.generate_pairs(5000, 150, seed=3)
    bb = synth.generate_pairs(5000, 150, BUILDIN_ADAPTERS["INLINE"], seed=4)
    (oa1, _, _), (oa2, _, _) = util.oracle_run(tp_a, ba)
    (ob1, _, _), (ob2, _, _) = util.oracle_run(tp_b, bb)
    with TrimEngine(tp_a, slots=2, max_reads=5000, max_stride=152) as ea, \
            TrimEngine(tp_b, slots=2, max_reads=5000, max_stride=152) as eb:
        for rep in range(6):
            ra = ea.submit(rep & 1, ba.seq1, ba.qual1, ba.len1, ba.seq2, ba.qual2, ba.len2)
            rb = eb.submit(rep & 1, bb.seq1, bb.qual1, bb.len1, bb.seq2, bb.qual2, bb.len2)
            ea.wait(rep & 1)
            eb.wait(rep & 1)
            assert (ra[0] == oa1).all() and (ra[2] == oa2).all()
            assert (rb[0] == ob1).all() and (rb[2] == ob2).all()
        s1, _ = ea.stats()
        assert s1.n_reads == 6 * 5000
    engines = []
    try:
        for _ in range(8):
            engines.append(TrimEngine(tp_a, slots=0))
        with pytest.raises(capi.CsError):
            TrimEngine(tp_a, slots=0)
    finally:
        for e in engines:
            e.close()
    with TrimEngine(tp_a, slots=0):  # slots are released on close
        pass


def test_device_reproduces_frozen_result_checksums():
    """The committed CRC-32 values of the oracle's results on a seeded 200k-pair batch
    (tests/golden/synth_results_crc.json): the device path without the oracle in the loop."""
    import json
    import sys
    from pathlib import Path

    import make_results_golden as g

    def device_results(tp, batch):
        with TrimEngine(tp, device=0, slots=1, max_reads=batch.n, max_stride=batch.stride) as eng:
            r1, _, r2 = eng.trim(batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2)
        return r1, r2

    for case in json.loads((util.GOLDEN / "synth_results_crc.json").read_text()):
        got = g.crc_case(case["scheme"], case["flags"], case["rule"], device_results)
        assert (got["crc_r1"], got["crc_r2"]) == (case["crc_r1"], case["crc_r2"]), case


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_fuzz_odd_alphabets_qualities_and_lengths(seed):
    """Reads nobody should feed a trimmer but somebody will: lower case, IUPAC codes, dots, every
    printable quality character, lengths from 0 up, adapters planted with random damage -- through
    randomly drawn presets and flags, device against oracle."""
    rng = random.Random(1000 + seed)
    presets = sorted(BUILDIN_ADAPTERS) + [
        "ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNNXX<XXXNNNN(CGATGT)AGATCGGAAGAGCACACGTC",
        "ACGTACGTACGTAC(GATTACA)NN>XNN(TGCA)GGCCTTAAGGCCAATT", "AAGCAGTGGTATCAACGCAGAGTACXXXXXX-NNNNNNNNCTGTCTCTTATACACATCT"]
    alphabets = ["ACGT", "ACGTN", "ACGTacgt", "ACGTNRYKMSWBDHV", "ACGT.", "AAAAAAAC", "TTTTTTTG"]
    for round_ in range(8):
        name = rng.choice(presets)
        scheme = BUILDIN_ADAPTERS.get(name, name)
        bc = BarcodeConfig(scheme)
        st = planmod.CutadaptConfig()
        st.trim_polyA = rng.random() < 0.7
        st.trim_polyA_wo_direction = rng.random() < 0.3
        st.conditional_cutter = rng.random() < 0.7
        st.force_anywhere = rng.random() < 0.3
        st.ensure_inline_barcode = rng.random() < 0.5
        st.min_length = rng.choice([0, 1, 20, 35, 151])
        st.min_quality = rng.choice([0, 2, 20, 30, 41, 93])
        st.force_trim_min_length = rng.choice([0, 50, 120, 10000])
        st.select_rule = rng.choice([0, 1])
        st.indel_tie = rng.choice([abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION])
        st.case_rule = abi.CS_CASE_SENSITIVE if rng.random() < 0.25 else abi.CS_CASE_FOLD
        st.shortcut = abi.CS_SHORTCUT_FIND if rng.random() < 0.25 else abi.CS_SHORTCUT_NONE
        paired = rng.random() < 0.7
        pieces = [bc.p5.fw, bc.p7.fw, bc.p5.rc, bc.p7.rc, "A" * 30, "T" * 30]
        reads1, reads2 = [], []
        for _ in range(1200):
            pair = []
            for _mate in range(2):
                alpha = rng.choice(alphabets)
                parts = []
                for _ in range(rng.randint(0, 4)):
                    if rng.random() < 0.5:
                        parts.append(util.random_dna(rng, rng.randint(0, 60), alpha))
                    else:
                        piece = rng.choice(pieces)
                        piece = piece[rng.randint(0, len(piece) // 2):][: rng.randint(1, len(piece))]
                        parts.append(util.mutate(rng, piece, rng.randint(0, 3), alpha))
                seq = "".join(parts)[: rng.choice([0, 1, 7, 19, 20, 21, 50, 149, 150, 151, 155])]
                qual = "".join(chr(rng.randint(33, 126)) for _ in seq)
                pair.append((seq, qual))
            reads1.append(pair[0])
            reads2.append(pair[1])
        batch = util.batch_from_reads(reads1, reads2 if paired else None)
        tp = util.compile_plan(scheme, st, paired, untrimmed_requested=rng.random() < 0.3)
        run_both(tp, batch)


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("use_filter", [True, False])
def test_full_hit_inside_plus_partial_hit_at_the_end(rule, use_filter):
    """Every read carries a damaged full-length adapter somewhere and an adapter prefix (exact or with
    an error) at its very end, at every distance from each other -- the case where the filter may drop
    the end-of-read rows (leftmost rule, far apart) or must keep them (score rule, or close together).
    All 64 lanes of a wave need the exact DP here, so the strip passes run four deep."""
    rng = random.Random(41 + rule)
    ad = "AGATCGGAAGAGCACACGTC"
    reads = []
    for _ in range(4000):
        head = util.random_dna(rng, rng.randint(0, 60))
        hit = util.mutate(rng, ad, rng.randint(0, 4))
        gap = util.random_dna(rng, rng.choice([0, 1, 2, 3, 5, 8, 9, 10, 11, 12, 15, 20, 30, 40, 70]))
        tail = util.mutate(rng, ad[: rng.randint(3, 19)], rng.choice([0, 0, 1, 1, 2]))
        s = (head + hit + gap + tail)[:200]
        reads.append((s, "I" * len(s)))
    for where, mo, shortcut, tie in (("BACK", 3, 1, 0), ("ANYWHERE", 3, 1, 1), ("BACK", 10, 0, 1), ("BACK", 3, 0, 0),
                                     ("ANYWHERE", 3, 0, 0)):
        tp = one_adapter_plan(ad, 0.2, mo, WHERE[where], abi.CS_REMOVE_AFTER, False, shortcut, rule, use_filter, tie)
        run_both(tp, util.batch_from_reads(reads), threads=8)
    # the same through the reversed aligner (RightmostFrontAdapter)
    rev = [(s[::-1], q) for s, q in reads]
    for shortcut in (1, 0):
        tp = one_adapter_plan(ad[::-1], 0.2, 10, WHERE["BACK"], abi.CS_REMOVE_BEFORE, True, shortcut, rule, use_filter)
        run_both(tp, util.batch_from_reads(rev), threads=8)


@pytest.mark.parametrize("rule", [0, 1])
def test_inexact_occurrence_in_front_of_an_exact_one(rule):
    """No short cut (the default): an exact copy does not win by itself.  A damaged copy at every distance
    in front of it -- inside the m/2 overlap window (the exact one replaces it), just outside, far away
    (leftmost rule: the damaged one stays; score rule: the exact one wins) -- and the filter's own
    decision 'exact hit right behind its own run' against the full DP."""
    rng = random.Random(97 + rule)
    ad = "AGATCGGAAGAGCACACGTC"
    reads = []
    for _ in range(3000):
        head = util.random_dna(rng, rng.randint(0, 40))
        bad = util.mutate(rng, ad, rng.randint(1, 4))
        gap = util.random_dna(rng, rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 20, 30]))
        if rng.random() < 0.3:
            bad, gap = "", ""
        s = (head + bad + gap + ad + util.random_dna(rng, rng.randint(0, 12)))[:220]
        reads.append((s, "I" * len(s)))
    batch = util.batch_from_reads(reads)
    util.soft_mask(batch, 0.1)
    for use_filter in (True, False):
        for tie in (abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION):
            tp = one_adapter_plan(ad, 0.2, 3, WHERE["BACK"], abi.CS_REMOVE_AFTER, False, 0, rule, use_filter, tie)
            run_both(tp, batch, threads=8)
    rev = util.batch_from_reads([(s[::-1], q) for s, q in reads])
    tp = one_adapter_plan(ad[::-1], 0.2, 10, WHERE["BACK"], abi.CS_REMOVE_BEFORE, True, 0, rule, True)
    run_both(tp, rev, threads=8)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_hits_that_settle_in_the_filter_stress(seed):
    """The filter settles exact and substitution-only hits without the DP.  Reads built to sit on
    every edge of those arguments: substitutions only / one indel, repeats and low-complexity adapters (ties),
    hits at the very start (alignments that begin in the first column), at the very end (rows against
    full-length cells), two copies at every distance, reads barely longer than m + k -- all flag sets that take
    the path, both selection rules, both tie orders, filter on (and the full DP as the reference point through
    the oracle)."""
    rng = random.Random(500 + seed)
    for trial in range(16):
        alpha = rng.choice(["ACGT", "ACGT", "AC", "ACG"])
        m = rng.choice([6, 8, 12, 13, 16, 20, 24, 31, 32])
        ref = util.random_dna(rng, m, alpha) if rng.random() < 0.7 else (util.random_dna(rng, 3, alpha) * 12)[:m]
        rate = rng.choice([0.1, 0.15, 0.2, 0.25, 0.34])
        k = int(rate * m)
        where = rng.choice(["BACK", "BACK", "BACK_NI", "PREFIX", "SUFFIX", "FRONT", "ANYWHERE"])
        rightmost = where == "BACK" and rng.random() < 0.4
        mo = rng.choice([1, 3, min(10, m), m])
        reads = []
        for _ in range(4000):
            hit = list(ref)
            for _s in range(rng.choice([0, 0, 1, 1, 2, 3, k, k + 1])):  # substitutions
                hit[rng.randrange(m)] = rng.choice(alpha)
            hit = "".join(hit)
            if rng.random() < 0.15:
                hit = util.mutate(rng, hit, 1, alpha)  # an indel or another substitution
            cut = rng.random()
            if cut < 0.25:
                hit = hit[: rng.randint(1, max(1, len(hit)))]  # partial copy (meant for the read end)
            elif cut < 0.35:
                hit = hit[rng.randint(0, max(0, len(hit) - 1)):]  # partial copy (meant for the read start)
            left = util.random_dna(rng, rng.choice([0, 0, 1, 2, k, m, m + k - 1, m + k, m + k + 1, 40]), alpha)
            right = util.random_dna(rng, rng.choice([0, 0, 0, 1, 2, k, m // 2, m, 30]), alpha)
            s = left + hit + right
            if rng.random() < 0.2:  # a second copy somewhere behind
                s += util.random_dna(rng, rng.choice([0, 1, m // 2, m])) + hit[: rng.randint(1, max(1, len(hit)))]
            if rightmost:
                s = s[::-1]
            reads.append((s, "I" * len(s)))
        batch = util.batch_from_reads(reads)
        seq = ref[::-1] if rightmost else ref
        remove = abi.CS_REMOVE_BEFORE if (rightmost or where in ("PREFIX", "FRONT")) else abi.CS_REMOVE_AFTER
        for rule in (0, 1):
            tie = rng.choice([abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION])
            tp = one_adapter_plan(seq, rate, mo, WHERE[where], remove, rightmost, 0, rule, True, tie)
            run_both(tp, batch, threads=8)


@pytest.mark.parametrize("read_len", [150, 151, 75, 250])
def test_fast_recoding_and_its_fallback(read_len, monkeypatch):
    """The scan kernel's fast re-coding (trim_kernel.hip.inc, encode4_pairs_fast): (ascii >> 1) & 7 IS the base code for
    A, C, G, T and N, one look-up checks that nothing else is there, a wave that finds something else stages the tile
    again with the exact form and stays with it.  (1) Clean rows padded with 'N' (what this package's producers write)
    through the fast form ALONE -- CUTSEQ_FAST_RECODE=2 never falls back, so a green run proves the fast codes themselves;
    (2) the same reads with zero padding, with garbage behind the reads, with IUPAC codes, lower case, '@', NUL and bytes
    above 127 inside them: the check must send those tiles to the exact form (default mode), results == oracle either way;
    (3) the aliases are real: 'B' reads as C and 'D' as T when the check is ignored, so mode 2 on dirty reads differs."""
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    clean = synth.generate_pairs(20_000, read_len, seed=read_len, poly_fraction=0.1, art5_fraction=0.02)
    assert (clean.seq1[:, read_len:] == ord("N")).all() and (clean.qual1[:, read_len:] == 0).all()
    monkeypatch.setenv("CUTSEQ_FAST_RECODE", "2")
    run_both(tp, clean)
    monkeypatch.delenv("CUTSEQ_FAST_RECODE")
    rng = np.random.default_rng(read_len)

    def variant(kind):
        b = SynthBatch(clean.seq1.copy(), clean.qual1.copy(), clean.len1.copy(), clean.seq2.copy(), clean.qual2.copy(), clean.len2.copy())
        for seq, lens in ((b.seq1, b.len1), (b.seq2, b.len2)):
            if kind == "zero padding":
                seq[:, read_len:] = 0
            elif kind == "garbage behind shorter reads":
                cut = rng.random(b.n) < 0.3
                lens[cut] = rng.integers(0, read_len, size=int(cut.sum())).astype(np.uint16)  # the bases behind stay: garbage
            else:  # odd bytes inside a few reads: some tiles fall back, most do not
                rows = np.flatnonzero(rng.random(b.n) < 0.01)
                odd = np.frombuffer(b"RYKMSWBDHVnacgt@\x00\x80\xffU.-*", dtype=np.uint8)
                for r in rows:
                    for _ in range(int(rng.integers(1, 4))):
                        seq[r, int(rng.integers(0, read_len))] = odd[int(rng.integers(0, odd.size))]
        return b

    for kind in ("zero padding", "garbage behind shorter reads", "odd bytes"):
        for mode in ("1", "0"):
            monkeypatch.setenv("CUTSEQ_FAST_RECODE", mode)
            run_both(tp, variant(kind))
        monkeypatch.delenv("CUTSEQ_FAST_RECODE")
    # (3) an adapter copy whose C is written 'B' and whose T is written 'D': no match for the oracle and the checked forms,
    # a match for the unchecked fast form
    ad = "AGATCGGAAGAGCACACGTC"
    bad_ad = ad.replace("C", "B", 1).replace("T", "D", 1)
    reads = [(util.random_dna(random.Random(i), 60) + (bad_ad if i % 2 else ad) + "GGTTGGTTGG", None) for i in range(256)]
    reads = [(s, "I" * len(s)) for s, _ in reads]
    b = util.batch_from_reads(reads, reads)
    b.seq1[b.seq1 == 0] = ord("N")
    b.seq2[b.seq2 == 0] = ord("N")
    tp1 = planmod.TrimPlan(r1=planmod.MateChain([planmod.back(ad, 0.0, 20, flag=abi.CS_F_ADAPTER3)]),
                           r2=planmod.MateChain([planmod.back(ad, 0.0, 20, flag=abi.CS_F_ADAPTER3)]), has_umi=False, min_length=0,
                           untrimmed_filter=False)
    g1, _ = run_both(tp1, b)
    assert ((g1["flags"] & abi.CS_F_ADAPTER3) != 0).tolist() == [i % 2 == 0 for i in range(256)]
    monkeypatch.setenv("CUTSEQ_FAST_RECODE", "2")
    with TrimEngine(tp1, device=0, slots=1, max_reads=b.n, max_stride=b.stride) as eng:
        u1, _, _ = eng.trim(b.seq1, b.qual1, b.len1, b.seq2, b.qual2, b.len2)
    assert ((u1["flags"] & abi.CS_F_ADAPTER3) != 0).all()  # unchecked, 'B' and 'D' pass for C and T: the check is not decoration
