"""CPU checks of the oracle itself (no GPU).

PARITY UNPINNED w.r.t. real cutadapt (absent here).  What these tests do pin:
  * construction-certain known answers that do not depend on tie-breaking,
  * the documented cutadapt user-guide examples for 3'/5' adapters,
  * the ConditionalCutter truth table (cutseq/run.py:145-161),
  * hand-computed BWA quality-trimming cases,
  * agreement of two independently written restatements (C intervals vs Python string
    slicing) on randomized inputs, every aligner flag set, both selection rules,
  * the op-chain compiler against the string-level pipeline for presets x flags.
"""
import random

import numpy as np
import pytest

import oracle
from oracle import pyref
from cutseq_amd import abi, plan as planmod, synth
import hostfmt
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig

import util

WHERE = {
    "BACK": abi.CS_WHERE_BACK, "FRONT": abi.CS_WHERE_FRONT, "PREFIX": abi.CS_WHERE_PREFIX,
    "SUFFIX": abi.CS_WHERE_SUFFIX, "FRONT_NI": abi.CS_WHERE_FRONT_NOT_INTERNAL,
    "BACK_NI": abi.CS_WHERE_BACK_NOT_INTERNAL, "ANYWHERE": abi.CS_WHERE_ANYWHERE,
}


# ---------------------------------------------------------------- aligner known answers


def test_locate_exact_internal():
    ad = "AGATCGGAAGAGCACACGTC"
    ins = "TTGACCTGAACCTTGGAACCTTGACCTGAA"  # shares no 3-mer prefix with the adapter start at its end
    read = ins + ad + "GGGTTT"
    for rule in (0, 1):
        assert oracle.locate(ad, read, 0.2, abi.CS_WHERE_BACK, 3, rule) == (0, 20, 30, 50, 20, 0)


def test_locate_no_match():
    assert oracle.locate("AGATCGGAAGAGC", "TTTTTTTTTTTTTTTTTTTTTTTTTTTT", 0.1, abi.CS_WHERE_BACK, 3) is None
    assert oracle.locate("AGATCGGAAGAGC", "", 0.1, abi.CS_WHERE_BACK, 3) is None


def test_cutadapt_guide_3prime_examples():
    """cutadapt user guide, regular 3' adapter `-a ADAPTER` (min overlap 3, no errors needed)."""
    a = pyref.BackAdapter("ADAPTER", 0.1, 3)
    cases = {
        "mysequenceADAPTERsomethingelse": "mysequence",
        "mysequenceADAPTER": "mysequence",
        "mysequenceADAP": "mysequence",
        "mysequenceADA": "mysequence",
        "ADAPTERsomething": "",
        "mysequence": "mysequence",
    }
    for read, want in cases.items():
        m = a.match_to(read)
        got = read if m is None else read[: m.rstart]
        assert got == want, read
        # C restatement agrees
        c = oracle.locate("ADAPTER", read, 0.1, abi.CS_WHERE_BACK, 3)
        if read.find("ADAPTER") < 0:
            assert (c is None) == (m is None)
            if c:
                assert c[2] == m.rstart


def test_cutadapt_guide_5prime_examples():
    """`-g ADAPTER` regular 5' adapter examples from the user guide (FRONT flags)."""
    cases = {
        "ADAPTERmysequence": "mysequence",
        "DAPTERmysequence": "mysequence",
        "TERmysequence": "mysequence",
        "somethingADAPTERmysequence": "mysequence",
        "mysequence": "mysequence",
    }
    for read, want in cases.items():
        hit = oracle.locate("ADAPTER", read, 0.1, abi.CS_WHERE_FRONT, 3)
        got = read if hit is None else read[hit[3]:]
        assert got == want, read
        p = pyref.Aligner("ADAPTER", 0.1, pyref.FRONT, 3).locate(read)
        assert p == hit


def test_leftmost_of_two_equal_occurrences():
    """user guide: 'the leftmost match is used for both 5' and 3' adapters'."""
    read = "cccADAPTERgggggADAPTERttt"
    for rule in (0, 1):
        hit = oracle.locate("ADAPTER", read, 0.1, abi.CS_WHERE_BACK, 3, rule)
        assert hit[2] == 3
        hit = oracle.locate("ADAPTER", read, 0.1, abi.CS_WHERE_FRONT, 3, rule)
        assert hit[3] == 10


def test_select_rules_differ_on_worse_then_better_occurrence():
    """first occurrence carries one mismatch, a later disjoint one is exact: cutadapt >= 4
    keeps the first (leftmost) hit inside Aligner.locate, the 3.x score rule jumps to the second."""
    ad = "ACGTTGCAATGCCGTA"
    bad = "ACGTTGCTATGCCGTA"
    read = "GGGGGGGG" + bad + "GGGGGGGGGGGGGGGGGGGG" + ad + "GG"
    left = oracle.locate(ad, read, 0.2, abi.CS_WHERE_BACK, 3, abi.CS_SELECT_LEFTMOST)
    score = oracle.locate(ad, read, 0.2, abi.CS_WHERE_BACK, 3, abi.CS_SELECT_SCORE)
    assert left[2] == 8 and left[5] == 1
    assert score[2] == 8 + 16 + 20 and score[5] == 0
    # the adapter-level answer is the aligner's (cutadapt >= 3: match_to has no str.find in front of it) ...
    m = pyref.BackAdapter(ad, 0.2, 3).match_to(read)
    assert (m.rstart, m.errors) == (8, 1)
    # ... unless the cutadapt <= 2.x short cut is switched on: then the exact copy behind it wins
    m = pyref.BackAdapter(ad, 0.2, 3, shortcut=pyref.SHORTCUT_FIND).match_to(read)
    assert (m.rstart, m.errors) == (44, 0)
    # the same through the op table / C oracle
    for sc, want in ((abi.CS_SHORTCUT_NONE, 8), (abi.CS_SHORTCUT_FIND, 44)):
        tp = _one_read_plan([planmod.back(ad, 0.2, 3, flag=abi.CS_F_ADAPTER3, shortcut=sc)])
        r, _ = _run_single(tp, read)
        assert int(r["stop"]) == want


def test_exact_hit_right_behind_an_inexact_run_replaces_it():
    """An exact copy is preceded by its own run of inexact candidates (the same occurrence seen k
    columns early): they overlap, the exact column scores higher and Aligner.locate stops there."""
    ad = "AGATCGGAAGAGCACACGTC"
    ins = "TTGACCTGAACCTTGGAACCTTGACCTGAA"
    for rule in (0, 1):
        assert oracle.locate(ad, ins + ad + "TTGG", 0.2, abi.CS_WHERE_BACK, 3, rule)[2:6] == (30, 50, 20, 0)


def test_indel_tie_rule_changes_the_origin():
    """cost_insertion == cost_deletion < cost_diag: the two orders take origin/score from different
    cells.  CS_TIE_INSERTION is SURVEY.md appendix B.2's order; both are restated, C == Python."""
    rng = random.Random(77)
    differ = 0
    for _ in range(3000):
        ref = util.random_dna(rng, rng.randint(6, 20))
        query = util.random_dna(rng, rng.randint(0, 10)) + util.mutate(rng, ref, rng.randint(1, 4)) + \
            util.random_dna(rng, rng.randint(0, 10))
        rate = rng.choice([0.2, 0.3])
        res = []
        for tie in (abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION):
            c = oracle.locate(ref, query, rate, abi.CS_WHERE_BACK, 3, 0, tie)
            p = pyref.Aligner(ref, rate, pyref.BACK, 3, 0, tie).locate(query)
            assert c == p
            res.append(c)
        differ += res[0] != res[1]
    assert differ > 0


def test_rightmost_front_adapter():
    ad = "ACACGACGCTCTTCCGATCT"
    read = "TT" + ad + "GGAGG" + ad + "CCATTGGA"
    m = pyref.RightmostFrontAdapter(ad, 0.2, 10).match_to(read)
    assert m.rstop == 2 + 20 + 5 + 20
    # one mismatch in the right copy: it is still the rightmost occurrence (the aligner walks the reversed
    # read and keeps its first hit); only the cutadapt <= 2.x rfind short cut prefers the exact left copy
    read2 = "TT" + ad + "GGAGG" + ad[:5] + "T" + ad[6:] + "CCATTGGA"
    m2 = pyref.RightmostFrontAdapter(ad, 0.2, 10).match_to(read2)
    assert m2.rstop == 47 and m2.errors == 1
    m2 = pyref.RightmostFrontAdapter(ad, 0.2, 10, shortcut=pyref.SHORTCUT_FIND).match_to(read2)
    assert m2.rstop == 22
    # no exact copy at all: aligner on the reversed strings finds the rightmost one
    bad = ad[:5] + "T" + ad[6:]
    read3 = "TT" + bad + "GGAGG" + bad + "CCATTGGA"
    m3 = pyref.RightmostFrontAdapter(ad, 0.2, 10).match_to(read3)
    assert m3.rstop == 47 and m3.errors == 1


def test_partial_adapter_needs_min_overlap():
    ad = "AGATCGGAAGAGCACACGTC"
    body = "TTGACCTGAACCTTGGAACCTTGACCTGTT"
    assert oracle.locate(ad, body + "AG", 0.2, abi.CS_WHERE_BACK, 3) is None
    assert oracle.locate(ad, body + "AGA", 0.2, abi.CS_WHERE_BACK, 3) == (0, 3, 30, 33, 3, 0)
    # 5 nt prefix with one mismatch: floor(5 * 0.2) = 1 error allowed
    hit = oracle.locate(ad, body + "AGTTC", 0.2, abi.CS_WHERE_BACK, 3)
    assert hit is not None and hit[3] == 35


def test_error_threshold_table_is_float_floor():
    for rate, m in ((0.2, 20), (0.15, 100), (0.1, 13), (0.2, 6)):
        op = planmod.AdapterOp("BackAdapter", "A" * m, rate, 3, abi.CS_WHERE_BACK, abi.CS_REMOVE_AFTER)
        thr = op.thresholds()
        for L in range(m + 1):
            assert all((c <= L * rate) == (c <= thr[L]) for c in range(m + 2))
        assert op.k == int(rate * m)
    op = planmod.AdapterOp("BackAdapter", "A" * 20, 0.2, 3, abi.CS_WHERE_BACK, abi.CS_REMOVE_AFTER)
    assert op.thresholds() == [0] * 5 + [1] * 5 + [2] * 5 + [3] * 5 + [4]


@pytest.mark.parametrize("where", sorted(WHERE))
@pytest.mark.parametrize("rule", [0, 1])
def test_c_locate_equals_python_locate_random(where, rule):
    """Two independent restatements agree cell for cell on adversarial random inputs
    (small alphabets provoke ties; lengths cross the m+k window edges)."""
    rng = random.Random(hash((where, rule)) & 0xFFFF)
    flags = WHERE[where]
    n_hits = 0
    for it in range(1500):
        alpha = rng.choice(["AC", "ACG", "ACGT", "A", "ACGTN"])
        m = rng.randint(1, 24)
        ref = util.random_dna(rng, m, alpha.replace("N", "") or "A")
        rate = rng.choice([0.0, 0.1, 0.15, 0.2, 0.34, 0.5])
        mo = rng.randint(1, m)
        style = rng.random()
        if style < 0.4:
            query = util.random_dna(rng, rng.randint(0, 60), alpha)
        else:
            core = util.mutate(rng, ref, rng.randint(0, 3), alpha)
            cut = rng.random()
            if cut < 0.3:
                core = core[: rng.randint(0, len(core))]
            elif cut < 0.6:
                core = core[rng.randint(0, len(core)):]
            query = util.random_dna(rng, rng.randint(0, 30), alpha) + core + util.random_dna(rng, rng.randint(0, 30), alpha)
            if rng.random() < 0.3:
                query += util.mutate(rng, ref, rng.randint(0, 2), alpha) + util.random_dna(rng, rng.randint(0, 8), alpha)
        c = oracle.locate(ref, query, rate, flags, mo, rule)
        p = pyref.Aligner(ref, rate, flags, min(mo, m), rule).locate(query)
        assert c == p, (ref, query, rate, mo)
        n_hits += c is not None
    assert n_hits > 100


# ---------------------------------------------------------------- quality trimming


def test_quality_trim_hand_cases():
    q = oracle.quality_trim_index
    assert q("IIIIIIII", 20) == 8                # all Q40: nothing trimmed
    assert q("IIII####", 20) == 4                # Q2 tail goes
    assert q("########", 20) == 0                # everything goes
    assert q("", 20) == 0
    assert q("IIII#I", 20) == 6                  # s = -20 at the last base: break at once
    assert q("III#I#", 20) == 5                  # 18 > 0 at the last base -> stop=5; then -20+18<0 break
    # BWA running sum: '5' = Q20 contributes 0 and never moves the maximum
    assert q("IIII5555", 20) == 8
    assert q("IIII5#5#", 20) == 5                # i=7: 18 (stop=7); i=6: 18; i=5: 36 (stop=5); i=4: 36; i=3: <0
    # low-quality base shielded by a long good tail
    assert q("II#IIIII", 20) == 8
    for s in ["IIII####", "I-9#I-9#", "#I#I#I#I", "9999----", "-"]:
        for cutoff in (0, 10, 20, 30, 41):
            assert q(s, cutoff) == pyref.quality_trim_index(s, 0, cutoff)[1] or \
                pyref.quality_trim_index(s, 0, cutoff) == (0, 0)


def test_quality_trim_random_vs_python():
    rng = random.Random(5)
    for _ in range(2000):
        s = "".join(rng.choice("#-9I5!+") for _ in range(rng.randint(0, 40)))
        cutoff = rng.choice([0, 2, 13, 20, 25, 40])
        start, stop = pyref.quality_trim_index(s, 0, cutoff)
        c = oracle.quality_trim_index(s, cutoff)
        assert start == 0
        assert c == stop, (s, cutoff)


# ---------------------------------------------------------------- ConditionalCutter


def _one_read_plan(ops, min_length=0):
    return planmod.TrimPlan(r1=planmod.MateChain(list(ops)), r2=None, has_umi=False, min_length=min_length,
                            untrimmed_filter=False)


def _run_single(tp, seq, qual=None):
    qual = qual or "I" * len(seq)
    batch = util.batch_from_reads([(seq, qual)])
    (res, cap2, st), _ = util.oracle_run(tp, batch)
    return res[0], st


def test_conditional_cutter_truth_table():
    """cutseq/run.py:154-155: skip iff (no adapter matched on this mate) and len < force_trim_min_length."""
    ad = "AGATCGGAAGAGCACACGTC"
    back = planmod.back(ad, 0.2, 3, flag=abi.CS_F_ADAPTER3)
    cond = planmod.CutOp(-8, conditional=True, force_min_len=50)
    tp = _one_read_plan([back, cond])
    body_short = "TTGACCTGAACCTTGGAACCTTGACCTGTT"      # 30 nt
    body_long = body_short + "CCATTGGACCTTGAACCTTGGACCTT"  # 56 nt
    r, _ = _run_single(tp, body_short)                    # unmatched, short  -> untouched
    assert (r["start"], r["stop"]) == (0, 30)
    r, _ = _run_single(tp, body_long)                     # unmatched, long   -> cut
    assert (r["start"], r["stop"]) == (0, len(body_long) - 8)
    r, _ = _run_single(tp, body_short + ad)               # matched, short    -> cut
    assert (r["start"], r["stop"]) == (0, 22) and r["flags"] & abi.CS_F_ADAPTER3
    r, _ = _run_single(tp, body_long + ad)                # matched, long     -> cut
    assert (r["start"], r["stop"]) == (0, len(body_long) - 8)
    # length is evaluated when the cutter runs (after earlier cuts)
    tp2 = _one_read_plan([planmod.CutOp(10), cond])
    r, _ = _run_single(tp2, body_long)                    # 56 - 10 = 46 < 50 -> skipped
    assert (r["start"], r["stop"]) == (10, len(body_long))
    # unconditional variant always cuts; python slicing semantics on short reads
    tp3 = _one_read_plan([planmod.CutOp(-8), planmod.CutOp(5, capture=1)])
    r, _ = _run_single(tp3, "ACGTAC")
    assert r["stop"] - r["start"] == 0 and r["cap_len"] == 0
    r, _ = _run_single(tp3, "ACGTACGTACG")                # 11 -> 3 left -> capture 3, empty read
    assert (r["cap_off"], r["cap_len"], r["stop"] - r["start"]) == (0, 3, 0)


def test_takarav3_readme_walkthrough():
    """reference README.md:13-26 on one constructed pair (R1 loses 8+6 at the 3' end after the
    adapter, 3 at the 5' end; R2 loses 8(UMI)+6 at 5', 3 at 3'; both names get _UMI)."""
    bc = BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"])
    st = planmod.CutadaptConfig()
    tp = planmod.compile_paired(bc, st)
    insert = "TTGACCTGAACCTTGGAACCTTGACCTGTTCCATTGGACC"  # 40 nt
    x5, x3, umi = "GGG", "CATCAT", "ACGTTGCA"
    from cutseq_amd.common import reverse_complement as rc
    r1 = x5 + insert + x3 + umi + bc.p7.fw + "GGGGGGGG"
    r2 = rc(umi) + rc(x3) + rc(insert) + rc(x5) + bc.p5.rc + "GGGGGGGG"
    batch = util.batch_from_reads([(r1, "I" * len(r1))], [(r2, "I" * len(r2))])
    (res1, _, _), (res2, _, _) = util.oracle_run(tp, batch)
    out = util.format_batch(tp, batch, [b"read7 1:N:0:X"], [b"read7 2:N:0:X"], res1, None, res2)
    rt, rec1, rec2 = out[0]
    assert rt == hostfmt.ROUTE_TRIMMED
    assert rec1 == f"@read7_{rc(umi)}\n{insert}\n+\n{'I' * 40}\n".encode()
    assert rec2 == f"@read7_{rc(umi)}\n{rc(insert)}\n+\n{'I' * 40}\n".encode()
    ref = util.pyref_run(BUILDIN_ADAPTERS["TAKARAV3"], st, batch, [b"read7 1:N:0:X"], [b"read7 2:N:0:X"])
    assert ref == out


# ---------------------------------------------------------------- chains: compiler + C oracle vs string pipeline

CHAIN_CASES = [
    ("TAKARAV3", {}, True),
    ("TAKARAV3", {"trim_polyA": True}, True),
    ("TAKARAV3", {"trim_polyA": True, "trim_polyA_wo_direction": True, "min_quality": 25}, True),
    ("TAKARAV3", {"conditional_cutter": False, "min_length": 35}, True),
    ("TAKARAV3", {"force_anywhere": True, "force_trim_min_length": 120}, True),
    ("TAKARAV3", {"trim_polyA": True, "auto_rc": True}, False),
    ("SACSEQV3", {"trim_polyA": True}, True),
    ("SACSEQV3", {"trim_polyA": True}, False),
    ("INLINE", {"ensure_inline_barcode": True}, True),
    ("INLINE", {"ensure_inline_barcode": True, "trim_polyA": True}, False),
    ("UNSTRANDED", {"trim_polyA": True}, True),
    ("SMALLRNA", {"auto_rc": True}, True),
    ("XGENRNA", {"auto_rc": True, "trim_polyA": True}, True),
    ("NEXTERA", {}, False),
    ("ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNNXX<XXXNNNN(CGATGT)AGATCGGAAGAGCACACGTC",
     {"ensure_inline_barcode": True, "trim_polyA": True}, True),
    ("ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNNXX<XXXNNNN(CGATGT)AGATCGGAAGAGCACACGTC",
     {"ensure_inline_barcode": True}, False),
    # the recalled cutadapt rules, each switched to its non-default setting (include/cutseq_hip.h)
    ("TAKARAV3", {"trim_polyA": True, "shortcut": abi.CS_SHORTCUT_FIND}, True),
    ("TAKARAV3", {"trim_polyA": True, "indel_tie": abi.CS_TIE_DELETION}, True),
    ("SACSEQV3", {"case_rule": abi.CS_CASE_SENSITIVE, "indel_tie": abi.CS_TIE_DELETION}, False),
]


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("name,flags,paired", CHAIN_CASES)
def test_chain_matches_string_pipeline(name, flags, paired, rule):
    scheme = BUILDIN_ADAPTERS.get(name, name)
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    st.select_rule = rule
    batch = synth.generate_pairs(300, 150, scheme, seed=11, chunk_index=len(name) + rule, single_end=not paired,
                                 poly_fraction=0.15, art5_fraction=0.05, indel_frac=0.2)
    # ragged lengths: truncate some reads, including to zero
    rng = np.random.default_rng(3)
    for lens in (batch.len1, batch.len2):
        if lens is None:
            continue
        cut = rng.random(batch.n) < 0.25
        lens[cut] = rng.integers(0, 150, size=int(cut.sum())).astype(np.uint16)
    util.soft_mask(batch, 0.2)
    names1 = [f"SIM:{i} 1:N:0:X".encode() for i in range(batch.n)]
    names2 = [f"SIM:{i} 2:N:0:X".encode() for i in range(batch.n)]
    names1[5], names2[5] = b"plain/1", b"plain/2"
    names1[6], names2[6] = b"dot.1", b"dot.2"
    names1[7], names2[7] = b"tab\tcomment/1", b"tab\tcomment/2"
    tp = util.compile_plan(scheme, st, paired)
    (res1, cap2, st1), m2 = util.oracle_run(tp, batch)
    got = util.format_batch(tp, batch, names1, names2, res1, cap2, m2[0] if m2 else None)
    want = util.pyref_run(scheme, st, batch, names1, names2 if paired else None)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, (i, g, w)
    routes = [g[0] for g in got]
    assert routes.count(hostfmt.ROUTE_TRIMMED) > 0


def test_fixture_subset_plumbing():
    """BASELINE.json config 1 on a 1000-pair slice of the reference's own input data
    (tests/golden/fixture1k_R{1,2}.fq.gz = first 1000 records of test/input_R{1,2}.fq.gz).
    The reference ships no expected output; checked: C oracle == string pipeline, pair
    conservation, output is a sub-interval, names = id_UMI with UMI = R2[0:8] (when no 5' adapter was cut first)."""
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R2.fq.gz")
    assert len(rec1) == len(rec2) == 1000
    batch = util.batch_from_records(rec1, rec2)
    st = planmod.CutadaptConfig()
    scheme = BUILDIN_ADAPTERS["TAKARAV3"]
    tp = util.compile_plan(scheme, st, True)
    names1, names2 = [r[0] for r in rec1], [r[0] for r in rec2]
    (res1, _, s1), (res2, _, s2) = util.oracle_run(tp, batch)
    got = util.format_batch(tp, batch, names1, names2, res1, None, res2)
    want = util.pyref_run(scheme, st, batch, names1, names2)
    assert got == want
    assert s1.n_reads == s2.n_reads == 1000
    n_short = sum(1 for g in got if g[0] == hostfmt.ROUTE_SHORT)
    assert 0 < n_short < 1000
    for i, (rt, o1, o2) in enumerate(got):
        h1, sq1 = o1.split(b"\n")[0], o1.split(b"\n")[1]
        assert sq1 in rec1[i][1]
        if not res2[i]["flags"] & abi.CS_F_ADAPTER5:  # no 5' artefact cut first: UMI = R2[0:8]
            assert h1 == b"@" + names1[i].split()[0] + b"_" + rec2[i][1][:8]
    # a good share of R1 carries the 3' adapter (~27-33 %)
    frac = s1.op_matched[1] / 1000
    assert 0.2 < frac < 0.6


def test_oracle_reproduces_frozen_result_checksums():
    """tests/golden/synth_results_crc.json (tests/make_results_golden.py): CRC-32 of the per-read
    results on a seeded 200k-pair batch; guards the oracle (and the synthetic generator) against drift."""
    import json
    import sys
    from pathlib import Path

    import make_results_golden as g

    frozen = json.loads((util.GOLDEN / "synth_results_crc.json").read_text())
    assert len(frozen) == len(g.CASES)
    case = frozen[0]  # one case on the CPU (the GPU suite checks all of them on the device)
    got = g.crc_case(case["scheme"], case["flags"], case["rule"], g.oracle_results)
    assert (got["crc_r1"], got["crc_r2"]) == (case["crc_r1"], case["crc_r2"])


@pytest.mark.parametrize("seed", list(range(1, 11)))
def test_fuzz_c_oracle_vs_string_pipeline_on_odd_inputs(seed):
    """The two restatements (C on intervals + native-style formatter vs Python on strings) on reads
    with lower case, IUPAC codes, dots, arbitrary printable qualities and awkward lengths, through
    randomly drawn schemes and flags."""
    rng = random.Random(7000 + seed)
    presets = sorted(BUILDIN_ADAPTERS) + [
        "ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNNXX<XXXNNNN(CGATGT)AGATCGGAAGAGCACACGTC",
        "ACGTACGTACGTAC(GATTACA)NN>XNN(TGCA)GGCCTTAAGGCCAATT"]
    alphabets = ["ACGT", "ACGTN", "ACGTacgt", "ACGTNRYKMSWBDHV", "ACGT.", "AAAAAAAC", "TTTTTTTG"]
    for _round in range(3):
        name = rng.choice(presets)
        scheme = BUILDIN_ADAPTERS.get(name, name)
        bc = BarcodeConfig(scheme)
        st = planmod.CutadaptConfig()
        st.trim_polyA = rng.random() < 0.7
        st.trim_polyA_wo_direction = rng.random() < 0.3
        st.conditional_cutter = rng.random() < 0.7
        st.force_anywhere = rng.random() < 0.3
        st.ensure_inline_barcode = rng.random() < 0.5
        st.auto_rc = rng.random() < 0.3
        st.min_length = rng.choice([0, 1, 20, 35])
        st.min_quality = rng.choice([0, 2, 20, 30, 41])
        st.select_rule = rng.choice([0, 1])
        paired = rng.random() < 0.6
        pieces = [bc.p5.fw, bc.p7.fw, bc.p5.rc, bc.p7.rc, "A" * 30, "T" * 30]
        reads1, reads2 = [], []
        for _ in range(250):
            pair = []
            for _mate in range(2):
                alpha = rng.choice(alphabets)
                parts = []
                for _ in range(rng.randint(0, 4)):
                    if rng.random() < 0.5:
                        parts.append(util.random_dna(rng, rng.randint(0, 50), alpha))
                    else:
                        piece = rng.choice(pieces)
                        piece = piece[rng.randint(0, len(piece) // 2):][: rng.randint(1, len(piece))]
                        parts.append(util.mutate(rng, piece, rng.randint(0, 3), alpha))
                seq = "".join(parts)[: rng.choice([0, 1, 7, 19, 20, 21, 50, 101, 150])]
                qual = "".join(chr(rng.randint(33, 126)) for _ in seq)
                pair.append((seq, qual))
            reads1.append(pair[0])
            reads2.append(pair[1])
        batch = util.batch_from_reads(reads1, reads2 if paired else None)
        names1 = [f"r{i} c".encode() for i in range(batch.n)]
        names2 = [f"r{i} d".encode() for i in range(batch.n)]
        untrimmed_requested = rng.random() < 0.3
        tp = util.compile_plan(scheme, st, paired, untrimmed_requested)
        (res1, cap2, _), m2 = util.oracle_run(tp, batch)
        got = util.format_batch(tp, batch, names1, names2, res1, cap2, m2[0] if m2 else None)
        want = util.pyref_run(scheme, st, batch, names1, names2 if paired else None, untrimmed_requested)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, (name, i, g, w)
