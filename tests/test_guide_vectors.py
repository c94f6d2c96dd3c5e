"""Known answers from the published cutadapt user guide (tests/guide_vectors.py) on all three
implementations: the string pipeline and the C oracle here, the HIP kernel under ``-m gpu``.

These are the only adapter/quality expectations in the suite that are not derived from this
repo's own restatement of cutadapt's source (parity unpinned).
"""
import numpy as np
import pytest

import oracle
from oracle import pyref
from cutseq_amd import abi, plan as planmod

import guide_vectors as gv
import util
from test_oracle import WHERE

VECTORS = gv.all_vectors()
IDS = [f"{group}-{kind}-{i}" for i, (group, kind, *_rest) in enumerate(VECTORS)]


def vector_plan(kind, adapter, rate, min_overlap, **rules):
    _cls, where, before, rightmost = gv.KINDS[kind]
    op = planmod.AdapterOp(kind, adapter, rate, min_overlap, WHERE[where],
                           abi.CS_REMOVE_BEFORE if before else abi.CS_REMOVE_AFTER, rightmost=rightmost,
                           match_flag=abi.CS_F_ADAPTER3)
    return planmod.TrimPlan(r1=planmod.MateChain([op]), r2=None, has_umi=False, min_length=0,
                            untrimmed_filter=False, **rules)


def kept_of(read: str, res) -> str:
    return read[int(res["start"]): int(res["stop"])]


@pytest.mark.parametrize("vec", VECTORS, ids=IDS)
def test_guide_vector_string_pipeline_and_c_oracle(vec):
    group, kind, adapter, rate, mo, read, kept = vec
    cls = getattr(pyref, gv.KINDS[kind][0])
    if kind in ("prefix", "suffix"):
        ad = cls(adapter, rate)
    else:
        ad = cls(adapter, rate, mo)
    m = ad.match_to(read)
    got = read if m is None else (read[m.rstop:] if m.remove_before else read[: m.rstart])
    assert got == kept
    batch = util.batch_from_reads([(read, "I" * len(read))])
    for rule in (abi.CS_SELECT_LEFTMOST, abi.CS_SELECT_SCORE):
        for tie in (abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION):
            (res, _, _), _ = util.oracle_run(vector_plan(kind, adapter, rate, mo, select_rule=rule, indel_tie=tie), batch)
            assert kept_of(read, res[0]) == kept, (rule, tie)


def test_case_vectors_need_the_fold():
    """The case vectors are exactly the ones a case-sensitive comparison gets wrong."""
    for kind, adapter, rate, mo, read, kept in gv.CASE:
        batch = util.batch_from_reads([(read, "I" * len(read))])
        (res, _, _), _ = util.oracle_run(vector_plan(kind, adapter, rate, mo, case_rule=abi.CS_CASE_SENSITIVE), batch)
        assert kept_of(read, res[0]) != kept


def test_error_tolerance_table():
    for rate, table in gv.ERROR_TOLERANCE:
        op = planmod.AdapterOp("BackAdapter", "ACGT" * 8, rate, 3, abi.CS_WHERE_BACK, abi.CS_REMOVE_AFTER)
        thr = op.thresholds()
        for length, errors in table.items():
            assert thr[length] == errors, (rate, length)


def test_quality_trimming_worked_example():
    for quals, cutoff, keep in gv.QUALITY:
        s = "".join(chr(33 + q) for q in quals)
        assert oracle.quality_trim_index(s, cutoff) == keep
        assert pyref.quality_trim_index(s, 0, cutoff) == (0, keep) or keep == 0


@pytest.mark.gpu
def test_guide_vectors_on_the_device():
    """All vectors through the C ABI: one engine per (kind, adapter, rate, overlap), every selection /
    tie rule, filter on and off; the device must print what the guide prints."""
    from cutseq_amd.engine import TrimEngine

    groups = {}
    for _group, kind, adapter, rate, mo, read, kept in VECTORS:
        groups.setdefault((kind, adapter, rate, mo), []).append((read, kept))
    for (kind, adapter, rate, mo), items in groups.items():
        batch = util.batch_from_reads([(r, "I" * len(r)) for r, _ in items])
        for rule in (abi.CS_SELECT_LEFTMOST, abi.CS_SELECT_SCORE):
            for tie in (abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION):
                for use_filter in (True, False):
                    tp = vector_plan(kind, adapter, rate, mo, select_rule=rule, indel_tie=tie, use_filter=use_filter)
                    with TrimEngine(tp, device=0, slots=1, max_reads=batch.n, max_stride=batch.stride) as eng:
                        res, _, _ = eng.trim(batch.seq1, batch.qual1, batch.len1)
                    for (read, kept), r in zip(items, res):
                        assert kept_of(read, r) == kept, (kind, adapter, read, rule, tie, use_filter)


@pytest.mark.gpu
def test_quality_worked_example_on_the_device():
    from cutseq_amd.engine import TrimEngine

    reads = []
    for quals, cutoff, keep in gv.QUALITY:
        assert cutoff == 10
        reads.append(("ACGTACGTAC"[: len(quals)], "".join(chr(33 + q) for q in quals)))
    batch = util.batch_from_reads(reads)
    tp = planmod.TrimPlan(r1=planmod.MateChain([planmod.QTrimOp(10)]), r2=None, has_umi=False, min_length=0,
                          untrimmed_filter=False)
    with TrimEngine(tp, device=0, slots=1, max_reads=batch.n, max_stride=batch.stride) as eng:
        res, _, _ = eng.trim(batch.seq1, batch.qual1, batch.len1)
    assert [int(r["stop"]) for r in res] == [keep for _, _, keep in gv.QUALITY]
    assert np.all(res["start"] == 0)
