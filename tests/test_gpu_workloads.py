"""The BASELINE.json configurations at bench scale, exactly as ``bench.py`` builds them
(``cutseq_amd.workloads``): same schemes, same barcode set, same seeded read generator -- HIP path against the
CPU oracle, bit for bit.

  config 4  the bench's CONFIG4_SCHEME (inline barcode + 8-nt UMI + dual adapters, --ensure-inline-barcode),
            300 k pairs                                            (reference chain: cutseq/run.py:592-603, 771-784)
  config 5  96-plex demultiplex, 200 k pairs: results, barcode index and ambiguity flag against 96 independent
            ``--ensure-inline-barcode`` runs of the oracle          (parity definition of SURVEY.md 8 f-4)
  config 3 / config 2  the headline and the single-end micro workload, 400 k / 500 k reads
"""
import numpy as np
import pytest

import oracle
from cutseq_amd import abi, plan as planmod, workloads
from cutseq_amd.common import BarcodeConfig
from cutseq_amd.engine import TrimEngine

from test_gpu_parity import run_both

pytestmark = pytest.mark.gpu


def test_config4_exact_bench_scheme_300k_pairs():
    tp = workloads.make_plan("config4")
    assert tp.untrimmed_filter  # --ensure-inline-barcode with an inline barcode in the scheme
    batch = workloads.make_batch("config4", 300_000)
    g1, g2 = run_both(tp, batch, threads=16)
    inline = (g1["flags"] & abi.CS_F_INLINE) != 0
    assert 0.5 < float(inline.mean()) < 1.0  # the generator plants the barcode; sequencing errors lose a few
    assert ((g1["flags"] & abi.CS_F_UNTRIMMED) != 0).any()  # a13: IsUntrimmedAny has something to route
    assert (g1["cap_len"] == 8).mean() > 0.99  # the 8-nt UMI behind the inline barcode leaves R1 for the read name


@pytest.mark.parametrize("switch", ["", "CUTSEQ_PAIR=0"])
@pytest.mark.parametrize("workload,n", [("config3", 400_000), ("config2", 500_000)])
def test_headline_and_single_end_workloads(workload, n, switch, monkeypatch):
    """Both walk the chain's first op(s) in front of the op loop (config 3: the 5' + 3' pair in one walk, config 2: its
    lone 3' adapter through the same code with one recurrence); CUTSEQ_PAIR=0 sends them through the op loop's own
    filters instead."""
    if switch:
        monkeypatch.setenv(*switch.split("="))
    run_both(workloads.make_plan(workload), workloads.make_batch(workload, n), threads=16)


def test_config5_96plex_200k_pairs_equal_96_independent_runs():
    n = 200_000
    codes = workloads.config5_barcodes()
    assert len(codes) == 96 and len(set(codes)) == 96
    tp = workloads.make_plan("config5")
    assert tp.demux is not None and tp.untrimmed_filter
    batch = workloads.make_batch("config5", n)
    bc = np.empty(n, dtype=np.uint8)
    with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
        res = eng.submit(0, batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2, bc=bc)
        eng.wait(0)
        g1, _, g2 = res
        gst1, gst2 = eng.stats()
    amb = (g1["flags"] & abi.CS_F_AMBIGUOUS) != 0
    assert amb.mean() < 0.02

    # 96 independent runs of the reference shape: --ensure-inline-barcode with ONE barcode each.  Only mate 1's
    # chain carries the barcode (PrefixAdapter, run.py:592-597); mate 2's chain is the same in every run.
    st = planmod.CutadaptConfig()
    st.ensure_inline_barcode = True
    runs = []
    o2 = None
    for index, code in enumerate(codes):
        one = planmod.compile_paired(BarcodeConfig(workloads.config5_scheme([code])), st)
        a1, n1, a2, n2 = one.pack()
        params = one.params()
        o1, _, _ = oracle.trim_mate(a1, n1, params, batch.seq1, batch.qual1, batch.len1, threads=16)
        runs.append(o1)
        if index == 0:
            o2, _, ost2 = oracle.trim_mate(a2, n2, params, batch.seq2, batch.qual2, batch.len2, threads=16)
            chain2 = bytes(a2)
        else:
            assert bytes(a2) == chain2  # mate 2 never sees which barcode it is
    matched = np.stack([(o1["flags"] & abi.CS_F_INLINE) != 0 for o1 in runs])  # [barcode, read]
    assert np.array_equal(matched.sum(axis=0) > 1, amb)  # the flag marks exactly the reads several runs claim
    want_bc = np.where(matched.any(axis=0), matched.argmax(axis=0), abi.CS_DEMUX_NONE).astype(np.uint8)
    assert np.array_equal(bc[~amb], want_bc[~amb])
    assert np.all(matched[bc[amb].astype(np.int64), np.nonzero(amb)[0]])  # an ambiguous read goes to a claimant
    assigned = float((bc != abi.CS_DEMUX_NONE).mean())
    assert 0.9 < assigned < 0.995, assigned  # 2 % foreign barcodes, damaged ones beyond one error
    assert len(np.unique(bc[bc != abi.CS_DEMUX_NONE])) == 96  # every barcode of the plex is in use
    own = np.where(bc == abi.CS_DEMUX_NONE, 0, bc).astype(np.int64)  # unassigned reads look the same in every run
    want1 = np.stack(runs)[own, np.arange(n)]
    want1["flags"] |= np.where(amb, abi.CS_F_AMBIGUOUS, 0).astype(np.uint8)
    bad = np.nonzero(g1 != want1)[0]
    assert bad.size == 0, (bad[:5], g1[bad[:5]], want1[bad[:5]])
    assert np.array_equal(g2, o2)
    assert int(gst1.op_matched[2]) == int((bc != abi.CS_DEMUX_NONE).sum())  # op 2 of mate 1 is the demultiplexer
    assert int(gst2.out_bp) == int(ost2.out_bp)


def test_one_large_launch_equals_the_same_rows_in_small_launches():
    """BASELINE config 3 is quoted on ONE resident batch of 100 M pairs (bench.py's default launch; it checks this
    property itself at full size, ``full_size_launch_equals_piecewise_launches``).  The size-independent form here:
    3 M pairs in one launch -- 46 875 tiles, big and small hand-out units, every queue reservation path -- against the same
    rows in launches of 1 M, 1 M and the ragged rest; the first 300 k pairs of it against the oracle."""
    n = 3_000_017
    tp = workloads.make_plan("config3")
    batch = workloads.make_batch("config3", n)
    with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
        w1, _, w2 = eng.trim(batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2)
        w1, w2 = w1.copy(), w2.copy()
        for lo, hi in ((0, 1_000_000), (1_000_000, 2_000_000), (2_000_000, n)):
            p1, _, p2 = eng.trim(batch.seq1[lo:hi], batch.qual1[lo:hi], batch.len1[lo:hi],
                                 batch.seq2[lo:hi], batch.qual2[lo:hi], batch.len2[lo:hi])
            assert np.array_equal(p1, w1[lo:hi]) and np.array_equal(p2, w2[lo:hi])
    a1, n1, a2, n2 = tp.pack()
    m = 300_000
    o1, _, _ = oracle.trim_mate(a1, n1, tp.params(), batch.seq1[:m], batch.qual1[:m], batch.len1[:m], threads=16)
    o2, _, _ = oracle.trim_mate(a2, n2, tp.params(), batch.seq2[:m], batch.qual2[:m], batch.len2[:m], threads=16)
    assert np.array_equal(w1[:m], o1) and np.array_equal(w2[:m], o2)
