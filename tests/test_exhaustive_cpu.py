"""CPU twin of the exhaustive small-universe sweep (tests/universe.py): the two independently written restatements of
``Aligner.locate`` -- C on intervals (oracle/cutseq_oracle.c, thresholds as an integer table) and Python on strings
(oracle/pyref.py, ``cost <= length * rate`` in floating point) -- agree on EVERY read of a small universe for every one
of the 504 aligner settings.  The GPU test (tests/test_gpu_exhaustive.py) then holds the HIP kernels to the C oracle on
the large universe.  Path protected: cutseq/run.py:332-370, 544-615.
"""
import multiprocessing as mp

import numpy as np
import pytest

import oracle
from oracle import pyref
from cutseq_amd import abi

import universe as U
import util


def test_the_rotation_covers_every_setting_and_every_adapter():
    for alphabet, per in (("AC", 6), ("ACG", 1)):
        n = len(U.adapters(alphabet))
        seen = set()
        for a in range(n):
            mine = list(U.settings_of(a, per))
            assert len(set(mine)) == per
            seen.update(mine)
        assert seen == set(range(len(U.SETTINGS))), (alphabet, len(seen))
        groups = U.schedule(n, per)
        assert sum(len(items) for _r, _t, items in groups) == n * per and len(groups) == 4
    for k in (0, 1, 2):  # the rates really give k, and thresholds that step inside the adapter
        for m in range(3, 8):
            op = U.adapter_op("A" * (m - 1) + "C", (k, "1", abi.CS_WHERE_BACK, False, 0, 0))
            assert op.k == k and op.thresholds()[m] == k and op.thresholds()[0] == 0


def _reads_small():
    parts = [U.reads_universe("AC", 10), U.reads_universe("ACG", 6), U.reads_universe("ACN", 6), U.reads_universe("ACGN", 4)]
    return U.stack(parts, 12)


def _pyref_results(args):
    """One (adapter, setting) against every read, through the Python restatement -> [(start, stop, matched)]."""
    seq_str, setting, reads = args
    op = U.adapter_op(seq_str, setting)
    rule, tie = setting[4], setting[5]
    al = pyref.Aligner(op.sequence[::-1] if op.rightmost else op.sequence, op.max_error_rate, op.where, op.min_overlap, rule, tie)
    out = []
    for read in reads:
        n = len(read)
        aln = al.locate(read[::-1] if op.rightmost else read)
        if aln is None:
            out.append((0, n, 0))
            continue
        qs, qe = aln[2], aln[3]
        if op.rightmost:
            qs, qe = n - qe, n - qs
        out.append((qe, n, 1) if op.remove == abi.CS_REMOVE_BEFORE else (0, qs, 1))
    return out


def test_c_oracle_equals_python_restatement_on_the_whole_small_universe():
    """Every read over {A,C} up to length 10, over {A,C,G} and {A,C,N} up to length 6, over {A,C,G,N} up to length 4
    (4 574 reads) x all 504 settings, each setting on three adapters (two of the 120 of length 3..6 over {A,C}, one of
    the 108 of length 3..4 over {A,C,G}; every adapter is met): 6.9 M alignments, each computed twice."""
    seq, qual, lens = _reads_small()
    reads = [seq[i, :lens[i]].tobytes().decode() for i in range(seq.shape[0])]
    ads2, ads3 = U.adapters("AC", 3, 6), U.adapters("ACG", 3, 4)
    jobs = []
    for s in range(len(U.SETTINGS)):
        for ad in (ads2[s % len(ads2)], ads2[(s * 7 + 60) % len(ads2)], ads3[s % len(ads3)]):
            jobs.append((ad, U.SETTINGS[s], reads))
    assert {j[0] for j in jobs} >= set(ads2) | set(ads3)
    workers = max(1, min(8, oracle.host_threads()))
    with mp.get_context("fork").Pool(workers) as pool:
        py = pool.map(_pyref_results, jobs, chunksize=4)
    bad = []
    for (ad, setting, _), want in zip(jobs, py):
        tp = U.one_op_plan([U.adapter_op(ad, setting)], None, setting[4], setting[5])
        a1, n1, _a2, _n2 = tp.pack()
        got, _, _ = oracle.trim_mate(a1, n1, tp.params(), seq, qual, lens)
        w = np.array(want, dtype=np.int64)
        same = (got["start"] == w[:, 0]) & (got["stop"] == w[:, 1]) & (((got["flags"] & abi.CS_F_ADAPTER3) != 0) == (w[:, 2] != 0))
        if not same.all():
            i = int(np.flatnonzero(~same)[0])
            bad.append((ad, setting, reads[i], (int(got["start"][i]), int(got["stop"][i]), int(got["flags"][i])), want[i]))
    assert not bad, bad[:5]
