"""Exhaustive small-universe parity sweep of the HIP kernels against the C oracle.

tests/universe.py says what is enumerated and why the settings rotate over the adapters.  Every read of the universe x
every adapter of length 3..7 over the same alphabet, each adapter under several of the 504 aligner settings (all of
them met), filter on: device results == C oracle, bit for bit.  The C oracle itself is held to the independent Python
restatement on a smaller universe by the CPU twin (tests/test_exhaustive_cpu.py).  What this guards is the verdict of
the bit-parallel filter -- the rules that settle a hit WITHOUT the exact DP (trim_kernel.hip.inc, myers_verdict:
exact hit, substitution-only hit, the hit one column on, anchored prefix hits, dropped end rows) -- and the windows it
hands to the exact DP; path protected: ``Aligner.locate`` as cutseq/run.py:332-370, 544-615 calls it.

Reads are resident on the device (uploaded once per case); a plan is a PAIRED plan whose two mates run two different
(adapter, setting) ops over the same reads.  Free-ended forward ops additionally run with CUTSEQ_LOG_ALWAYS=1, which
sends them through the scan kernel's merged forward walk (myers_pair) instead of the op loop's filter -- short adapters
are otherwise kept out of it (their candidates would flood the item log).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle
from cutseq_amd import abi
from cutseq_amd.engine import TrimEngine

import universe as U

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, oracle.host_threads()))

# settings per adapter (tests/universe.py: the rotation meets all 504 settings from 3 per two-letter adapter on).
# CUTSEQ_EXHAUSTIVE_FULL=1: the WHOLE product over the two-letter universe (248 adapters x 504 settings x 160 595 reads =
# 2.0e10 alignments, ~8 minutes on the GPU box) and twelve settings per three-letter adapter; run by hand once per round,
# result in profiles/rNN_exhaustive_full.log.
FULL = os.environ.get("CUTSEQ_EXHAUSTIVE_FULL") == "1"
PER_AC = len(U.SETTINGS) if FULL else 48
PER_ACG, PER_ACG7 = (12, 6) if FULL else (3, 2)
PAIR_PARTNERS = 24 if FULL else 6
GROUPS = [(abi.CS_SELECT_LEFTMOST, abi.CS_TIE_INSERTION), (abi.CS_SELECT_LEFTMOST, abi.CS_TIE_DELETION),
          (abi.CS_SELECT_SCORE, abi.CS_TIE_INSERTION), (abi.CS_SELECT_SCORE, abi.CS_TIE_DELETION)]


class Resident:
    """One batch of reads on the device, both mates looking at the same rows, with result arrays for each mate."""

    def __init__(self, seq, qual, lens):
        dev = torch.device("cuda:0")
        self.seq, self.qual, self.lens = seq, qual, lens
        self.n, self.stride = seq.shape
        self.t = [torch.from_numpy(seq).to(dev), torch.from_numpy(qual).to(dev), torch.from_numpy(lens.view(np.int16)).to(dev)]
        self.out = [torch.zeros((self.n, 8), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.r = [abi.cs_reads(self.t[0].data_ptr(), self.t[1].data_ptr(), self.t[2].data_ptr(), o.data_ptr(), None, None)
                  for o in self.out]
        self.want = [np.zeros(self.n, dtype=abi.RESULT_DTYPE) for _ in range(2)]
        self.stream = torch.cuda.Stream(device=dev)
        self.handle = C.c_void_p(self.stream.cuda_stream)

    def run(self, tp):
        """-> the two mates' device results (views, valid until the next run)."""
        with TrimEngine(tp, device=0, slots=0) as eng:
            eng.trim_device(self.r[0], self.r[1], self.n, self.stride, stream=self.handle)
            self.stream.synchronize()
        return [o.cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1) for o in self.out]

    def expect(self, tp):
        a1, n1, a2, n2 = tp.pack()
        params = tp.params()
        oracle.trim_mate(a1, n1, params, self.seq, self.qual, self.lens, threads=THREADS, out=self.want[0])
        oracle.trim_mate(a2, n2, params, self.seq, self.qual, self.lens, threads=THREADS, out=self.want[1])
        return self.want

    def check(self, tp, what, monkeypatch=None, log_always=False):
        want = self.expect(tp)
        forms = [""] + (["CUTSEQ_LOG_ALWAYS"] if log_always else [])
        for env in forms:
            if env:
                monkeypatch.setenv(env, "1")
            try:
                got = self.run(tp)
            finally:
                if env:
                    monkeypatch.delenv(env)
            for mate in (0, 1):
                if not np.array_equal(got[mate], want[mate]):
                    i = int(np.flatnonzero(got[mate] != want[mate])[0])
                    read = self.seq[i, :self.lens[i]].tobytes().decode()
                    raise AssertionError(f"{what[mate]} {env or 'default'}: read {read!r}: device {got[mate][i]} != oracle {want[mate][i]} "
                                         f"({int((got[mate] != want[mate]).sum())} of {self.n} reads differ)")


def merged_walk_takes(setting):
    """Free query ends, forward: the ops a chain may open with through the merged walk (cs_plan_create, solo_first)."""
    _k, _mo, where, rightmost, _rule, _tie = setting
    return not rightmost and where in (abi.CS_WHERE_BACK, abi.CS_WHERE_FRONT, abi.CS_WHERE_ANYWHERE)


def sweep(res, ads, per_adapter, rule, tie, monkeypatch):
    items = [it for r, t, its in U.schedule(len(ads), per_adapter) if (r, t) == (rule, tie) for it in its]
    # free-ended forward ops first, two by two (the plan then also runs through the merged walk), then the others
    items.sort(key=lambda it: (not merged_walk_takes(U.SETTINGS[it[1]]), it))
    done = 0
    for lo in range(0, len(items), 2):
        pair = items[lo:lo + 2]
        if len(pair) == 1:
            pair = pair * 2
        ops = [U.adapter_op(ads[a], U.SETTINGS[si]) for a, si in pair]
        tp = U.one_op_plan([ops[0]], [ops[1]], rule, tie)
        walk = any(merged_walk_takes(U.SETTINGS[si]) for _a, si in pair)
        res.check(tp, [f"adapter {ads[a]} setting {U.SETTINGS[si]}" for a, si in pair], monkeypatch, log_always=walk)
        done += 2
    return done


@pytest.mark.parametrize("rule,tie", GROUPS)
def test_every_read_over_two_letters_against_every_adapter(rule, tie, monkeypatch):
    """{A,C}: 131 071 reads up to length 16 + 29 524 over {A,C,N} up to length 9, x the 248 adapters of length 3..7, 48
    settings each (every setting on 23 or 24 adapters): 11 904 ops x 160 595 reads = 1.9e9 alignments over the four cases."""
    res = Resident(*U.stack([U.reads_universe("AC", 16), U.reads_universe("ACN", 9)], 16))
    assert res.n == 131071 + 29524
    assert sweep(res, U.adapters("AC"), PER_AC, rule, tie, monkeypatch) > 0


@pytest.mark.parametrize("rule,tie", GROUPS)
def test_every_read_over_three_letters_against_every_adapter(rule, tie, monkeypatch):
    """{A,C,G}: 88 573 reads up to length 10 + 21 845 over {A,C,G,N} up to length 7 x the 1 080 adapters of length 3..6;
    the 2 187 adapters of length 7 x the 9 841 reads up to length 8 + 1 365 over {A,C,G,N} up to length 5.  Three (two)
    settings per adapter: every setting on 15 or more adapters."""
    ads = U.adapters("ACG")
    short = [a for a in ads if len(a) <= 6]
    assert len(ads) == 3267 and len(short) == 1080
    res = Resident(*U.stack([U.reads_universe("ACG", 10), U.reads_universe("ACGN", 7)], 12))
    assert res.n == 88573 + 21845
    sweep(res, short, PER_ACG, rule, tie, monkeypatch)
    res = Resident(*U.stack([U.reads_universe("ACG", 8), U.reads_universe("ACGN", 5)], 8))
    # (the rotation is over the adapter's index in the FULL list: lengths 3..6 and length 7 do not repeat each other's settings)
    items7 = [(a, si) for r, t, its in U.schedule(len(ads), PER_ACG7) if (r, t) == (rule, tie) for a, si in its if len(ads[a]) == 7]
    for lo in range(0, len(items7), 2):
        pair = (items7[lo:lo + 2] * 2)[:2]
        ops = [U.adapter_op(ads[a], U.SETTINGS[si]) for a, si in pair]
        walk = any(merged_walk_takes(U.SETTINGS[si]) for _a, si in pair)
        res.check(U.one_op_plan([ops[0]], [ops[1]], rule, tie), [f"adapter {ads[a]} setting {U.SETTINGS[si]}" for a, si in pair],
                  monkeypatch, log_always=walk)


@pytest.mark.parametrize("rule,tie", GROUPS)
def test_the_leading_pair_on_the_two_letter_universe(rule, tie, monkeypatch):
    """The pair that opens every chain the reference compiles (cutseq/run.py:332-355, 544-590): a RightmostFrontAdapter
    and a BackAdapter (or its --force-anywhere form) behind it, every 5' adapter of length 5..7 over {A,C} with six 3'
    adapters of the same set, k of either op in {0, 1, 2}, on every read over {A,C} up to length 14 -- as it is, and behind
    fixed prefixes that move the reads across the walk's groups of eight columns.  Default plan (two scans: short
    adapters are not log-friendly) and CUTSEQ_LOG_ALWAYS=1 (the merged forward walk of myers_pair)."""
    from cutseq_amd import plan as planmod
    ads = U.adapters("AC", 5, 7)
    core, core12 = U.reads_universe("AC", 14), U.reads_universe("AC", 12)
    parts = [U.stack([core], 32), U.stack([core12], 32, pad_front=b"CAACC"), U.stack([core12], 32, pad_front=b"GGGGGGGGGGGG")]
    res = Resident(*[np.concatenate([p[i] for p in parts]) for i in range(3)])
    g = GROUPS.index((rule, tie))
    mine = [(i, p) for i in range(len(ads)) for p in range(PAIR_PARTNERS) if (i + p) % 4 == g]

    def chain(i, p):
        a5, a3 = ads[i], ads[(i * 37 + 11 + 41 * p) % len(ads)]
        k5, k3 = (i + p) % 3, (i // 3 + p // 3) % 3
        mo5 = 3 if (i // 9 + p) % 2 else len(a5)
        mo3 = 1 if (i // 18 + p // 2) % 2 else 3
        anywhere = (i // 36 + p) % 3 == 1
        return [planmod.rightmost_front(a5, U.rate_for(k5, len(a5)), mo5, abi.CS_F_ADAPTER5),
                planmod.back(a3, U.rate_for(k3, len(a3)), mo3, anywhere, abi.CS_F_ADAPTER3)], (a5, k5, mo5, a3, k3, mo3, anywhere)

    for lo in range(0, len(mine), 2):
        pair = (mine[lo:lo + 2] * 2)[:2]
        (c1, d1), (c2, d2) = chain(*pair[0]), chain(*pair[1])
        res.check(U.one_op_plan(c1, c2, rule, tie), [f"pair {d1}", f"pair {d2}"], monkeypatch, log_always=True)
