"""A small universe, ENUMERATED: every read over a 2-, 3- or 4-letter alphabet up to some length against every short
adapter over the same alphabet, and every setting of the aligner the reference can reach (and a few it cannot).

VERDICT r4, missing item 3: the filter's "settle without the DP" rules (trim_kernel.hip.inc, myers_verdict; DESIGN.md
section 2.2) were guarded by randomized adversarial tests only, and two of them were wrong as first written.  A sample
finds a wrong rule when it happens to hit it; an enumeration cannot miss it inside its universe.  The path protected is
``Aligner.locate`` as the reference calls it (cutseq/run.py:332-370, 544-615).

What is enumerated (shared by the CPU twin tests/test_exhaustive_cpu.py -- C oracle against the Python restatement --
and the GPU test tests/test_gpu_exhaustive.py -- HIP kernels against the C oracle):

  reads     READS2  {A,C}    length 0..16   131 071      READS2N  {A,C,N}    length 0..9    29 524
            READS3  {A,C,G}  length 0..10    88 573      READS3N  {A,C,G,N}  length 0..7    21 845
  adapters  every string of length 3..7 over {A,C} (248) / over {A,C,G} (3 267)
  settings  SETTINGS = k in {0, 1, 2} (error rate (k + 1/2) / m) x min_overlap in {1, 3, m} x the seven Where flag sets
            x rightmost on / off x both selection rules x both indel tie orders = 504

The full triple product (reads x adapters x settings) is 1.6e11 alignments over the two-letter universe alone: out of
reach for the CPU oracle (a few million alignments per second).  So reads x adapters is enumerated in full and the
settings ROTATE over the adapters: adapter number a gets settings (a * STRIDE + j) mod 504 for j < PER_ADAPTER --
504 is coprime to STRIDE, so every setting meets several adapters (and every read), every adapter meets several
settings (and every read).  Deterministic: no seed, the same alignments every run.
"""
from __future__ import annotations

import itertools
from typing import Iterator, List, Sequence, Tuple

import numpy as np

from cutseq_amd import abi, plan as planmod

WHERES = (abi.CS_WHERE_BACK, abi.CS_WHERE_FRONT, abi.CS_WHERE_PREFIX, abi.CS_WHERE_SUFFIX,
          abi.CS_WHERE_FRONT_NOT_INTERNAL, abi.CS_WHERE_BACK_NOT_INTERNAL, abi.CS_WHERE_ANYWHERE)
REMOVE_BEFORE_WHERES = (abi.CS_WHERE_FRONT, abi.CS_WHERE_PREFIX, abi.CS_WHERE_FRONT_NOT_INTERNAL)

# (k, min_overlap kind, where, rightmost, select rule, indel tie); the last two are plan-wide (cs_params)
SETTINGS: List[Tuple[int, str, int, bool, int, int]] = [
    (k, mo, where, rightmost, rule, tie)
    for rule in (abi.CS_SELECT_LEFTMOST, abi.CS_SELECT_SCORE)
    for tie in (abi.CS_TIE_INSERTION, abi.CS_TIE_DELETION)
    for k in (0, 1, 2)
    for mo in ("1", "3", "m")
    for where in WHERES
    for rightmost in (False, True)
]
assert len(SETTINGS) == 504
STRIDE = 125  # coprime to 504: adapter a starts its settings at a * STRIDE


def reads_universe(alphabet: str, max_len: int, stride: int | None = None):
    """Every string over ``alphabet`` of length 0..max_len, shortest first, in lexicographic order of the letters'
    indices -> (seq [n, stride] uint8, zero-padded; len [n] uint16)."""
    letters = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    a = len(alphabet)
    stride = (max(max_len, 1) + 3) // 4 * 4 if stride is None else stride
    seqs, lens = [], []
    for L in range(max_len + 1):
        n = a ** L
        idx = np.arange(n, dtype=np.int64)
        rows = np.zeros((n, stride), dtype=np.uint8)
        for j in range(L):
            rows[:, j] = letters[(idx // (a ** (L - 1 - j))) % a]
        seqs.append(rows)
        lens.append(np.full(n, L, dtype=np.uint16))
    return np.concatenate(seqs), np.concatenate(lens)


def stack(parts: Sequence[Tuple[np.ndarray, np.ndarray]], stride: int, pad_front: bytes = b""):
    """Several universes as one batch of rows ``stride`` wide; ``pad_front``: a fixed prefix in front of every read (the
    same reads at another offset from the row's / the column groups' boundaries)."""
    n = sum(p[0].shape[0] for p in parts)
    seq = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    at, f = 0, len(pad_front)
    front = np.frombuffer(pad_front, dtype=np.uint8)
    for s, ln in parts:
        k = s.shape[0]
        w = min(s.shape[1], stride - f)
        assert int(ln.max(initial=0)) <= w
        seq[at:at + k, :f] = front
        seq[at:at + k, f:f + w] = s[:, :w]
        lens[at:at + k] = ln + f
        at += k
    qual = np.where(seq > 0, ord("I"), 0).astype(np.uint8)
    return seq, qual, lens


def adapters(alphabet: str, lo: int = 3, hi: int = 7) -> List[str]:
    return ["".join(t) for m in range(lo, hi + 1) for t in itertools.product(alphabet, repeat=m)]


def rate_for(k: int, m: int) -> float:
    """An error rate with int(rate * m) == k and a threshold table that steps inside the adapter: (k + 1/2) / m."""
    return (k + 0.5) / m


def adapter_op(seq: str, setting) -> planmod.AdapterOp:
    k, mo, where, rightmost, _rule, _tie = setting
    m = len(seq)
    op = planmod.AdapterOp("exhaustive", seq, rate_for(k, m), {"1": 1, "3": 3, "m": m}[mo], where,
                           abi.CS_REMOVE_BEFORE if where in REMOVE_BEFORE_WHERES else abi.CS_REMOVE_AFTER,
                           rightmost=rightmost, match_flag=abi.CS_F_ADAPTER3)
    assert op.k == k, (seq, setting, op.k)
    return op


def settings_of(adapter_index: int, per_adapter: int) -> Iterator[int]:
    for j in range(per_adapter):
        yield (adapter_index * STRIDE + j) % len(SETTINGS)


def schedule(n_adapters: int, per_adapter: int):
    """-> [(rule, tie, [(adapter index, setting index), ...])]: the (adapter, setting) pairs of the rotation, grouped by
    the plan-wide switches so that two of them can share one paired plan (mate 1 and mate 2 run different ops over the
    same reads)."""
    groups = {}
    for a in range(n_adapters):
        for si in settings_of(a, per_adapter):
            s = SETTINGS[si]
            groups.setdefault((s[4], s[5]), []).append((a, si))
    return [(rule, tie, items) for (rule, tie), items in sorted(groups.items())]


def one_op_plan(ops1, ops2, rule: int, tie: int, use_filter: bool = True) -> planmod.TrimPlan:
    return planmod.TrimPlan(r1=planmod.MateChain(list(ops1)), r2=planmod.MateChain(list(ops2)) if ops2 is not None else None,
                            has_umi=False, min_length=0, untrimmed_filter=False, select_rule=rule, use_filter=use_filter,
                            indel_tie=tie)
