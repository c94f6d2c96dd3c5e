"""The BASELINE.json configurations as (plan, seeded synthetic batch) pairs.

One definition for ``bench.py`` (what is timed) and ``tests/`` (what is checked against the oracle), so the
parity tests cover the very schemes, barcode sets and read generators the bench lines are quoted on.

  config2  10 M single-end 150 bp reads, one 3' adapter AGATCGGAAGAGC, 10 % error tolerance
  config3  paired 2x150 bp, full TAKARAV3 scheme + --trim-polyA (UMI + mask + poly-T/A + q-trim): the headline
  config4  paired 2x150 bp, custom -a scheme: inline barcode + 8-nt UMI + dual adapters, --ensure-inline-barcode
           (reference chain: cutseq/run.py:592-603, 771-784)
  config5  paired 2x150 bp, 96-plex inline-barcode demultiplex + dual adapters + q-trim in one pass (extension:
           equals 96 ``--ensure-inline-barcode`` runs)
"""
from __future__ import annotations

import random
from typing import List, Optional

import numpy as np

from . import plan as planmod, synth
from .common import BUILDIN_ADAPTERS, BarcodeConfig

READ_LEN = 150
WORKLOADS = ("config3", "config2", "config4", "config5")
CONFIG4_SCHEME = "ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNN>AGATCGGAAGAGCACACGTC"
CONFIG5_PLEX, CONFIG5_LEN = 96, 8


def _edit_distance(a: str, b: str) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def config5_barcodes(seed: int = 96) -> List[str]:
    """96 barcodes of 8 nt, pairwise edit distance >= 3 (seeded greedy pick)."""
    rng = random.Random(seed)
    codes: List[str] = []
    while len(codes) < CONFIG5_PLEX:
        c = "".join(rng.choice("ACGT") for _ in range(CONFIG5_LEN))
        if all(_edit_distance(c, o) >= 3 for o in codes):
            codes.append(c)
    return codes


def config5_scheme(codes) -> str:
    return f"ACACGACGCTCTTCCGATCT({codes[0]})>AGATCGGAAGAGCACACGTC"


def is_paired(workload: str) -> bool:
    return workload != "config2"


def scheme_label(workload: str) -> str:
    return {"config3": "TAKARAV3", "config2": "-a AGATCGGAAGAGC", "config4": CONFIG4_SCHEME,
            "config5": "P5(96 x 8 nt)>P7 --demux-barcodes"}[workload]


def make_plan(workload: str, use_filter: bool = True) -> planmod.TrimPlan:
    if workload == "config2":
        tp = planmod.single_adapter_plan("AGATCGGAAGAGC", 0.1, 3, min_length=0)
    elif workload == "config4":
        st = planmod.CutadaptConfig()
        st.ensure_inline_barcode = True
        tp = planmod.compile_paired(BarcodeConfig(CONFIG4_SCHEME), st)
    elif workload == "config5":
        codes = config5_barcodes()
        st = planmod.CutadaptConfig()
        st.demux_barcodes = codes
        tp = planmod.compile_paired(BarcodeConfig(config5_scheme(codes)), st)
    elif workload == "config3":
        st = planmod.CutadaptConfig()
        st.trim_polyA = True
        tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
    else:
        raise ValueError(f"unknown workload {workload!r}")
    tp.use_filter = use_filter
    return tp


def make_batch(workload: str, n: int, first_index: int = 0, threads: Optional[int] = None) -> synth.SynthBatch:
    """``n`` pairs (config 2: reads) of the workload, global indices ``first_index ..``: any split of the index
    range over calls / ranks yields the same bytes (config 5's barcode planting is keyed by ``first_index`` of the
    call, so compare like with like there)."""
    if workload == "config5":
        codes = config5_barcodes()
        batch = synth.generate_pairs(n, READ_LEN, config5_scheme(codes), first_index=first_index, threads=threads)
        # every pair gets one of the 96 barcodes (2 % get none of them), 1 % sequencing error per base on top
        rng = np.random.default_rng(first_index + 5)
        table = np.frombuffer("".join(codes).encode(), dtype=np.uint8).reshape(CONFIG5_PLEX, CONFIG5_LEN)
        pick = rng.integers(0, CONFIG5_PLEX, size=n)
        bc_bases = table[pick].copy()
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        foreign = rng.random(n) < 0.02
        bc_bases[foreign] = acgt[rng.integers(0, 4, size=(int(foreign.sum()), CONFIG5_LEN))]
        err = rng.random((n, CONFIG5_LEN)) < 0.01
        bc_bases[err] = acgt[rng.integers(0, 4, size=int(err.sum()))]
        batch.seq1[:, :CONFIG5_LEN] = bc_bases
        return batch
    if workload == "config4":
        return synth.generate_pairs(n, READ_LEN, CONFIG4_SCHEME, first_index=first_index, threads=threads)
    if workload == "config3":
        return synth.generate_pairs(n, READ_LEN, first_index=first_index, threads=threads)
    if workload == "config2":
        return synth.generate_single_adapter(n, READ_LEN, first_index=first_index, threads=threads)
    raise ValueError(f"unknown workload {workload!r}")


def fill_device(workload: str, n: int, ptrs, first_index: int = 0, stream=None) -> int:
    """:func:`make_batch` written straight into device arrays by the generator's device form (same bytes as the host
    form for the same global indices; ``ptrs`` as :func:`synth.generate_pairs_device` takes them) -> row stride.
    config 5 plants its barcodes with numpy on the host and keeps the host form."""
    if workload == "config4":
        return synth.generate_pairs_device(n, ptrs, READ_LEN, CONFIG4_SCHEME, first_index=first_index, stream=stream)
    if workload == "config3":
        return synth.generate_pairs_device(n, ptrs, READ_LEN, first_index=first_index, stream=stream)
    if workload == "config2":
        return synth.generate_pairs_device(n, ptrs, READ_LEN, "ACACGACGCTCTTCCGATCT>AGATCGGAAGAGC", single_end=True,
                                           poly_fraction=0.0, art5_fraction=0.0, first_index=first_index, stream=stream)
    raise ValueError(f"no device generator for workload {workload!r}")
